// flat_common.hpp -- constants, key packing, bf16 helpers, tile arguments and geometries shared by the pre-filter kernels
// (part of the pre-filter translation unit pf_flat.hip: included there, in order; gfx950 only)
#pragma once

namespace pf {

#ifndef PF_TK
#define PF_TK 16
#endif
constexpr int TK = PF_TK;                         // K slab depth of the distance tiles
constexpr int KQ = TK / 4;                        // lanes covering one row of a slab (16 B each)
#ifndef PF_SEL_CAP
#define PF_SEL_CAP 2048
#endif
constexpr uint32_t SEL_CAP = PF_SEL_CAP;          // reservoir capacity (keys)
constexpr uint32_t K_MAX = 1024;                  // largest k
// k_select: one workgroup per query, of 1024 threads when there are few queries (at most one workgroup per CU: the in-LDS
// sorts run with every pair on its own thread) and of 256 threads for batches (more workgroups resident per CU)
constexpr uint32_t SEL_ROUND = 1024;              // keys a reservoir round can add
constexpr uint64_t KEY_INF = 0x7F800000FFFFFFFFull;   // (+inf, id 2^32-1): sorts after every real key

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ uint64_t make_key(float d, uint32_t id) { return ((uint64_t)__float_as_uint(d) << 32) | id; }

// ---- bf16 operands --------------------------------------------------------------------------------------------
// SIFT-like vectors (the reference's dataset: 8-bit values stored as fp32) are integers of magnitude <= 256: exact in
// bf16 (8 significant bits), every product x*y is an integer of at most 2^16 and, with d <= 128, every partial sum of a dot
// product is an integer of magnitude <= 2^23 -- exactly representable in fp32.  The bf16 matrix instruction
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation) therefore returns the same accumulator, bit for bit, as the k-ordered
// fp32 fmaf chain of the f32 instruction, whatever order it adds in, at 16 times the rate.  Eligibility is CHECKED ON THE
// DEVICE, value by value (integer, |v| <= 256): the base when the index is created, the queries at the start of every
// search (per 128-query tile).  Nothing is assumed about the data, and a search needs no host synchronisation to pick its
// path.  Operands that fail the check keep the bf16 tiles as a CONSERVATIVE FILTER (k_l2_tile16): the image is the nearest
// bf16 of every value, the thresholds are lowered by the bound on that rounding, and the distance of every survivor is the
// fp32 chain over the fp32 rows -- (D, I) are the fp32-operand loop's either way.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;      // 16 bytes in registers (HIP's uint4 struct in an array stays in scratch)
constexpr float BF16_EXACT_MAX = 256.f;
__device__ __forceinline__ bool bf16_exact(float v) { return v == rintf(v) && fabsf(v) <= BF16_EXACT_MAX; }

// Three bf16 pieces of an fp32 value, most significant first, by truncation: v = p0 + p1 + p2 exactly (24 significant bits
// = 3 x 8; every remainder v - p is exact).  +-inf comes back as (+-inf, 0, 0).
__device__ __forceinline__ void bf16_split3(float v, uint32_t (&piece)[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t b = __float_as_uint(v) & 0xFFFF0000u;
        piece[i] = b >> 16;
        v = (b & 0x7FFFFFFFu) == 0x7F800000u ? 0.f : v - __uint_as_float(b);
    }
}
constexpr uint32_t BF16_ONE = 0x3F80u, BF16_SIGN = 0x8000u;
// nearest bf16 (ties to even) of a finite fp32: |bf16 - v| <= 2^-8 |v| (what the filter margin of the inexact path prices)
__device__ __forceinline__ uint16_t bf16_rne(float v) {
    const uint32_t b = __float_as_uint(v);
    return (uint16_t)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
}
constexpr uint32_t AUX16 = 8;                   // 16-bit words a base row of the image carries behind its d values (below)


using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// ---- 8-bit operands in matrix-fragment order (the streamed int8 walk, flat_tile8.hpp) -----------------------------------------------
// v_mfma_i32_16x16x64_i8 takes, per lane l, 16 consecutive k of row / column l & 15 starting at k = 16 (l >> 4).  The image keeps the base in exactly
// that order: a PIECE of 1 KiB = 16 consecutive base rows x one 64-deep k-step, byte address ((l >> 4) * 16 + (l & 15)) * 16 inside the piece =
// the lane's 16 bytes, so one global_load_dwordx4 per wave IS an operand fragment (1 KiB contiguous, no LDS staging, no transposition).  Pieces
// are ordered [16-row block][k-step]; rows are padded with zeros (value - 128 = 0 contributes nothing) to whole 64-deep k-steps.
__host__ __device__ constexpr uint32_t frag8_ksteps(uint32_t d) { return (d + 63) / 64; }
__host__ __device__ inline size_t frag8_offset(size_t row, uint32_t k, uint32_t ksteps) {
    return ((row >> 4) * ksteps + (k >> 6)) * 1024 + ((((k & 63u) >> 4) << 4) + (row & 15)) * 16 + (k & 15u);
}
// the columns' threshold halves, two per lane and step of 32 rows adjacent (row 16 b + c of a step at 2 c + b): one 8-byte load per lane
__host__ __device__ inline size_t frag8_c0_index(size_t row) { return (row & ~(size_t)31) + ((row & 15) << 1) + ((row >> 4) & 1); }

// One 128x128 tile of distances per workgroup (256 threads = 4 waves, each wave a 64x64 quadrant as
// 2x2 MFMA 32x32 tiles).  Rows of the tile are queries, columns are base rows, so that a stored
// accumulator register covers 32 consecutive floats of one query's slab row.
struct TileArgs {
    const float *xq; const float *xb; const float *qn; const float *bn;
    float *slab;            // [nq][slab_ld]                         (FILTER == false)
    uint32_t nq, d; size_t nb_first, nb_count; uint32_t slab_ld;
    const float *tau;       // [nq] running k-th distance            (FILTER == true)
    uint32_t *cand_cnt;     // [nq] survivors appended so far (may exceed cap: overflow marker)
    uint64_t *cand;         // [nq][cap] packed keys
    uint32_t cap;
    uint32_t n_qtiles;
    // exactly-representable data (see "bf16 operands" below): 16-bit images of the queries / the base, and per 128-query
    // tile a word that is non-zero when some value of the tile is NOT exactly representable (then the fp32 loop runs)
    const uint16_t *xq16; const uint16_t *xb16; const uint32_t *q_inexact;
    // 8-bit data (every value an integer in [0, 255]: "8-bit integer operands" below): images of value - 128 as int8, the base rows with
    // their half of the threshold behind them; q_inexact bit 2 = some value of the query tile is outside that range
    const int8_t *xq8; const int8_t *xb8;
    uint32_t base_exact;    // every value of the base is exactly representable in bf16
    float bn_max;           // largest |y|^2 of the base (the inexact path's filter margin)
    uint32_t i8_old;        // the LDS-tiled int8 walk (tile16_walk<.., I8>) instead of the streamed one (tile8_walk): batches of one or two query tiles (pf_flat.hip), or PF_FLAT_I8_OLD
    // the streamed int8 walk (flat_tile8.hpp): the base in fragment order, its columns' threshold halves, the queries' sum (x - 128)
    const int8_t *xb8f; const int *c0f; const int *qsx8;
    uint32_t only_flagged;  // k_l2_tile beside the slab tiles (flat_wide16.hpp): only the query tiles whose word has bit 1 set
};

// Tile geometry: TM queries x TN base rows per workgroup of 256 threads (4 waves laid out WM x WN); a wave owns
// MI x NJ MFMA blocks of 32 x 32.
//   128 x 128 (2 x 2 waves, 2 x 2 blocks)   the batch geometry: every operand value fetched from LDS feeds two MFMAs
//   TM = 32 / 64, TN = 256 (1 x 4 waves)    small batches: a 128-row tile would spend 4x / 2x the matrix work on
//                                           padding rows and turn an HBM-bound scan of the base into an MFMA-bound one
template <int TM_, int TN_, int WM_, int WN_>
struct TileGeo {
    static constexpr int TM = TM_, TN = TN_, WM = WM_, WN = WN_;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int MI = TM / (32 * WM), NJ = TN / (32 * WN);   // MFMA blocks per wave
    static constexpr int ROWS_PER_IT = THREADS / KQ;
    static constexpr int ITA = (TM + ROWS_PER_IT - 1) / ROWS_PER_IT, ITB = TN / ROWS_PER_IT;   // fetch/commit iterations per thread
    static constexpr int LDA = TM + 1, LDB = TN + 1;                 // k-major LDS rows padded by one float
    static_assert(THREADS == 256 && TN % ROWS_PER_IT == 0 && (TM % ROWS_PER_IT == 0 || TM < ROWS_PER_IT) && MI >= 1 && NJ >= 1, "unsupported tile geometry");
};
using GeoBatch = TileGeo<128, 128, 2, 2>;
// the bf16 tiles (k_l2_tile16): 128 queries x PF_B16_TN base rows per workgroup.  128 columns: two 34 KiB column tiles + the survivor list =
// 80 KiB, two workgroups per CU at a 256-register budget.  64 columns: 47 KiB, THREE workgroups per CU at 168 registers -- a wave does half
// the matrix work per barrier, but a third wave per SIMD fills the pipe while the others wait (measured: DESIGN.md 4.3).
#ifndef PF_B16_TN
#define PF_B16_TN 128
#endif
using Geo16 = TileGeo<128, PF_B16_TN, PF_B16_TN == 64 ? 4 : 2, PF_B16_TN == 64 ? 1 : 2>;   // 64 columns: 4 x 1 waves of 32 x 64 (32 query-fragment registers, not 64)
constexpr int B16_WG_PER_CU = PF_B16_TN == 64 ? 3 : 2;
// rows of 144 .. 256 values: the query fragments of a 64-row wave tile would fill 128 registers, so a wave takes 32 query rows x 64 columns
// (64 fragment registers at d = 256) and the column tile is 64 rows (2 x 33 KiB at d = 256: still two workgroups per CU)
using Geo16W = TileGeo<128, 64, 4, 1>;
template <int D, bool WIDE = (D > 128)> struct Geo16Of { using type = Geo16; static constexpr int WG_PER_CU = B16_WG_PER_CU; };
template <int D> struct Geo16Of<D, true> { using type = Geo16W; static constexpr int WG_PER_CU = 2; };
using GeoSmall64 = TileGeo<64, 256, 1, 4>;
using GeoSmall32 = TileGeo<32, 256, 1, 4>;

// ---- wave-level helpers -------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {                      // orders this wave's LDS traffic for the compiler; the hardware keeps it in order
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// inclusive prefix sum over the 64 lanes by DPP (no LDS round trips: a scan by __shfl_up is six dependent ds_bpermute)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);       // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);       // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);       // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);       // row_shr:8: every row of 16 holds its own scan
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);       // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);       // row_bcast:31 into rows 2 and 3
    return v;
}

}  // namespace pf
