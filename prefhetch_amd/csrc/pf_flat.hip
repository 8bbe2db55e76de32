// pf_flat.hip -- brute-force squared-L2 pre-filter (IndexFlatL2::search semantics), the gathered
// exact distances of Server::preciseSearch and the row gather of Server::preciseVectorPIR /
// retrieve_centroids.  gfx950 only.
//
// Reference call sites this stands in for:
//   faiss::IndexFlatL2 m_Quantizer           /root/reference/include/server/server_lib.h:14, src/server/server_lib.cpp:33
//   sort_nearest_centroids (executed L2 shortlist)  /root/reference/src/client/client_lib.cpp:50-81
//   Server::preciseSearch                     /root/reference/src/server/server_lib.cpp:140-167
//   Server::preciseVectorPIR / retrieve_centroids   server_lib.cpp:169-196 / 101-109
//
// Search pipeline (all launches on the caller's stream, nothing synchronises):
//   distances  k_l2_tile: LDS-tiled fp32 tiles of ||x||^2 + ||y||^2 - 2 x.y (the decomposition faiss uses for
//              nq >= 20, clamped at 0) on the f32 matrix pipe -- v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered
//              fmaf chain, so a distance is plain IEEE fp32 and can be recomputed exactly by scalar code.
//   bootstrap  the first few thousand base rows go through a [nq][chunk] slab and k_select (one workgroup per
//              query keeping a reservoir of packed (distance bits, id) keys in LDS, compacted by a bitonic sort
//              when it fills -- faiss ReservoirTopN does the same on the CPU for k >= 100).  This yields each
//              query's running top-k and its k-th distance tau.
//   streaming  the rest of the base is processed in geometrically growing chunks whose tile kernel FILTERS in its
//              epilogue: only distances <= tau are appended to the query's candidate list (survivors counted with
//              ballots, every row of a wave reserving its range in one atomic round trip; per workgroup through LDS
//              when there are few queries) -- expected k * chunk / rows_seen survivors -- and k_select merges them into the running
//              top-k and tightens tau.  No distance slab is written or re-read.  If a candidate list overflows
//              (adversarial row order), k_select re-derives that query's chunk exactly by recomputing the
//              distances with the same fmaf chain: slower, never wrong.
// Keys order by (distance, id), which fixes the tie order faiss leaves undefined.
#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>
#include "pf_common.hpp"

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

// row norms, fp32 fma chain in index order
__global__ void __launch_bounds__(256) k_row_norms(const float *__restrict__ x, size_t n, uint32_t d, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *r = x + i * d;
    float acc = 0.f;
    for (uint32_t k = 0; k < d; ++k) acc = fmaf(r[k], r[k], acc);
    out[i] = acc;
}

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

// row norms (fp32 fma chain in index order) + 16-bit image + eligibility.  A workgroup of 64 threads takes ROWS rows: the rows
// are read coalesced (and converted / checked) by all lanes into LDS, then lane r chains row r's norm out of LDS (row pitch
// d + 1 floats: conflict-free).  d <= MAXD <= PREP_MAX_D; wider rows take the one-thread-per-row kernel below.  ROWS = 64 (32) for the
// base (millions of rows), 4 for a batch of queries (1024 rows in 64 rows per workgroup were 16 workgroups and 37 us).
// The image has `pitch16` 16-bit words per row.  With `aux` (the base), words d .. d+7 of a row hold the column's half of the
// threshold term the bf16 tiles feed to the matrix pipe as a ninth k-step: (-b0, -b1, -b2, 1, 1, 1, 0, 0), b0 + b1 + b2 =
// |y|^2 / 2 exactly (bf16_split3); the query's half is built by the tile kernel (k_l2_tile16).
constexpr uint32_t PREP_MAX_D = 256;                            // rows up to 128 values: 64 per workgroup; up to 256: 32 (the staging tile stays at 33 KiB)
template <uint32_t ROWS, uint32_t MAXD>
__global__ void __launch_bounds__(64) k_rows_prep(const float *__restrict__ x, size_t n, uint32_t d, float *__restrict__ norms,
                                                  uint16_t *__restrict__ x16, uint32_t pitch16, bool aux, uint32_t *__restrict__ inexact,
                                                  uint32_t rows_per_flag, int8_t *__restrict__ x8 = nullptr, uint32_t pitch8 = 0) {
    __shared__ float tile[ROWS * (MAXD + 1)];
    const size_t r0 = (size_t)blockIdx.x * ROWS;
    const uint32_t rows = (uint32_t)(n - r0 < ROWS ? n - r0 : ROWS), total = rows * d, lane = threadIdx.x;
    const float *src = x + r0 * d;
    uint32_t bad = 0, bad8 = 0;                                   // bit (row / rows_per_flag within this block's span) ... kept per lane
    for (uint32_t e = lane; e < total; e += 64) {
        const float v = src[e];
        const uint32_t r = e / d, k = e - r * d;
        tile[r * (d + 1) + k] = v;
        if (x16) x16[(r0 + r) * pitch16 + k] = bf16_rne(v);                              // exact when the value passes; nearest otherwise
        const uint32_t fbit = 1u << (rows_per_flag ? ((r0 + r) / rows_per_flag - r0 / rows_per_flag) : 0);
        if (inexact && !bf16_exact(v)) bad |= fbit;
        if (x8) {                                                                        // 8-bit data: value - 128 as int8 (meaningless, and flagged, otherwise)
            const bool ok8 = v == rintf(v) && v >= 0.f && v <= 255.f;
            x8[(r0 + r) * (size_t)pitch8 + k] = (int8_t)(ok8 ? (int)v - 128 : 0);
            if (!ok8) bad8 |= fbit;
        }
    }
    if (bad | bad8) {                                             // a block of <= 64 rows touches at most two flags (rows_per_flag >= 64) or one
        const uint32_t w0 = (bad & 1u) | ((bad8 & 1u) << 2), w1 = ((bad >> 1) & 1u) | (((bad8 >> 1) & 1u) << 2);
        if (w0) atomicOr(&inexact[rows_per_flag ? r0 / rows_per_flag : 0], w0);
        if (w1) atomicOr(&inexact[r0 / rows_per_flag + 1], w1);
    }
    __syncthreads();
    if (lane < rows) {
        const float *row = tile + lane * (d + 1);
        float acc = 0.f;
        for (uint32_t k = 0; k < d; ++k) acc = fmaf(row[k], row[k], acc);
        norms[r0 + lane] = acc;
        if (x16 && aux) {
            uint32_t b[3];
            bf16_split3(0.5f * acc, b);
            u32x4 w;
            w[0] = (b[0] ^ BF16_SIGN) | ((b[1] ^ BF16_SIGN) << 16);
            w[1] = (b[2] ^ BF16_SIGN) | (BF16_ONE << 16);
            w[2] = BF16_ONE | (BF16_ONE << 16);
            w[3] = 0;
            *reinterpret_cast<u32x4 *>(x16 + (r0 + lane) * pitch16 + d) = w;        // 16-byte aligned: d and pitch16 are multiples of 8
        }
        if (x8 && aux) {                                         // the column's half of the integer threshold (tile16_walk): c0 = -floor(C / 2), C = |y|^2 - 256 sum (y - 128)
            int sy = 0;
            for (uint32_t k = 0; k < d; ++k) sy += (int)row[k] - 128;
            const int Cc = (int)acc - 256 * sy;
            u32x4 w;
            w[0] = (uint32_t)(-(Cc >> 1)); w[1] = (uint32_t)sy; w[2] = 0; w[3] = 0;      // (sum y' for the unfiltered launch, which forms distances)
            *reinterpret_cast<u32x4 *>(x8 + (r0 + lane) * (size_t)pitch8 + d) = w;   // 16-byte aligned: d and pitch8 are multiples of 16
        }
    }
}

// Inexact base: the column's share of the filter margin goes into its threshold words, b0 + b1 + b2 = |y|^2 (1/2 - 1.05 x 2^-8)
// (k_l2_tile16: the bound on the operands' rounding is (2^-8 + 2^-17)(|x|^2 + |y|^2), priced per column -- a base whose rows
// differ widely in length would otherwise pay the longest row's margin in every column).
constexpr float BF16_MARGIN = 1.05f * 0x1p-8f;
__global__ void __launch_bounds__(256) k_aux_margin(uint16_t *__restrict__ x16, const float *__restrict__ norms, size_t n, uint32_t d, uint32_t pitch16) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    uint32_t b[3];
    bf16_split3(norms[r] * (0.5f - BF16_MARGIN), b);
    u32x4 w;
    w[0] = (b[0] ^ BF16_SIGN) | ((b[1] ^ BF16_SIGN) << 16);
    w[1] = (b[2] ^ BF16_SIGN) | (BF16_ONE << 16);
    w[2] = BF16_ONE | (BF16_ONE << 16);
    w[3] = 0;
    *reinterpret_cast<u32x4 *>(x16 + r * pitch16 + d) = w;
}

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

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

// A ROWS-row x 32-k slab is staged in two steps so that the global loads of slab s+1 are in flight while the
// matrix pipe works on slab s: fetch (global -> registers: thread t holds row t/KQ + ROWS_PER_IT*it, k = (t%KQ)*4 .. +3)
// and commit (registers -> LDS, transposed to lds[k][row]).
// FAST (d a multiple of the slab depth): no k bounds, and rows past the end re-read the last valid row instead of
// being predicated off -- their products land in accumulator rows / columns the epilogue never emits.
template <bool FAST, int ROWS, int IT, int ROWS_PER_IT>
__device__ __forceinline__ void slab_fetch(float4 (&v)[IT], const float *__restrict__ src, size_t row0, size_t rows_valid,
                                           uint32_t d, uint32_t k0, int tid) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int row = tid / KQ + ROWS_PER_IT * it;
        const uint32_t k = k0 + (tid % KQ) * 4;
        if (ROWS < ROWS_PER_IT && row >= ROWS) break;            // operand narrower than one sweep of the workgroup
        if constexpr (FAST) {
            const size_t rr = (size_t)row < rows_valid ? (size_t)row : rows_valid - 1;
            v[it] = *reinterpret_cast<const float4 *>(src + (row0 + rr) * (size_t)d + k);
            continue;
        }
        v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((size_t)row < rows_valid) {
            const float *p = src + (row0 + row) * (size_t)d + k;
            if (((d & 3) == 0) && k + 3 < d) v[it] = *reinterpret_cast<const float4 *>(p);
            else {
                if (k < d) v[it].x = p[0];
                if (k + 1 < d) v[it].y = p[1];
                if (k + 2 < d) v[it].z = p[2];
                if (k + 3 < d) v[it].w = p[3];
            }
        }
    }
}

template <int ROWS, int IT, int ROWS_PER_IT, int LD>
__device__ __forceinline__ void slab_commit(float *lds, const float4 (&v)[IT], int tid) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int row = tid / KQ + ROWS_PER_IT * it;
        if (ROWS < ROWS_PER_IT && row >= ROWS) break;
        const int kk = (tid % KQ) * 4;
        lds[(kk + 0) * LD + row] = v[it].x;
        lds[(kk + 1) * LD + row] = v[it].y;
        lds[(kk + 2) * LD + row] = v[it].z;
        lds[(kk + 3) * LD + row] = v[it].w;
    }
}

// Epilogue of one distance tile, shared by the fp32 and the bf16 loops: acc = x.y of TM x TN (query, base row) pairs as the
// 32x32 matrix instructions leave it, C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31].  `stage`: at least 4*TM floats
// of LDS that no wave still reads.  Contains barriers (before any wave-uniform early return): call from all waves.
// EXACT (the bf16 tiles: every quantity an integer below 2^24): dist <= tau  <=>  x.y >= (|x|^2 - tau)/2 + |y|^2/2, all three
// terms and their sum exactly representable (half-integers of magnitude <= 2^23), so the first sweep compares the
// accumulator with a per-(row, column) threshold -- one add and one compare per distance instead of add, fma, compare --
// and returns the verdicts of the distance test bit for bit.  Keys are still built from the distance itself.
// PRESTAGED: the caller has already written this query tile's rows to `sA` (l2_tile_stage_rows) and passed a barrier.
template <class GEO, bool EXACT>
__device__ __forceinline__ void l2_tile_stage_rows(float *sA, int tid, float row_qn, float row_tau) {
    constexpr int TM = GEO::TM;
    if (tid < TM) {                                              // (norm, threshold) pairs: one 8-byte LDS read per use
        sA[2 * tid] = row_qn;
        sA[2 * tid + 1] = row_tau;
        if constexpr (EXACT) sA[3 * TM + tid] = 0.5f * (row_qn - row_tau);   // +inf for rows past nq (tau = -inf): nothing passes
        reinterpret_cast<uint32_t *>(sA)[2 * TM + tid] = 0;     // per-row survivor count of this workgroup (small batches)
    }
}
template <bool FILTER, class GEO, bool AGG, bool EXACT = false, bool PRESTAGED = false>
__device__ __forceinline__ void l2_tile_epilogue(const TileArgs &p, f32x16 (&acc)[GEO::MI][GEO::NJ], float *sA, size_t q0, int wm, int tid,
                                                 const size_t (&col)[GEO::NJ], const bool (&col_ok)[GEO::NJ], const float (&bnv)[GEO::NJ],
                                                 float row_qn, float row_tau) {
    constexpr int TM = GEO::TM, MI = GEO::MI, NJ = GEO::NJ;
    const int lane = tid & 63;
    // epilogue: C[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
    if constexpr (FILTER) {
        // per-query norm and threshold of this tile's rows, staged in LDS (sA is free now)
        if constexpr (!PRESTAGED) {
            __syncthreads();
            l2_tile_stage_rows<GEO, EXACT>(sA, tid, row_qn, row_tau);
            __syncthreads();
        }
    }
    if constexpr (FILTER && AGG) {
        // Small batches: few queries take every survivor of the chunk, so one global atomic per half-wave would
        // serialise on a handful of counters.  Survivors are first counted per row in LDS, then each row reserves its
        // range with ONE global atomic per workgroup, then the keys are written.
        uint32_t *s_cnt = reinterpret_cast<uint32_t *>(sA) + 2 * TM, *s_base = s_cnt + TM;
        auto verdict = [&](int i, int r, float2 qt, float (&dist)[NJ], bool (&pass)[NJ], uint32_t (&hm)[NJ]) {
            uint32_t tot = 0;
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                dist[jj] = fmaf(-2.f, acc[i][jj][r], qt.x + bnv[jj]);
                pass[jj] = dist[jj] <= qt.y;
                hm[jj] = (uint32_t)(__ballot(pass[jj]) >> (lane & 32));
                tot += __popc(hm[jj]);
            }
            return tot;
        };
        // (norm, threshold) pairs are fetched from LDS eight registers at a time: one exposed latency per batch
        auto pairs = [&](int i, int r8, float2 (&qts)[8]) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                qts[e] = *reinterpret_cast<const float2 *>(sA + 2 * (wm + 32 * i + ((r8 + e) & 3) + 8 * ((r8 + e) >> 2) + 4 * (lane >> 5)));
        };
        uint32_t loc[MI][16];                                        // offset of this half-wave inside its row's range
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r8 = 0; r8 < 16; r8 += 8) {
                float2 qts[8];
                pairs(i, r8, qts);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = r8 + e;
                    float dist[NJ]; bool pass[NJ]; uint32_t hm[NJ];
                    const uint32_t tot = verdict(i, r, qts[e], dist, pass, hm);
                    const int lrow = wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    loc[i][r] = ((lane & 31) == 0 && tot) ? atomicAdd(&s_cnt[lrow], tot) : 0u;
                }
            }
        __syncthreads();
        if (tid < TM) { const uint32_t n = s_cnt[tid]; s_base[tid] = n ? atomicAdd(&p.cand_cnt[q0 + tid], n) : 0u; }
        __syncthreads();
        const uint32_t below = (1u << (lane & 31)) - 1u;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r8 = 0; r8 < 16; r8 += 8) {
                float2 qts[8];
                pairs(i, r8, qts);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = r8 + e;
                    float dist[NJ]; bool pass[NJ]; uint32_t hm[NJ];
                    if (__ballot(verdict(i, r, qts[e], dist, pass, hm) != 0) == 0) continue;
                    const int lrow = wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const size_t row = q0 + lrow;
                    uint32_t base = s_base[lrow] + __shfl(loc[i][r], lane & 32);
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj) {
                        if (pass[jj]) {
                            const uint32_t pos = base + __popc(hm[jj] & below);
                            if (pos < p.cap) p.cand[row * p.cap + pos] = make_key(dist[jj] < 0.f ? 0.f : dist[jj], (uint32_t)(p.nb_first + col[jj]));
                        }
                        base += __popc(hm[jj]);
                    }
                }
            }
        return;
    }
    if constexpr (FILTER) {
        // Batches: the survivors of one accumulator register of one half-wave all belong to ONE query, and every query of
        // the wave's 32*MI rows shows up in exactly one (register, half) -- so lane L can own local row L.  First sweep:
        // count each row's survivors (ballots; the counts land in their lanes with v_writelane).  Then every lane with a
        // non-zero count reserves its row's range with one atomic -- all rows of the wave in ONE memory round trip instead
        // of one dependent round trip per row.  Second sweep, only over registers that had survivors: write the keys.
        uint32_t row_cnt = 0, hit[MI];
        float bnh[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) bnh[jj] = 0.5f * bnv[jj];        // NaN past the end of the chunk: compares false
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            hit[i] = 0;
#pragma unroll
            for (int r8 = 0; r8 < 16; r8 += 8) {
            // (norm, threshold) pairs of eight registers fetched from LDS together: one latency per batch
            float2 qts[8];
            float rqs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int lrow = wm + 32 * i + ((r8 + e) & 3) + 8 * ((r8 + e) >> 2) + 4 * (lane >> 5);
                if constexpr (EXACT) rqs[e] = sA[3 * TM + lrow];
                else qts[e] = *reinterpret_cast<const float2 *>(sA + 2 * lrow);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = r8 + e;
                uint64_t m[NJ], any = 0;
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    if constexpr (EXACT) m[jj] = __ballot(acc[i][jj][r] >= rqs[e] + bnh[jj]);
                    else {
                        const float dist = fmaf(-2.f, acc[i][jj][r], qts[e].x + bnv[jj]);
                        m[jj] = __ballot(dist <= qts[e].y);              // tau >= 0: same verdict before and after the clamp at 0
                    }
                    any |= m[jj];
                }
                if (any == 0) continue;                                  // wave-uniform, and the common case in late chunks
                uint32_t t0 = 0, t1 = 0;
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    t0 += __builtin_popcount((uint32_t)m[jj]);
                    t1 += __builtin_popcount((uint32_t)(m[jj] >> 32));
                }
                const int rho = 32 * i + (r & 3) + 8 * (r >> 2);         // local row of half 0; half 1 is 4 rows further
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(row_cnt) : "s"(t0), "n"(rho));
                asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(row_cnt) : "s"(t1), "n"(rho + 4));
                hit[i] |= 1u << r;
            }
            }
        }
        uint32_t any_hit = 0;
#pragma unroll
        for (int i = 0; i < MI; ++i) any_hit |= hit[i];
        if (any_hit == 0) return;                                    // wave-uniform
        uint32_t row_base = 0;
        if (row_cnt) row_base = atomicAdd(&p.cand_cnt[q0 + wm + lane], row_cnt);      // lanes >= 32*MI hold 0
        const uint32_t below = (1u << (lane & 31)) - 1u;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (hit[i] == 0) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!((hit[i] >> r) & 1u)) continue;                     // wave-uniform
                const int rho = 32 * i + (r & 3) + 8 * (r >> 2);
                const int lrow = wm + rho + 4 * (lane >> 5);
                const size_t row = q0 + lrow;
                const float2 qt = *reinterpret_cast<const float2 *>(sA + 2 * lrow);
                const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)row_base, rho);
                const uint32_t b1 = (uint32_t)__builtin_amdgcn_readlane((int)row_base, rho + 4);
                uint32_t base = (lane & 32) ? b1 : b0;
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const float dist = fmaf(-2.f, acc[i][jj][r], qt.x + bnv[jj]);
                    const bool pass = dist <= qt.y;
                    const uint32_t hm = (uint32_t)(__ballot(pass) >> (lane & 32));
                    if (pass) {
                        const uint32_t pos = base + __popc(hm & below);
                        if (pos < p.cap) p.cand[row * p.cap + pos] = make_key(dist < 0.f ? 0.f : dist, (uint32_t)(p.nb_first + col[jj]));
                    }
                    base += __popc(hm);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t row = q0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row >= p.nq) continue;
            const float qnv = p.qn[row];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                if (!col_ok[jj]) continue;
                float dist = fmaf(-2.f, acc[i][jj][r], qnv + bnv[jj]);
                dist = dist < 0.f ? 0.f : dist;
                p.slab[row * p.slab_ld + col[jj]] = dist;
            }
        }
    }
}

// One fp32 distance tile (query tile qt, column tile ct of the chunk): body of k_l2_tile, also the fallback of the bf16
// kernel for query tiles that are not exactly representable.  smem: F32_TILE_LDS<GEO> bytes, 16-byte aligned.
template <class GEO> constexpr size_t F32_TILE_LDS = sizeof(float) * 2 * TK * (GEO::LDA + GEO::LDB);
template <bool FILTER, class GEO, bool FAST, bool AGG>
__device__ __forceinline__ void l2_tile_f32(const TileArgs &p, char *smem, uint32_t qt, uint32_t ct) {
    constexpr int TM = GEO::TM, TN = GEO::TN, LDA = GEO::LDA, LDB = GEO::LDB, MI = GEO::MI, NJ = GEO::NJ, RPI = GEO::ROWS_PER_IT;
    float (*sAb)[TK * LDA] = reinterpret_cast<float (*)[TK * LDA]>(smem);   // two k-slabs in flight: one feeds the MFMAs, the next is being filled
    float (*sBb)[TK * LDB] = reinterpret_cast<float (*)[TK * LDB]>(smem + sizeof(float) * 2 * TK * LDA);
    float *const sA = sAb[0];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t q0 = (size_t)qt * TM;
    const size_t c0 = (size_t)ct * TN;                         // column inside the chunk
    if (c0 >= p.nb_count) return;
    const size_t q_valid = p.nq - q0 < (size_t)TM ? p.nq - q0 : (size_t)TM;
    const size_t c_valid = p.nb_count - c0 < (size_t)TN ? p.nb_count - c0 : (size_t)TN;
    const int wm = (wave / GEO::WN) * (32 * MI), wn = (wave % GEO::WN) * (32 * NJ);
    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    // operands of the epilogue, requested now so that their latency hides under the whole tile: the norms of this lane's
    // columns (NaN past the end of the chunk when filtering: such a distance compares false with every threshold) and,
    // for the first TM threads, one query row's (norm, threshold)
    size_t col[NJ]; bool col_ok[NJ]; float bnv[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        col[jj] = c0 + wn + 32 * jj + (lane & 31);
        col_ok[jj] = col[jj] < p.nb_count;
        bnv[jj] = col_ok[jj] ? p.bn[p.nb_first + col[jj]] : (FILTER ? __builtin_nanf("") : 0.f);
    }
    float row_qn = 0.f, row_tau = -INFINITY;                      // rows past nq: nothing passes
    if constexpr (FILTER) {
        if (tid < TM && q0 + tid < p.nq) { row_qn = p.qn[q0 + tid]; row_tau = p.tau[q0 + tid]; }
    }
    float4 ra[GEO::ITA], rb[GEO::ITB];
    slab_fetch<FAST, TM, GEO::ITA, RPI>(ra, p.xq, q0, q_valid, p.d, 0, tid);
    slab_fetch<FAST, TN, GEO::ITB, RPI>(rb, p.xb, p.nb_first + c0, c_valid, p.d, 0, tid);
    slab_commit<TM, GEO::ITA, RPI, LDA>(sAb[0], ra, tid);
    slab_commit<TN, GEO::ITB, RPI, LDB>(sBb[0], rb, tid);
    if (TK < p.d) {
        slab_fetch<FAST, TM, GEO::ITA, RPI>(ra, p.xq, q0, q_valid, p.d, TK, tid);
        slab_fetch<FAST, TN, GEO::ITB, RPI>(rb, p.xb, p.nb_first + c0, c_valid, p.d, TK, tid);
    }
    __syncthreads();
    // slab s feeds the matrix pipe from buffer s&1 while slab s+1 (in registers since the previous iteration) is
    // committed to the other buffer and slab s+2 is requested from memory: one barrier per slab
    for (uint32_t k0 = 0, cur = 0; k0 < p.d; k0 += TK, cur ^= 1) {
        if (k0 + TK < p.d) {
            slab_commit<TM, GEO::ITA, RPI, LDA>(sAb[cur ^ 1], ra, tid);
            slab_commit<TN, GEO::ITB, RPI, LDB>(sBb[cur ^ 1], rb, tid);
            if (k0 + 2 * TK < p.d) {
                slab_fetch<FAST, TM, GEO::ITA, RPI>(ra, p.xq, q0, q_valid, p.d, k0 + 2 * TK, tid);
                slab_fetch<FAST, TN, GEO::ITB, RPI>(rb, p.xb, p.nb_first + c0, c_valid, p.d, k0 + 2 * TK, tid);
            }
        }
        // operand fragments of k-step s+1 are read from LDS while the MFMAs of step s run
        float a[2][MI], b[2][NJ];
        const float *fa = sAb[cur] + (lane >> 5) * LDA + wm + (lane & 31), *fb = sBb[cur] + (lane >> 5) * LDB + wn + (lane & 31);
#pragma unroll
        for (int i = 0; i < MI; ++i) a[0][i] = fa[32 * i];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) b[0][jj] = fb[32 * jj];
#pragma unroll
        for (int ks = 0; ks < TK; ks += 2) {
            const int cur = (ks >> 1) & 1, nxt = cur ^ 1;
            if (ks + 2 < TK) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[nxt][i] = fa[(ks + 2) * LDA + 32 * i];
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) b[nxt][jj] = fb[(ks + 2) * LDB + 32 * jj];
            }
            __builtin_amdgcn_sched_barrier(0);         // keep the reads ahead of the MFMAs they overlap with
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][jj], acc[i][jj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    l2_tile_epilogue<FILTER, GEO, AGG>(p, acc, sA, q0, wm, tid, col, col_ok, bnv, row_qn, row_tau);
}

// AGG (the 32- and 64-row geometries): survivors are aggregated per row in LDS before the global append -- with few
// queries the global counters are hot, and a workgroup there spans 256 columns of every row.
template <bool FILTER, class GEO, bool FAST, bool AGG = false>   // AGG only matters with FILTER
__global__ void __launch_bounds__(GEO::THREADS, 2) k_l2_tile(TileArgs p) {
    __shared__ __align__(16) char smem[F32_TILE_LDS<GEO>];
    // XCD-aware tile order (1-D grid): blocks b and b+8 share an XCD under round-robin placement, so XCD x takes the
    // column tiles = x (mod 8) and runs all query tiles of one column tile back to back -- the base tile is
    // fetched from HBM once into that XCD's L2 and re-read from there by the other query tiles.
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    l2_tile_f32<FILTER, GEO, FAST, AGG>(p, smem, j % p.n_qtiles, (j / p.n_qtiles) * 8 + xcd);
}

// ---- bf16 tiles (d = 64 or 128) ---------------------------------------------------------------------------------------
// What the matrix pipe computes here is the FILTER VALUE itself, not the dot product: a ninth k-step adds the threshold,
//     acc = x.y - |y|^2/2 - R,         R = (|x|^2 - tau)/2 - margin,
// the column's half coming with the base row (three bf16 pieces behind its d values, k_rows_prep), the row's half built once
// per walk from the staged thresholds (three pieces of R against three ones).  dist <= tau  <=>  x.y >= (|x|^2 - tau)/2 +
// |y|^2/2 (all terms half-integers below 2^23 on this path), so a distance can only pass if acc >= 0: the epilogue reads
// SIGN BITS, one v_alignbit_b32 per accumulator value.  (Before: add, subtract, shift per value on the vector pipe -- with
// two waves per SIMD the tile walk is bound by the instructions a wave issues, 830 per tile of which 32 were matrix
// instructions; phase stamps in tools/flat_stamps.py.)
// Exactness.  While |R| <= 2^22 every partial sum of the nine k-steps is a half-integer of magnitude <= 2^24, the
// accumulator is exact in any order of addition, the margin is 0 and the filter is the distance test itself.  A larger |R|
// (tau far above the query norm: data with negative values, or tau = +inf) may round partial sums, by less than
// 2^-19 |R| in total: margin = max(256, 2^-14 |R|) keeps the filter conservative, and a candidate too many is harmless
// (k_select orders candidates by their distance).  The DISTANCE of a survivor does not come from the accumulator: a
// survivor is parked as (row, base id) and flush() recomputes its dot product from the two 16-bit rows, sixteen lanes
// per survivor (v_dot2c_f32_bf16: integer sums below 2^24, exact in any order) -- the same number, bit for bit, as the
// fp32 chain.  That costs about 7 instructions per survivor, once per walk, instead of a register-indexed read and ~85
// instructions inside the tile loop.
// One workgroup keeps the bf16 image of its 128-query tile in registers (whole k) and walks `group` consecutive 128-row
// column tiles: a tile of the base image (rows of d values + 8 threshold words = PITCH bytes) is one contiguous block of
// memory and is copied as such by LDS-DMA into one of two LDS buffers (the odd row pitch in 16-byte units makes ds_read_b128
// of 32 consecutive rows conflict-free): while the matrix pipe and the sign sweep work on tile t, tile t+1 is on its way
// into the other buffer -- one barrier per tile.
// Inexact operands (a base or a query tile with a value that is not exactly representable; flags set on the device): the
// same tiles as a conservative filter -- margin 1.05 x 2^-8 (|x|^2 + |y|^2) on the thresholds, the survivors' distances by
// the fp32 chain over the fp32 rows (flush) -- and fp32 tiles where that filter cannot help (k_l2_tile16, select_one).
struct Pend16 {
    static constexpr uint32_t CAP = 1760;     // 2 x 34 KiB of operands + 2 KiB of rows + this list fit twice into a CU's 160 KiB
    static constexpr uint32_t HIGH = CAP * 3 / 4;   // a list longer than this is worked off at once; a shorter one waits for more (pend16_flush)
    uint32_t id[CAP];                  // base row
    uint8_t loc[CAP];                  // local query row
    uint32_t rcnt[128], rbase[128];
    uint32_t n;
    uint32_t wcnt[2][4];               // verdict records each wave holds, by tile parity (the int8 walk's rings)
};

#ifdef PF_FLAT_STAMPS        // experiments (tools/flat_stamps.py): s_memtime at the phase boundaries of the tile walk
#define PF_FS_WGS 32
#define PF_FS_TILES 16
#define PF_FS_K 6
__device__ unsigned long long pf_flat_stamp_buf[PF_FS_WGS * 4 * PF_FS_TILES * PF_FS_K];
#define PF_FSTAMP(k) do { if (fs_on && (tid & 63) == 0 && ct - ct0 < PF_FS_TILES) \
    pf_flat_stamp_buf[(((blockIdx.x - 256) * 4 + (tid >> 6)) * PF_FS_TILES + (ct - ct0)) * PF_FS_K + (k)] = __builtin_readcyclecounter(); } while (0)
__device__ unsigned long long pf_flat_flush_stamp_buf[PF_FS_WGS * 4 * 8 * 8];      // [workgroup][wave][flush of the walk][stamp]
#define PF_FLSTAMP(k) do { if (p.nb_count >= 400000 && blockIdx.x >= 256 && blockIdx.x < 256 + PF_FS_WGS && (tid & 63) == 0 && flush_no < 8) \
    pf_flat_flush_stamp_buf[(((blockIdx.x - 256) * 4 + (tid >> 6)) * 8 + flush_no) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PF_FSTAMP(k) do { } while (0)
#define PF_FLSTAMP(k) do { } while (0)
#endif

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v);       // (defined with the selection kernels below)

// The verdict words of MT tiles (surv[u][jj]: tile ct_base + u, column block jj; bit 31 - s = accumulator row s of this lane)
// are decoded here, once per MT tiles, by the lane that owns them: each survivor takes a slot of the list and its index
// within its query row (LDS atomics), a row with survivors reserves its range of the candidate list with ONE global atomic,
// then sixteen lanes per survivor recompute the dot product from the two 16-bit rows and the group's first lane writes the
// key.  A list too small for everything (dense early chunks) is worked off in rounds: the words not yet decoded stay in
// the registers.  Barriers inside: call from all threads.
#ifndef PF_FLUSH_U
#define PF_FLUSH_U 6
#endif
#ifndef PF_APPROX_UNROLL       // float4 pieces of a survivor's base row requested before the first is used (fp32 chain of the inexact path)
#define PF_APPROX_UNROLL 8      // ... query rows staged in LDS (a walk's last flush)
#endif
#ifndef PF_APPROX_UNROLL2
#define PF_APPROX_UNROLL2 4     // ... both rows from memory
#endif
#define PF_FLUSH_INLINE __forceinline__
// RING: the verdict words come from this wave's ring of records in LDS instead (the int8 walk appends a (word, tile, column block, lane) record per
// non-zero word as the tile ends -- a ballot and an LDS write, no barrier -- and calls this only when a ring is nearly full or the walk ends:
// the parking above cost a quarter of the int8 walk's time); rc records, ct_base = the walk's first tile.
template <int D, int MT, int NJ, int TN, bool I8 = false, bool RING = false>
__device__ PF_FLUSH_INLINE void pend16_flush(const TileArgs &p, Pend16 &pd, const float *sA, size_t q0, int tid, uint32_t (&surv)[MT][NJ],
                                             uint32_t ct_base, int wm, int wn, bool approx, char *xstage, uint32_t q_valid, bool final,
                                             uint32_t flush_no = 0, const uint2 *ring = nullptr, uint32_t rc = 0) {
    (void)flush_no;
    using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
    // lanes per survivor: LU = D / 8 of them hold 16 bytes of both rows each, rounded up to a power of two (L) for the DPP sum
    // (I8: the rows of the int8 images, 16 values per lane: half the lanes and half the bytes per survivor, twice the survivors per pass)
    constexpr uint32_t LU = I8 ? D / 16 : D / 8, L = LU <= 2 ? 2 : LU <= 4 ? 4 : LU <= 8 ? 8 : LU <= 16 ? 16 : 32, G = 256 / L;         // G survivors per pass
    const int lane = tid & 63;
    uint32_t left = 0;
    if constexpr (!RING) {
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) left += __popc(surv[u][jj]);
    }
    uint32_t cur = 0, meta = 0, rb = 0;                               // RING: what is left of this lane's current record; the next batch of 64 records
    PF_FLSTAMP(0);
    // Parking (LDS only) happens at every call; the expensive part -- barriers, a returning global atomic per row, the rows of
    // every survivor fetched again -- only once the list is long (HIGH), overflowed (a lane could not park everything), or the
    // walk ends (`final`).  In the long late chunks a workgroup parks ~90 survivors per call: it now pays for ONE round trip to
    // memory per walk instead of one per MT tiles (the flushes were 35 % of the tile kernels' time: profiles/r03_flat_ablation.txt).
    for (;;) {
        PF_FLSTAMP(1);
        if constexpr (RING) {
            // batches of 64 records, a record per lane, until the ring is empty or the list is full (a lane could not park every bit of its record)
            for (;;) {
                if (__ballot(cur != 0) == 0) {
                    if (rb >= rc) break;                                 // wave-uniform
                    const uint32_t idx = rb + (uint32_t)lane;
                    const uint2 rec = idx < rc ? ring[idx] : make_uint2(0u, 0u);
                    cur = rec.x; meta = rec.y;
                    rb += 64;
                }
                const uint32_t cnt = (uint32_t)__popc(cur);
                const uint32_t incl = wave_incl_scan(cnt);
                const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);      // > 0: some lane holds a record
                uint32_t slot = 0;
                if (lane == 0) slot = atomicAdd(&pd.n, tot);
                slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot) + incl - cnt;
                uint32_t take = slot < Pend16::CAP ? Pend16::CAP - slot : 0u;
                take = cnt < take ? cnt : take;
                const uint32_t ls = meta & 63u, jj = (meta >> 6) & 3u, trel = meta >> 8;
                while (take) {                                           // highest set bit first (a lane rarely holds more than one)
                    const int b = 31 - __builtin_clz(cur);
                    cur &= ~(1u << b);
                    const int sb = 31 - b, r = sb & 15;
                    const uint32_t lrow = (uint32_t)(wm + 2 * (sb & 16) + (r & 3) + 8 * (r >> 2) + 4 * (ls >> 5));
                    pd.id[slot] = (uint32_t)(p.nb_first + (size_t)(ct_base + trel) * TN + wn + 32 * jj + (ls & 31u));
                    pd.loc[slot] = (uint8_t)lrow;
                    atomicAdd(&pd.rcnt[lrow], 1u);
                    ++slot; --take;
                }
                if (__ballot(cur != 0)) break;                           // the list is full
            }
            left = (cur != 0 || rb < rc) ? 1u : 0u;
        } else {
        // slots: ONE returning LDS atomic per wave (a prefix sum over the lanes' counts), not one per lane with survivors
        const uint32_t incl = wave_incl_scan(left);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t slot = 0;
        if (tot) {                                                       // wave-uniform
            if (lane == 0) slot = atomicAdd(&pd.n, tot);
            slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot) + incl - left;
        }
        uint32_t take = slot < Pend16::CAP ? Pend16::CAP - slot : 0u;
        take = left < take ? left : take;
        left -= take;
        auto park_at = [&](uint32_t w, int b) {                          // survivor = bit b of verdict word w = u * NJ + jj
            const uint32_t u = w / NJ, jj = w % NJ;
            const int s = 31 - b, r = s & 15;
            const uint32_t lrow = (uint32_t)(wm + 2 * (s & 16) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
            pd.id[slot] = (uint32_t)(p.nb_first + (size_t)(ct_base + u) * TN + wn + 32 * jj + (lane & 31));
            pd.loc[slot] = (uint8_t)lrow;
            atomicAdd(&pd.rcnt[lrow], 1u);                              // no return value: the position inside the row is drawn when the key is written
            ++slot;
        };
        if (!tot) {
            // nothing in this wave
        } else if (__ballot(take > 2 || left != 0) == 0) {
            // Sparse case (the long late chunks: ~20 survivors per wave and call): no lane holds more than two.  The words are
            // scanned WITHOUT branches into at most two (word, bit) pairs per lane, then the pairs are parked -- the word-by-word
            // loop below costs a vector-compare -> scalar-branch round trip per word (6 000 cycles per call: phase stamps)
            uint32_t e0 = ~0u, e1 = ~0u;
#pragma unroll
            for (int u = 0; u < MT; ++u)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const uint32_t m = surv[u][jj], w = (uint32_t)(u * NJ + jj);
                    const int b = 31 - __builtin_clz(m | 1u);            // (m | 1: defined for m = 0, unused then)
                    const uint32_t m2 = m & ~(1u << b);
                    const int b2 = 31 - __builtin_clz(m2 | 1u);
                    const uint32_t pk = (w << 5) | (uint32_t)b, pk2 = (w << 5) | (uint32_t)b2;
                    e1 = (m != 0 && e0 != ~0u) ? pk : e1;
                    e0 = (m != 0 && e0 == ~0u) ? pk : e0;
                    e1 = m2 != 0 ? pk2 : e1;
                    surv[u][jj] = 0;
                }
            if (e0 != ~0u) park_at(e0 >> 5, (int)(e0 & 31u));
            if (e1 != ~0u) park_at(e1 >> 5, (int)(e1 & 31u));
        } else {
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                uint32_t m = surv[u][jj];
                if (__ballot(m != 0) == 0) continue;                  // wave-uniform: nothing in this word anywhere in the wave
                auto park = [&]() {                                   // highest set bit of m: one survivor
                    const int b = 31 - __builtin_clz(m);
                    m &= ~(1u << b);
                    park_at((uint32_t)(u * NJ + jj), b);
                    --take;
                };
                // a lane rarely holds more than one bit of a word: the first one without a loop (a loop iteration costs a vector
                // compare -> exec round trip; sixteen word loops were 5 400 of a flush's 15 000 cycles), the rest -- if any lane of
                // the wave has one -- in the loop
                if (m && take) park();
                if (__ballot(m && take)) while (m && take) park();
                surv[u][jj] = m;
            }
        }
        }   // (!RING)
        PF_FLSTAMP(2);
#ifdef PF_ABL_NOEMIT   // ablation (timing only, wrong results): the list is built and dropped
        if constexpr (RING) { __syncthreads(); if (tid == 0) pd.n = 0; if (tid < 128) pd.rcnt[tid] = 0; __syncthreads(); if (__syncthreads_or(left != 0) == 0) return; continue; }
#endif
        const bool any_left = __syncthreads_or(left != 0) != 0;          // (the barrier: everything parked is visible)
        PF_FLSTAMP(3);
        const uint32_t have = pd.n;
        if (have == 0 || (!any_left && !final && have <= Pend16::HIGH)) return;      // workgroup-uniform: the list waits for more
        const uint32_t n = pd.n < Pend16::CAP ? pd.n : Pend16::CAP;
        if (approx) {                                                 // workgroup-uniform
            // inexact operands: the distance of a survivor is the k-ordered fp32 chain over the fp32 rows -- what the fp32 tiles
            // and the oracle evaluate -- one lane per survivor (the order of the additions is part of the result).  A lane
            // reading its own two rows 16 bytes at a time makes the texture path see 64 different cache lines per instruction; the
            // query rows are only 128 different ones, so -- when the tile buffers are free: the walk's last flush -- they are
            // copied into LDS once, coalesced (row pitch D * 4 + 16 bytes: conflict-free 16-byte reads by 64 different rows).
            constexpr uint32_t XP = D * 4 + 16;
            // the tile buffers hold all 128 staged rows, or (64-column tiles) half of them: then the list is worked off in two halves by row
            // ... both of them at the walk's last flush, the one whose tile is done at a flush in mid-walk (the other holds the next tile): the list is
            // worked off in rounds of as many query rows as fit -- 128 / 64 at d = 128, 64 / 32 at d = 256
            constexpr uint32_t BUF = TN * (D + AUX16) * 2u;
            constexpr uint32_t X_ALL = 2u * BUF >= 128u * XP ? 128u : 64u, X_ONE = BUF >= 128u * XP ? 128u : BUF >= 64u * XP ? 64u : 32u;
            static_assert(2u * BUF >= X_ALL * XP && BUF >= X_ONE * XP, "the staged query rows fit the tile buffers");
            const uint32_t XROWS = final ? X_ALL : X_ONE;
            if (tid < 128) {
                const uint32_t c = pd.rcnt[tid];
                pd.rbase[tid] = c ? atomicAdd(&p.cand_cnt[q0 + tid], c) : 0u;
                pd.rcnt[tid] = 0;
            }
            for (uint32_t r0 = 0; r0 < (xstage ? 128u : 1u); r0 += XROWS) {           // (without staging: one round over everything)
                if (xstage) {
                    if (r0) __syncthreads();                              // the first half's readers are done
                    for (uint32_t i = tid; i < XROWS * (D / 4); i += 256) {
                        const uint32_t row = r0 + i / (D / 4), seg = i % (D / 4);
                        *reinterpret_cast<float4 *>(xstage + (row - r0) * XP + seg * 16) =
                            reinterpret_cast<const float4 *>(p.xq + (q0 + (row < q_valid ? row : q_valid - 1)) * (size_t)D)[seg];
                    }
                }
                __syncthreads();
                for (uint32_t e = tid; e < n; e += 256) {
                    const uint32_t row = pd.loc[e], id = pd.id[e];
                    if (xstage && (row < r0 || row >= r0 + XROWS)) continue;
                    const uint32_t pos = atomicAdd(&pd.rbase[row], 1u);
                    if (pos >= p.cap) continue;                           // the list of this query overflowed: k_select rescans the chunk
                    const float4 *y = reinterpret_cast<const float4 *>(p.xb + (size_t)id * D);
                    float acc = 0.f;
                    if (xstage) {
                        const float4 *x = reinterpret_cast<const float4 *>(xstage + (row - r0) * XP);
#pragma unroll PF_APPROX_UNROLL
                        for (int t = 0; t < D / 4; ++t) {
                            const float4 a = x[t], b = y[t];
                            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
                        }
                    } else {
                        const float4 *x = reinterpret_cast<const float4 *>(p.xq + (q0 + row) * (size_t)D);
#pragma unroll PF_APPROX_UNROLL2
                        for (int t = 0; t < D / 4; ++t) {
                            const float4 a = x[t], b = y[t];
                            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
                        }
                    }
                    const float dist = fmaf(-2.f, acc, sA[2 * row] + p.bn[id]);
                    p.cand[(q0 + row) * p.cap + pos] = make_key(dist < 0.f ? 0.f : dist, id);
                }
            }
            __syncthreads();
            if (tid == 0) pd.n = 0;
            __syncthreads();                                          // the reset is visible before anyone parks again
            if (!any_left) return;
            continue;
        }
        // U survivors per group and pass: their rows are requested first, and in the first pass the per-row reservations (a
        // returning global atomic per row with survivors) travel at the same time -- one round trip to memory, not two
        constexpr int U = PF_FLUSH_U;
        const uint32_t g = (uint32_t)tid / L, l = (uint32_t)tid % L;
        for (uint32_t e0 = 0; e0 < n; e0 += G * U) {
            u32x4 va[U], vb[U];
            uint32_t loc[U], id[U];
            float bnv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t e = e0 + u * G + g < n ? e0 + u * G + g : n - 1;     // idle groups of the last pass repeat the last survivor
                loc[u] = pd.loc[e]; id[u] = pd.id[e];
                if (LU == L || l < LU) {
                    if constexpr (I8) {
                        va[u] = *reinterpret_cast<const u32x4 *>(p.xq8 + (q0 + loc[u]) * (size_t)D + 16 * l);
                        vb[u] = *reinterpret_cast<const u32x4 *>(p.xb8 + (size_t)id[u] * (D + 16) + 16 * l);
                    } else {
                        va[u] = *reinterpret_cast<const u32x4 *>(p.xq16 + (q0 + loc[u]) * (size_t)D + 8 * l);
                        vb[u] = *reinterpret_cast<const u32x4 *>(p.xb16 + (size_t)id[u] * (D + AUX16) + 8 * l);
                    }
                } else {
                    constexpr uint32_t Z = I8 ? 0x80808080u : 0u;                  // (int8 image: value 0 is stored as -128)
                    va[u] = u32x4{Z, Z, Z, Z}; vb[u] = u32x4{Z, Z, Z, Z};          // lanes past the row (its lanes are not a power of two)
                }
                bnv[u] = p.bn[id[u]];
            }
            if (e0 == 0) {                                            // workgroup-uniform
                PF_FLSTAMP(4);
                if (tid < 128) {
                    const uint32_t c = pd.rcnt[tid];
                    pd.rbase[tid] = c ? atomicAdd(&p.cand_cnt[q0 + tid], c) : 0u;
                    pd.rcnt[tid] = 0;
                }
                __syncthreads();
                PF_FLSTAMP(5);
            }
            // positions inside the rows' reserved ranges: running counts in LDS, all U requested before the first is used
            uint32_t pos[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pos[u] = (l == 0 && e0 + u * G + g < n) ? atomicAdd(&pd.rbase[loc[u]], 1u) : ~0u;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float s = 0.f;
                if constexpr (I8) {
                    // the stored bytes are value - 128: flipping the top bit gives the value back as an unsigned byte, and v_dot4_u32_u8 the exact x.y
                    uint32_t si = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint32_t wa = va[u][w] ^ 0x80808080u, wb = vb[u][w] ^ 0x80808080u;
                        si = __builtin_amdgcn_udot4(wa, wb, si, false);
                    }
                    si += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)si, 0xB1, 0xf, 0xf, true);                            // quad_perm [1,0,3,2]
                    if constexpr (L >= 4) si += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)si, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
                    if constexpr (L >= 8) si += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)si, 0x141, 0xf, 0xf, true);     // row_half_mirror
                    static_assert(!I8 || L <= 8, "int8 rows of at most 128 values");
                    s = (float)si;                                     // below 2^24: exact
                } else {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t wa = va[u][w], wb = vb[u][w];     // through scalars: __builtin_bit_cast applied to va[u][w] itself reads element 0 four times (hipcc 7.2)
                    s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, wa), __builtin_bit_cast(bf16x2, wb), s, false);
                }
                // sum over the L lanes of the group (DPP: quad permutes, then mirrors within 8 and 16 lanes): every lane ends with the total
                s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xf, 0xf, true));     // quad_perm [1,0,3,2]
                if constexpr (L >= 4) s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xf, 0xf, true));     // quad_perm [2,3,0,1]
                if constexpr (L >= 8) s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xf, 0xf, true));    // row_half_mirror
                if constexpr (L >= 16) s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x140, 0xf, 0xf, true));   // row_mirror
                if constexpr (L == 32) s += __shfl_xor(s, 16);          // the neighbouring row of 16 lanes (integers: any order of additions is exact)
                }
                if (pos[u] < p.cap) {                                 // (~0 for idle lanes and groups)
                    const uint32_t row = loc[u];
                    const float dist = fmaf(-2.f, s, sA[2 * row] + bnv[u]);
                    p.cand[(q0 + row) * p.cap + pos[u]] = make_key(dist < 0.f ? 0.f : dist, id[u]);
                }
            }
        }
        PF_FLSTAMP(6);
        __syncthreads();
        if (tid == 0) pd.n = 0;
        __syncthreads();                                              // the reset is visible before anyone parks again
        if (!any_left) return;
    }
}

// FILTER epilogue of the bf16 tiles: the accumulators hold the filter value (above), a distance can pass only where the sign
// bit is clear.  Row s = 16 i + r of a lane ends up in bit 31 - s of the lane's word for its column block.  That is all a tile
// does about its survivors: the words stay in registers until flush() decodes them.
template <class GEO>
__device__ __forceinline__ void l2_tile_verdicts16(f32x16 (&acc)[GEO::MI][GEO::NJ], const bool (&col_ok)[GEO::NJ], uint32_t (&surv)[GEO::NJ]) {
    constexpr int MI = GEO::MI, NJ = GEO::NJ;
    static_assert(MI * 16 == 32 || MI * 16 == 16, "one verdict word per column block: 32 (or 16) accumulator rows per lane");
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        uint32_t fail = 0;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) fail = __builtin_amdgcn_alignbit(fail, __float_as_uint(acc[i][jj][r]), 31);
        if constexpr (MI == 1) fail = (fail << 16) | 0xFFFFu;           // 16 rows per lane: they sit in the word's upper half, the lower half never passes
        surv[jj] = col_ok[jj] ? ~fail : 0u;                              // columns past the end of the chunk re-read rows of the next one
    }
}

#ifndef PF_FLAT_MT
#define PF_FLAT_MT 8
#endif
#ifndef PF_DMA_SPREAD
#define PF_DMA_SPREAD 1       // the LDS-DMA requests of the next column tile interleaved with this tile's matrix instructions (k_l2_tile16)
#endif
using i32x4v = __attribute__((ext_vector_type(4))) int;
using i32x16v = __attribute__((ext_vector_type(16))) int;
// one matrix instruction of the tile loop on 16-byte operand fragments: 32 x 32 x 16 bf16 -> fp32, or 32 x 32 x 32 int8 -> int32 (the accumulator
// registers hold the integers' bit patterns; the verdict sweep only reads their sign bits)
template <bool I8>
__device__ __forceinline__ f32x16 tile_mma(const bf16x8 a, const bf16x8 b, const f32x16 c) {
    if constexpr (I8)
        return __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, a), __builtin_bit_cast(i32x4v, b), __builtin_bit_cast(i32x16v, c), 0, 0, 0));
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// I8: the operands are the int8 images (8-bit data), the matrix instruction v_mfma_i32_32x32x32_i8 -- the cycles of the bf16 instruction at twice
// the depth, 16 instead of 36 of them per tile at d = 128, half the bytes copied into and read from LDS.  The accumulators start at the
// (row + column) halves of the threshold instead of zero (what the ninth k-step does for bf16) and are exact integers.
constexpr int AUX8 = 16;                        // bytes a base row of the int8 image carries behind its d values: c0 (int32), 12 spare
template <bool FILTER, int D, bool I8>                              // D = row length (a multiple of 16 up to 256): every loop below is compile-time
__device__ __forceinline__ void tile16_walk(const TileArgs &p, const uint32_t group, const uint32_t n_groups, char *smem, float *stage, Pend16 &pend,
                                            const uint32_t qt, const uint32_t grp, const uint32_t qflags) {
    using GEO = typename Geo16Of<D>::type;
    constexpr int TM = GEO::TM, TN = GEO::TN, MI = GEO::MI, NJ = GEO::NJ, PITCH = I8 ? D + AUX8 : (D + (int)AUX16) * 2;
    constexpr int KS = I8 ? 32 : 16, STEPS = D / KS;                 // depth of a matrix instruction, k-steps of a tile
    constexpr uint32_t PIECES = TN * PITCH / 16, SWEEPS = PIECES / 256, REM = PIECES % 256;      // 16-byte pieces of a column tile: D = 128: 8 x 256 + 128
    static_assert(PITCH % 32 == 16 && (TN == 128 || TN == 64) && TM == 128 && D % KS == 0, "odd row pitch in 16-byte units; 128 x 128 or 128 x 64 tiles");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (FILTER) {
        if (tid < TM) pend.rcnt[tid] = 0;
        if (tid == 0) pend.n = 0;
    }
    const uint32_t n_ct = (uint32_t)((p.nb_count + TN - 1) / TN);
    const uint32_t ct0 = grp * group, ct1 = ct0 + group < n_ct ? ct0 + group : n_ct;
    // Operands that are NOT exactly representable (the base as a whole, or this query tile; flags set on the device): the
    // unfiltered bootstrap launch writes distances, so it runs the fp32 tile body; a filtered launch keeps the bf16 tiles as a
    // CONSERVATIVE FILTER (row thresholds lowered by the bound on the rounding of the operands, below) and flush() evaluates
    // the survivors with the fp32 chain.
    // A query tile whose candidate lists overflowed in an earlier chunk (bit 1, set by the selection kernel) is one the bf16
    // tiles do not filter -- every distance within the rounding of the operands of the threshold: margin ~ 2^-8 (|x|^2 + |y|^2)
    // against a spread of distances far below that -- and runs fp32 tiles from then on.
    const bool approx = !I8 && (!p.base_exact || (qflags & 1u));    // (qflags: workgroup-uniform; the caller picked I8 for exact 8-bit operands only)
    if (!I8 && ((!FILTER && approx) || (FILTER && (qflags & 2u)))) {
        for (uint32_t ct = ct0; ct < ct1; ++ct) {
            l2_tile_f32<FILTER, GEO, true, false>(p, smem, qt, ct);
            __syncthreads();
        }
        return;
    }
    const size_t q0 = (size_t)qt * TM;
    const uint32_t q_valid = (uint32_t)(p.nq - q0 < (size_t)TM ? p.nq - q0 : (size_t)TM);
    const int wm = (wave / GEO::WN) * (32 * MI), wn = (wave % GEO::WN) * (32 * NJ);
    char *const sB16_0 = smem, *const sB16_1 = smem + TN * PITCH;   // column tiles alternate between two buffers: ONE barrier per tile
    // a column tile is PIECES consecutive 16-byte pieces of the image (the allocation is padded by one tile of zero rows, so
    // the last tile of the base reads in bounds) and is copied as such by LDS-DMA (global_load_lds_dwordx4: no registers, no
    // ds_write): lane t moves pieces t, t + 256, ...; one wave-instruction fills 1 KiB of LDS from its wave-uniform base.
    // Tile t+1 is requested at the top of tile t, into the buffer whose readers passed the barrier that ended tile t-1, and
    // waited for (vmcnt(0)) before the barrier that ends tile t.
    float bn_next[NJ];
    // sweep `it` (0 .. SWEEPS: the last one is the remainder) of column tile ct into buf
    auto stage_sweep = [&](uint32_t ct, char *buf, uint32_t it) {
        const char *img = I8 ? reinterpret_cast<const char *>(p.xb8) : reinterpret_cast<const char *>(p.xb16);
        const char *src = img + (p.nb_first + (size_t)ct * TN) * (size_t)PITCH + tid * 16;
        char *dst = buf + wave * 1024;
        if (it < SWEEPS)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4096 * it),
                                             (__attribute__((address_space(3))) void *)(dst + 4096 * it), 16, 0, 0);
        else if (REM && (uint32_t)tid < REM)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4096 * SWEEPS),
                                             (__attribute__((address_space(3))) void *)(dst + 4096 * SWEEPS), 16, 0, 0);
    };
    auto stage_b = [&](uint32_t ct, char *buf) {
#pragma unroll
        for (uint32_t it = 0; it <= SWEEPS; ++it) stage_sweep(ct, buf, it);
    };
    auto fetch_bn = [&](uint32_t ct) {                               // column norms: the unfiltered (bootstrap) epilogue forms distances
        const size_t c0 = (size_t)ct * TN;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const size_t c = c0 + wn + 32 * jj + (lane & 31);
            bn_next[jj] = c < p.nb_count ? p.bn[p.nb_first + c] : 0.f;
        }
    };
    // the first tile is requested BEFORE the query fragments and row thresholds are loaded: one round trip to memory for the
    // prologue of a walk instead of two (a walk is 8 tiles of ~2 us; the serialised prologue was ~4 us of it)
    stage_b(ct0, sB16_0);
    if constexpr (!FILTER) fetch_bn(ct0);
    // The query operand never changes during the walk: each wave keeps its fragments in registers (lane l: row l & 31 of each
    // 32-row block, 8 consecutive k of every 16-deep step = 16 bytes of the bf16 row image; rows past the end re-read the last
    // valid row -- their products land in accumulator rows the epilogue never emits)
    // (int8: lane l holds 16 consecutive k of every 32-deep step, again 16 bytes)
    bf16x8 afrag[MI][STEPS];
    {
        const char *abase = I8 ? reinterpret_cast<const char *>(p.xq8 + q0 * (size_t)D) : reinterpret_cast<const char *>(p.xq16 + q0 * (size_t)D);
        constexpr int ESZ = I8 ? 1 : 2;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const uint32_t r = wm + 32 * i + (lane & 31);
            const char *row = abase + ((size_t)(r < q_valid ? r : q_valid - 1) * D + (lane >> 5) * (KS / 2)) * ESZ;
#pragma unroll
            for (int ks = 0; ks < STEPS; ++ks) afrag[i][ks] = *reinterpret_cast<const bf16x8 *>(row + ks * 32);
        }
    }
    float row_qn = 0.f, row_tau = -INFINITY;                        // rows past nq: nothing passes
    if constexpr (FILTER) {
        if (tid < TM && q0 + tid < p.nq) { row_qn = p.qn[q0 + tid]; row_tau = p.tau[q0 + tid]; }
        // the rows of the query tile are the same for every column tile: staged once
        l2_tile_stage_rows<GEO, true>(stage, tid, row_qn, row_tau);
        if constexpr (I8) {
            // Integer thresholds.  With x' = x - 128, y' = y - 128 and S = sum x'y' (what the matrix instruction accumulates):
            // x.y = S + 128 (sum x' + sum y') + 16384 d, and dist < tau <=> 2 S > R + C with the row's R = |x|^2 - tau - 256 sum x' - 32768 d
            // and the column's C = |y|^2 - 256 sum y' (all exact integers below 2^26).  2 S > T <=> S >= floor(T / 2) + 1; the accumulators
            // start at r0 + c0 = -(floor(R / 2) + 1) - floor(C / 2) >= -(floor((R + C) / 2) + 1): a distance can pass only where S + r0 + c0 >= 0
            // (a superset by at most the one value at the boundary -- every survivor's distance is evaluated exactly by the flush).
            if (tid < TM) {
                int r0 = -(1 << 30);                                 // rows past nq: nothing passes
                if (q0 + tid < p.nq) {
                    if (row_tau == INFINITY) r0 = 1 << 30;           // fewer than k results so far: everything passes
                    else {
                        const uint32_t *w = reinterpret_cast<const uint32_t *>(p.xq8 + (q0 + tid) * (size_t)D);
                        int sx = 0;
#pragma unroll 8
                        for (int t = 0; t < D / 4; ++t) sx = __builtin_amdgcn_sdot4((int)w[t], 0x01010101, sx, false);
                        const int R = (int)row_qn - (int)row_tau - 256 * sx - 32768 * D;
                        r0 = -(R >> 1) - 1;                          // (>> of a negative int: floor)
                    }
                }
                reinterpret_cast<int *>(stage)[3 * TM + tid] = r0;
            }
        }
    } else if constexpr (I8) {
        // the unfiltered (bootstrap) launch forms distances: x.y = S + 128 (sum x' + sum y') + 16384 d -- the row's sum here, the column's behind its row
        if (tid < TM) {
            const uint32_t r = (uint32_t)tid < q_valid ? (uint32_t)tid : q_valid - 1;
            const uint32_t *w = reinterpret_cast<const uint32_t *>(p.xq8 + (q0 + r) * (size_t)D);
            int sx = 0;
#pragma unroll 8
            for (int t = 0; t < D / 4; ++t) sx = __builtin_amdgcn_sdot4((int)w[t], 0x01010101, sx, false);
            reinterpret_cast<int *>(stage)[3 * TM + tid] = 128 * sx + 16384 * D;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the row half of the threshold k-step: lanes 0..31 carry (1, 1, 1, -r0, -r1, -r2, 0, 0) of their row for k = 0..7, lanes
    // 32..63 (k = 8..15) zeros; r0 + r1 + r2 = R (header comment)
    bf16x8 a_aux[MI];
    int r0v[I8 ? MI : 1][16];                                         // int8: the row halves of the thresholds of this lane's accumulator rows
    if constexpr (I8) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) r0v[i][r] = reinterpret_cast<const int *>(stage)[3 * TM + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
    }
    if constexpr (FILTER && !I8) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int arow = wm + 32 * i + (lane & 31);
            const float rq = stage[3 * TM + arow];                                // (|x|^2 - tau) / 2; +inf for rows past nq
            const float big = fabsf(rq) * 0x1p-14f;
            float margin = fabsf(rq) <= 0x1p22f ? 0.f : (big > 256.f ? big : 256.f);
            // exact operands in rows beyond 128 values: x.y is still exact (integers up to 2^24), but the threshold step adds half-integers to
            // it at magnitudes up to 2^26, where fp32 has none -- its handful of additions can be off by a few units in 2^26: 2^-20 of the
            // bound (|x|^2 + max |y|^2) / 2 + |R| on every partial sum covers them
            if (D > 128 && !approx) margin += 0x1p-20f * (0.5f * (stage[2 * arow] + p.bn_max) + fabsf(rq));
            // inexact operands: |bf16(x).bf16(y) - x.y| <= (2^-7 + 2^-16) sum |x_i y_i| <= (2^-8 + 2^-17) (|x|^2 + |y|^2); the fp32
            // chain that decides in the end, the accumulation inside the matrix pipe and the pieces of the thresholds add a few
            // 2^-24 of the same sum (about 4e-5 (|x|^2 + |y|^2) in all): 1.05 x 2^-8 covers them.
            // An inexact base carries its columns' share in their threshold words (k_aux_margin); an exact base facing an inexact query
            // tile does not, and the row pays for the longest column.
            if (approx) margin += BF16_MARGIN * (stage[2 * arow] + (p.base_exact ? p.bn_max : 0.f)) + 0x1p-20f * fabsf(rq);
            uint32_t r[3];
            bf16_split3(fabsf(rq) == INFINITY ? rq : rq - margin, r);
            u32x4 w;
            w[0] = BF16_ONE | (BF16_ONE << 16);
            w[1] = BF16_ONE | ((r[0] ^ BF16_SIGN) << 16);
            w[2] = (r[1] ^ BF16_SIGN) | ((r[2] ^ BF16_SIGN) << 16);
            w[3] = 0;
            if (lane >= 32) w = u32x4{0, 0, 0, 0};
            a_aux[i] = __builtin_bit_cast(bf16x8, w);
        }
    }
#ifdef PF_FLAT_STAMPS
    const bool fs_on = FILTER && p.nb_count >= 400000 && blockIdx.x >= 256 && blockIdx.x < 256 + PF_FS_WGS;
#endif
    // MT tiles between two flushes: their verdict words stay in registers (a 16-register vector written through a wave-uniform
    // index: the tile loop stays rolled -- unrolled MT times it ran out of registers, and a single scratch reload inside the
    // loop makes hipcc wait for vmcnt(0), i.e. for the LDS-DMA of the next tile, before the matrix work)
    constexpr int MT = PF_FLAT_MT;
    using survx = __attribute__((ext_vector_type(MT * NJ))) uint32_t;
    survx sv;
#pragma unroll
    for (int e = 0; e < MT * NJ; ++e) sv[e] = 0;
    // The int8 walk keeps its verdict words in LDS instead: its two column tiles leave room behind them in the tile buffers of the kernel (sized for
    // the bf16 tiles) for a ring of (word, tile, column block, lane) records per wave, appended to as a tile ends and decoded when a ring is nearly
    // full or the walk ends -- one call of pend16_flush per walk in the long chunks instead of one per MT tiles.
    constexpr size_t SMEM16 = 2 * (size_t)TN * (D + AUX16) * 2 > F32_TILE_LDS<GEO> ? 2 * (size_t)TN * (D + AUX16) * 2 : F32_TILE_LDS<GEO>;
    constexpr uint32_t RING_ROOM = I8 ? (uint32_t)((SMEM16 - 2 * (size_t)TN * PITCH) / (4 * sizeof(uint2))) : 0u;
    constexpr uint32_t RCAP = RING_ROOM >= 1024 ? 1024u : RING_ROOM >= 512 ? 512u : 256u;          // records per wave
    static_assert(!I8 || (RING_ROOM >= 256 && NJ * 64 <= 128), "a ring takes at least two tiles' worth of records");
    uint2 *const ring = reinterpret_cast<uint2 *>(smem + 2 * (size_t)TN * PITCH) + (size_t)wave * RCAP;
    uint32_t rc = 0;                                                  // records in this wave's ring (wave-uniform)
    for (uint32_t ct = ct0; ct < ct1; ++ct) {
        const uint32_t u = (ct - ct0) % MT, cur = (ct - ct0) & 1u;
        char *const buf_cur = cur ? sB16_1 : sB16_0, *const buf_nxt = cur ? sB16_0 : sB16_1;
        PF_FSTAMP(0);
        // The copies of tile ct+1 are requested BETWEEN the matrix instructions of this tile, a sweep per k-step (PF_DMA_SPREAD): a
        // copy instruction holds the wave's issue for ~60-80 cycles; all nine at the top of the tile were 690 cycles in which this
        // wave fed nothing to the matrix pipe, one behind the first matrix instruction of a k-step hides under the 128 cycles the
        // step's four instructions occupy the pipe for.
        const bool more = ct + 1 < ct1;                                // workgroup-uniform
#ifdef PF_ABL_NODMA   // ablation (timing only, wrong results): no copies after the walk's second tile
        if (more && ct < ct0 + 1) stage_b(ct + 1, buf_nxt);
#elif !PF_DMA_SPREAD
        if (more) stage_b(ct + 1, buf_nxt);                           // in flight under this tile's matrix work and epilogue
#endif
        PF_FSTAMP(1);
        const size_t c0 = (size_t)ct * TN;
        size_t col[NJ]; bool col_ok[NJ]; float bnv[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            col[jj] = c0 + wn + 32 * jj + (lane & 31);
            col_ok[jj] = col[jj] < p.nb_count;
            bnv[jj] = FILTER ? 0.f : bn_next[jj];
        }
        if constexpr (!FILTER) { if (ct + 1 < ct1) fetch_bn(ct + 1); }
        // column fragments of k-step s+1 are read from LDS while the matrix instructions of step s run (fenced: left to
        // itself hipcc hoists every fragment read of the tile to the top)
        const char *fbx = buf_cur + (wn + (lane & 31)) * PITCH, *fb = fbx + (lane >> 5) * 16;
        f32x16 acc[MI][NJ];
        if constexpr (I8 && FILTER) {                                // the thresholds' halves instead of zero: row half from registers, column half behind the row
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int c0v = *reinterpret_cast<const int *>(fbx + 32 * jj * PITCH + D);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][jj][r] = __builtin_bit_cast(float, r0v[i][r] + c0v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
        }
        bf16x8 b[2][NJ];
        PF_FSTAMP(2);
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) b[0][jj] = *reinterpret_cast<const bf16x8 *>(fb + 32 * jj * PITCH);
#pragma unroll
        for (int ks = 0; ks < STEPS; ++ks) {
            const int c = ks & 1, n = c ^ 1;
            if (ks + 1 < STEPS) {
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) b[n][jj] = *reinterpret_cast<const bf16x8 *>(fb + 32 * jj * PITCH + (ks + 1) * 32);
            } else if constexpr (FILTER && !I8) {                    // the threshold words behind the row: same 16 bytes for both lane halves
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) b[n][jj] = *reinterpret_cast<const bf16x8 *>(fbx + 32 * jj * PITCH + D * 2);
            }
            __builtin_amdgcn_sched_barrier(0);
#if PF_DMA_SPREAD && !defined(PF_ABL_NODMA)
            // sweeps ks and (for the last step, when STEPS < SWEEPS + 1) the rest, behind the step's first matrix instruction
            acc[0][0] = tile_mma<I8>(afrag[0][ks], b[c][0], acc[0][0]);
            if (more) {
                if (ks + 1 < STEPS) stage_sweep(ct + 1, buf_nxt, ks);
                else {
#pragma unroll
                    for (uint32_t it = STEPS - 1; it <= SWEEPS; ++it) stage_sweep(ct + 1, buf_nxt, it);
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj)
                    if (i || jj) acc[i][jj] = tile_mma<I8>(afrag[i][ks], b[c][jj], acc[i][jj]);
#else
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = tile_mma<I8>(afrag[i][ks], b[c][jj], acc[i][jj]);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (FILTER && !I8) {
            constexpr int c = STEPS & 1;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_aux[i], b[c][jj], acc[i][jj], 0, 0, 0);
        }
        if constexpr (I8 && !FILTER) {                               // integer S -> x.y as fp32 (below 2^24: exact), what the epilogue expects
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int syv = 128 * *reinterpret_cast<const int *>(fbx + 32 * jj * PITCH + D + 4);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float bits = acc[i][jj][r];               // through a scalar: __builtin_bit_cast applied to the vector element itself reads element 0 (hipcc 7.2)
                        acc[i][jj][r] = (float)(__float_as_int(bits) + r0v[i][r] + syv);
                    }
            }
        }
        PF_FSTAMP(3);
        if constexpr (FILTER) {
            uint32_t s1[NJ];
            l2_tile_verdicts16<GEO>(acc, col_ok, s1);
#pragma unroll
#ifdef PF_ABL_NOSURV   // ablation (timing only, wrong results): the verdicts are computed and dropped -- nothing to flush
            for (int jj = 0; jj < NJ; ++jj) sv[u * NJ + jj] = s1[jj] & (p.nq == 0xFFFFFFFFu ? ~0u : 0u);
#else
            for (int jj = 0; jj < NJ; ++jj) {
                if constexpr (I8) {
                    const uint64_t m = __ballot(s1[jj] != 0);
                    if (m) {                                            // wave-uniform
                        if (s1[jj]) ring[rc + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = make_uint2(s1[jj], ((ct - ct0) << 8) | ((uint32_t)jj << 6) | (uint32_t)lane);
                        rc += (uint32_t)__popcll(m);
                    }
                } else {
                    sv[u * NJ + jj] = s1[jj];                           // wave-uniform index: v_movreld
                }
            }
#endif
            if constexpr (I8) { if (lane == 0) pend.wcnt[(ct - ct0) & 1u][wave] = rc; }      // (read after the tile's barrier)
            PF_FSTAMP(4);
        } else {
            // q0 made opaque per tile: otherwise hipcc hoists the row addresses of the slab stores out of the tile loop
            size_t q0t = q0;
            asm volatile("" : "+s"(q0t));
            l2_tile_epilogue<false, GEO, false>(p, acc, stage, q0t, wm, tid, col, col_ok, bnv, row_qn, row_tau);
        }
#ifndef PF_ABL_NOBAR   // ablation (timing only, wrong results): no per-tile barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's pieces of tile ct+1 have landed
        __syncthreads();                                                // the tile's one barrier: the other buffer is complete
#endif
        PF_FSTAMP(5);
        if constexpr (I8 && FILTER) {
            const uint32_t *wc = pend.wcnt[(ct - ct0) & 1u];
            const uint32_t c01 = wc[0] > wc[1] ? wc[0] : wc[1], c23 = wc[2] > wc[3] ? wc[2] : wc[3];
            const bool full = (c01 > c23 ? c01 : c23) > RCAP - 128;      // a tile adds at most 128 records to a ring
            if (full || ct + 1 == ct1) {                                // workgroup-uniform (every wave read the same four counts)
                uint32_t none[MT][NJ] = {};
#ifndef PF_ABL_NODRAIN   // ablation (timing only, wrong results): the records are appended and dropped
                pend16_flush<D, MT, NJ, TN, true, true>(p, pend, stage, q0, tid, none, ct0, wm, wn, false, nullptr, q_valid, ct + 1 == ct1, 0, ring, rc);
#endif
                rc = 0;
            }
        } else if constexpr (FILTER) {
            if (u == MT - 1 || ct + 1 == ct1) {                         // workgroup-uniform
                uint32_t surv[MT][NJ];
#pragma unroll
                for (int e = 0; e < MT * NJ; ++e) surv[e / NJ][e % NJ] = sv[e];
                // (both tile buffers are free for the flush once no tile follows: nothing is in flight into them, nobody reads them; in mid-walk the
                // buffer of the tile just finished is -- its readers passed the barrier above, the next request into it comes with the next tile)
#ifdef PF_ABL_EXACTFLUSH   // ablation (timing only, wrong results on inexact data): survivors by the 16-bit dot products whatever the operands
                constexpr bool abl_exact = true;
#else
                constexpr bool abl_exact = false;
#endif
                pend16_flush<D, MT, NJ, TN, I8>(p, pend, stage, q0, tid, surv, ct - u, wm, wn, approx && !abl_exact, ct + 1 == ct1 ? smem : buf_cur, q_valid, ct + 1 == ct1, (ct - ct0) / MT);
#pragma unroll
                for (int e = 0; e < MT * NJ; ++e) sv[e] = 0;
            }
        }
    }
}

template <bool FILTER, int D>
__global__ void __launch_bounds__(256, Geo16Of<D>::WG_PER_CU) k_l2_tile16(TileArgs p, uint32_t group, uint32_t n_groups) {
    using GEO = typename Geo16Of<D>::type;
    constexpr int TM = GEO::TM, TN = GEO::TN, PITCH = (D + (int)AUX16) * 2;
    constexpr size_t SMEM = 2 * (size_t)TN * PITCH > F32_TILE_LDS<GEO> ? 2 * (size_t)TN * PITCH : F32_TILE_LDS<GEO>;   // the fp32 fallback borrows this LDS
    __shared__ __align__(16) char smem[SMEM];
    __shared__ __align__(16) float stage[4 * TM];                   // the epilogue's per-row (norm, threshold) pairs and counters
    __shared__ Pend16 pend;                                         // survivors parked until the end of the walk (FILTER)
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const uint32_t qt = j % p.n_qtiles, grp = (j / p.n_qtiles) * 8 + xcd;
    if (grp >= n_groups) return;
    const uint32_t qflags = p.q_inexact[qt];                        // workgroup-uniform
    // 8-bit data on both sides (the base's image exists only then; bit 2 of the tile's word: a query value outside [0, 255]; bit 1: its
    // lists overflowed, fp32 tiles from then on): the int8 matrix instruction.  Anything else: bf16 operands.
    if constexpr (D % 32 == 0 && D <= 128) {
        if (p.xb8 && !(qflags & 7u)) { tile16_walk<FILTER, D, true>(p, group, n_groups, smem, stage, pend, qt, grp, qflags); return; }
    }
#ifdef PF_ABL_I8ONLY   // experiment (register count of the int8 walk on its own; other query tiles are not processed: wrong results for them)
    if constexpr (!(D % 32 == 0 && D <= 128))
#endif
    tile16_walk<FILTER, D, false>(p, group, n_groups, smem, stage, pend, qt, grp, qflags);
}

// ---- selection -----------------------------------------------------------------------------------
struct SelArgs {
    const float *slab; uint32_t slab_ld;     // mode 0: distances of this chunk, [nq][slab_ld]
    size_t nb_first, nb_count;               // ids of this chunk are nb_first + column
    uint64_t *state;                         // [nq][k] keys carried between chunks (ascending)
    uint32_t *state_cnt;                     // [nq]
    float *tau;                              // [nq] k-th distance so far (+inf while fewer than k)
    uint32_t *cand_cnt; const uint64_t *cand; uint32_t cap;   // mode 1: survivors of the filtered tile kernel
    const float *xq, *xb, *qn, *bn; uint32_t d;                // mode 1 overflow fallback: exact recomputation
    uint32_t k;
    int mode, first, last;
    float *D; int64_t *I;                    // written when last (either may be null)
    uint32_t *packed;                        // written when last, if not null: [nq][k]{id low word, id high word, distance bits}
    uint32_t *q_flags;                       // bf16 tiles: per 128-query tile, bit 1 is set here when a candidate list of the tile overflowed
    float bn_max; uint32_t base_exact;       // ... or when the bootstrap predicts that the tiles will not filter (select_one)
};

// final results of one query position: the caller's (D, I) and / or the 12-byte exchange record of the multi-GPU gather
__device__ __forceinline__ void emit_result(const SelArgs &p, size_t pos, bool ok, uint64_t key) {
    const uint32_t dbits = ok ? (uint32_t)(key >> 32) : 0x7F800000u;            // +inf
    const int64_t id = ok ? (int64_t)(uint32_t)key : -1;
    if (p.D) p.D[pos] = __uint_as_float(dbits);
    if (p.I) p.I[pos] = id;
    if (p.packed) { uint32_t *r = p.packed + 3 * pos; r[0] = (uint32_t)id; r[1] = (uint32_t)((uint64_t)id >> 32); r[2] = dbits; }
}

// in-LDS bitonic sort of the first n keys (n a power of two, 64 <= n <= SEL_CAP; the rest must already be KEY_INF), ascending.
// The sort is bound by LDS traffic (four workgroups per CU run it at once), so the steps with stride 4, 2 and 1 of
// every merge size -- and the sizes 2, 4, 8 entirely -- run on eight consecutive keys held in registers: one LDS round
// trip for three steps (six for the three smallest sizes); 36 instead of 55 round trips at n = 1024.
template <uint32_t THREADS>
__device__ __forceinline__ void bitonic_sort(uint64_t *keys, int tid, uint32_t n = SEL_CAP) {
    // register phase: thread t owns keys [8t, 8t+8); `first` runs the complete networks of sizes 2, 4, 8, otherwise the
    // strides 4, 2, 1 of merge size `size` (>= 16: all eight keys of a thread then sort in the same direction)
    auto in_registers = [&](uint32_t size, bool first) {
        for (uint32_t t = tid; t < n / 8; t += THREADS) {
            uint64_t v[8];
            const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(keys + 8 * t);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const ulonglong2 w = src[e]; v[2 * e] = w.x; v[2 * e + 1] = w.y; }
            auto cx = [&](int i, int j, bool up) { const uint64_t a = v[i], b = v[j]; const bool sw = (a > b) == up; v[i] = sw ? b : a; v[j] = sw ? a : b; };
            if (first) {
#pragma unroll
                for (int i = 0; i < 8; i += 2) cx(i, i + 1, (i & 2) == 0);                                  // size 2
#pragma unroll
                for (int i = 0; i < 8; ++i) if (!(i & 2)) cx(i, i + 2, (i & 4) == 0);                       // size 4, stride 2
#pragma unroll
                for (int i = 0; i < 8; i += 2) cx(i, i + 1, (i & 4) == 0);                                  // size 4, stride 1
            }
            const bool up = first ? ((8 * t) & 8u) == 0 : ((8 * t) & size) == 0;                            // size 8 / size `size`
#pragma unroll
            for (int i = 0; i < 4; ++i) cx(i, i + 4, up);
#pragma unroll
            for (int i = 0; i < 8; ++i) if (!(i & 2) && !(i & 4) ) { cx(i, i + 2, up); cx(i + 4, i + 6, up); }
#pragma unroll
            for (int i = 0; i < 8; i += 2) cx(i, i + 1, up);
            ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(keys + 8 * t);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = make_ulonglong2(v[2 * e], v[2 * e + 1]);
        }
    };
    __syncthreads();
    in_registers(8, true);
    for (uint32_t size = 16; size <= n; size <<= 1) {
        // Pair t of an LDS step touches elements 2t - (t & (stride-1)) and + stride.  A wave always owns the same 64
        // consecutive pairs, which for stride <= 64 live in one aligned block of 128 elements: consecutive such steps
        // only exchange data inside the wave (LDS operations of a wave execute in order) and need no workgroup barrier.
        bool first_step = true;
        for (uint32_t stride = size >> 1; stride >= 8; stride >>= 1) {
            if (first_step || stride >= 64) __syncthreads();        // after a register phase, or data from other waves
            else __builtin_amdgcn_wave_barrier();
            first_step = false;
            for (uint32_t t = tid; t < n / 2; t += THREADS) {
                const uint32_t lo = 2 * t - (t & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
        }
        __syncthreads();
        in_registers(size, false);
    }
    __syncthreads();
}

// Reservoir scan of columns [0, nb_count): `dists(col, v)` yields the distances of columns col .. col+SEL_COLS-1.
// A round adds at most 1024 keys, so the reservoir is compacted when fewer slots remain.
template <uint32_t THREADS, class Dists>
__device__ __forceinline__ void reservoir_scan(uint64_t *keys, uint32_t &cnt, uint64_t &tau, uint32_t k, size_t nb_first,
                                               size_t nb_count, int tid, Dists &&dists) {
    constexpr int SEL_COLS = SEL_ROUND / THREADS;
    for (size_t base = 0; base < nb_count; base += SEL_ROUND) {
        const uint32_t c = cnt;                               // stable here: a barrier separates it from every add
        __syncthreads();                                      // ... and everyone has read it before the next add
        if (c > SEL_CAP - SEL_ROUND) {                        // workgroup-uniform
            bitonic_sort<THREADS>(keys, tid);
            if (tid == 0) { cnt = c < k ? c : k; tau = c >= k ? keys[k - 1] : KEY_INF; }
            __syncthreads();
            for (uint32_t i = cnt + tid; i < SEL_CAP; i += THREADS) keys[i] = KEY_INF;
            __syncthreads();
        }
        const uint64_t t = tau;
        const size_t col = base + (size_t)tid * SEL_COLS;
        float v[SEL_COLS];
#pragma unroll
        for (int e = 0; e < SEL_COLS; ++e) v[e] = INFINITY;
        if (col < nb_count) dists(col, v);
#pragma unroll
        for (int e = 0; e < SEL_COLS; ++e) {
            if (col + e < nb_count) {
                const uint64_t key = make_key(v[e], (uint32_t)(nb_first + col + e));
                if (key < t) { const uint32_t pos = atomicAdd(&cnt, 1u); keys[pos] = key; }
            }
        }
        __syncthreads();
    }
}

// Bootstrap without sorting the whole chunk: the k-th smallest distance of the slab row is found by radix selection on
// the fp32 bit pattern (distances are >= 0, so the bit patterns order like the values; four passes of a 256-bin LDS
// histogram, most significant byte first), then everything below it and everything equal to it is collected -- the
// caller's final sort of those few keys settles the order and, among equal distances, the smaller ids.  Returns false
// (nothing touched) when the ties at the k-th distance would not fit the reservoir; the reservoir scan handles that.
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
// Bit pattern of the k-th smallest of n non-negative fp32 values held in registers (value e of thread t is element
// t + e * THREADS; elements >= n are ignored): four passes of a 256-bin LDS histogram, most significant byte first.
template <uint32_t THREADS, int VPT>
__device__ __forceinline__ uint32_t radix_kth(const uint32_t (&u)[VPT], uint32_t n, uint32_t k, uint32_t *hist, uint32_t *ctl, int tid) {
    uint32_t prefix = 0, mask = 0, need = k;
    for (int pass = 3; pass >= 0; --pass) {
        for (uint32_t b = tid; b < 256; b += THREADS) hist[b] = 0;
        __syncthreads();
        // the leading bytes of distances are nearly constant (same exponent): a thread adds runs of equal bins in one atomic
        uint32_t run_bin = 0, run_len = 0;
#pragma unroll
        for (int e = 0; e < VPT; ++e) {
            if (tid + e * THREADS < n && (u[e] & mask) == prefix) {
                const uint32_t b = (u[e] >> (8 * pass)) & 255u;
                if (run_len && b != run_bin) { atomicAdd(&hist[run_bin], run_len); run_len = 0; }
                run_bin = b;
                ++run_len;
            }
        }
        if (run_len) atomicAdd(&hist[run_bin], run_len);
        __syncthreads();
        if (tid < 64) {                                              // the first wave finds the bin of the need-th value: a DPP scan over 4 bins per lane
            const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const uint32_t incl = wave_incl_scan(h0 + h1 + h2 + h3);
            const uint64_t hit = __ballot(incl >= need);
            const int L = hit ? __builtin_ctzll(hit) : 63;           // (need <= number of matching values, so some lane qualifies)
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)h0, L), b1 = (uint32_t)__builtin_amdgcn_readlane((int)h1, L),
                           b2 = (uint32_t)__builtin_amdgcn_readlane((int)h2, L), b3 = (uint32_t)__builtin_amdgcn_readlane((int)h3, L);
            uint32_t cum = (uint32_t)__builtin_amdgcn_readlane((int)incl, L) - (b0 + b1 + b2 + b3), bin = 4 * L;
            if (cum + b0 < need) { cum += b0; ++bin;
                if (cum + b1 < need) { cum += b1; ++bin;
                    if (cum + b2 < need) { cum += b2; ++bin; } } }
            if (tid == 0) {
                ctl[0] = prefix | (bin << (8 * pass));
                ctl[1] = need - cum;                                 // rank of the wanted element inside the chosen bin
            }
        }
        __syncthreads();
        prefix = ctl[0]; need = ctl[1]; mask |= 0xFFu << (8 * pass);
        __syncthreads();
    }
    return prefix;
}

// Bootstrap without sorting the whole chunk: the k-th smallest distance of the slab row is found by radix selection on
// the fp32 bit pattern (distances are >= 0, so the bit patterns order like the values), then everything below it and
// everything equal to it is collected -- the caller's final sort of those few keys settles the order and, among equal
// distances, the smaller ids.  Returns false (nothing touched) when the ties at the k-th distance would not fit the
// reservoir; the reservoir scan handles that.
template <uint32_t THREADS>
__device__ __forceinline__ bool radix_bootstrap(uint64_t *keys, uint32_t &cnt, uint32_t *hist, uint32_t *ctl, uint32_t k, const float *row,
                                                size_t nb_first, uint32_t n, int tid) {
    constexpr int VPT = 8192 / THREADS;                              // the chunk (at most 8192 rows) lives in registers: one trip to memory
    uint32_t u[VPT];
#pragma unroll
    for (int e = 0; e < VPT; ++e) {
        const uint32_t col = tid + e * THREADS;
        u[e] = col < n ? __float_as_uint(row[col]) : 0xFFFFFFFFu;    // the filler is above every distance (and above NaN patterns in use)
    }
    const uint32_t prefix = radix_kth<THREADS, VPT>(u, n, k, hist, ctl, tid);
    // prefix = bit pattern of the k-th smallest distance; count what is below / equal
    if (tid == 0) { ctl[2] = 0; ctl[3] = prefix; }
    __syncthreads();
    uint32_t take = 0;
#pragma unroll
    for (int e = 0; e < VPT; ++e) take += (tid + e * THREADS < n) && u[e] <= prefix;
    if (take) atomicAdd(&ctl[2], take);
    __syncthreads();
    if (ctl[2] > SEL_CAP) return false;                              // workgroup-uniform: a plateau of ties wider than the reservoir
#pragma unroll
    for (int e = 0; e < VPT; ++e) {
        const uint32_t col = tid + e * THREADS;
        if (col < n && u[e] <= prefix) { const uint32_t pos = atomicAdd(&cnt, 1u); keys[pos] = ((uint64_t)u[e] << 32) | (uint32_t)(nb_first + col); }
    }
    __syncthreads();
    return true;
}

// Before a merge is sorted: the k-th smallest distance among the n keys in LDS by radix selection, then only the keys at
// or below it (k of them plus ties) move to the front -- the sort that orders them (and settles ties by id) runs on the next
// power of two above k instead of above k + candidates (256 keys instead of 1024 at k = 200: a fifth of the work).
template <uint32_t THREADS>
__device__ __forceinline__ void radix_cut(uint64_t *keys, uint32_t &cnt, uint32_t *hist, uint32_t *ctl, uint32_t k, int tid) {
    constexpr int VPT = SEL_CAP / THREADS;
    const uint32_t n = cnt;                                          // stable: the caller passed a barrier
    uint64_t v[VPT];
    uint32_t u[VPT];
#pragma unroll
    for (int e = 0; e < VPT; ++e) {
        const uint32_t i = tid + e * THREADS;
        v[e] = i < n ? keys[i] : KEY_INF;
        u[e] = (uint32_t)(v[e] >> 32);
    }
    const uint32_t prefix = radix_kth<THREADS, VPT>(u, n, k, hist, ctl, tid);    // barriers inside: every key is in registers by now
    if (tid == 0) cnt = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VPT; ++e)
        if (tid + e * THREADS < n && u[e] <= prefix) keys[atomicAdd(&cnt, 1u)] = v[e];
    __syncthreads();
    const uint32_t m = cnt;
    uint32_t n_sort = 64;
    while (n_sort < m) n_sort <<= 1;
    for (uint32_t i = m + tid; i < n_sort; i += THREADS) keys[i] = KEY_INF;     // the sort's padding
    __syncthreads();
}

// One workgroup per query.  mode 0: scan the chunk's slab.  mode 1: merge the filtered candidates into the
// running top-k, or -- if the candidate list overflowed -- rescan the chunk exactly.
template <uint32_t THREADS>
__device__ __forceinline__ void select_one(const SelArgs &p, const size_t q) {
    constexpr int SEL_COLS = SEL_ROUND / THREADS;
    __shared__ __align__(16) uint64_t keys[SEL_CAP];
    __shared__ uint32_t cnt;
    __shared__ uint64_t tau;
    const int tid = threadIdx.x;
    const uint32_t k = p.k;
    const uint32_t c0 = p.first ? 0u : p.state_cnt[q];
    const uint32_t nc = p.mode == 1 ? p.cand_cnt[q] : 0u;
    const bool merge = p.mode == 1 && nc <= p.cap;            // workgroup-uniform
    // A list that overflowed: this chunk is rescanned exactly (below), and the bf16 tiles -- if they produced it -- are not
    // filtering for this query tile (distances closer together than the operands' rounding resolves): its later chunks take fp32 tiles.
    if (p.mode == 1 && !merge && p.q_flags && threadIdx.x == 0) atomicOr(&p.q_flags[q / 128], 2u);
    for (uint32_t i = tid; i < SEL_CAP; i += THREADS) {
        uint64_t v = KEY_INF;
        if (i < c0) v = p.state[q * k + i];
        else if (merge && i - c0 < nc) v = p.cand[q * p.cap + (i - c0)];     // c0 + nc <= k + cap <= SEL_CAP
        keys[i] = v;
    }
    // (the state of a batch search is unordered between chunks -- merge_wave -- so the running threshold comes from p.tau)
    if (tid == 0) { cnt = merge ? c0 + nc : c0; tau = c0 == k ? make_key(p.tau[q], 0xFFFFFFFFu) : KEY_INF; }
    __syncthreads();
    __shared__ uint32_t hist[256], ctl[20];
    bool done = false;
    if (p.mode == 0 && c0 == 0 && p.nb_count > k && p.nb_count <= 8192)               // first chunk, more rows than results
        done = radix_bootstrap<THREADS>(keys, cnt, hist, ctl, k, p.slab + q * (size_t)p.slab_ld, p.nb_first, (uint32_t)p.nb_count, tid);
    // Batches (256 threads: every later merge is merge_wave or the full sort of the last chunk, neither assumes an ordered state): when exactly k keys came
    // back -- no ties at the k-th distance to cut by id -- the bootstrap's state goes out as it is, unsorted; the sort below was 6 of this kernel's 30 us.
    // (Inexact operands keep it: the density estimate below reads the sorted keys.)
    if (THREADS == 256 && done && !p.last && cnt == k && !(p.q_flags && (!p.base_exact || (p.q_flags[q / 128] & 1u)))) {     // workgroup-uniform
        for (uint32_t i = tid; i < k; i += THREADS) p.state[q * k + i] = keys[i];
        if (tid == 0) { p.state_cnt[q] = k; p.tau[q] = __uint_as_float(ctl[3]); p.cand_cnt[q] = 0; }
        return;
    }
    if (p.mode == 0 && !done) {
        const float *row = p.slab + q * (size_t)p.slab_ld;
        const bool vec = (p.slab_ld & 3) == 0;
        reservoir_scan<THREADS>(keys, cnt, tau, k, p.nb_first, p.nb_count, tid, [&](size_t col, float (&v)[SEL_COLS]) {
            if constexpr (SEL_COLS == 4) {
                if (vec && col + 3 < p.nb_count) {
                    const float4 f = *reinterpret_cast<const float4 *>(row + col);
                    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
                    return;
                }
            }
            for (int e = 0; e < SEL_COLS; ++e) if (col + e < p.nb_count) v[e] = row[col + e];
        });
    } else if (p.mode == 1 && !merge) {
        // overflow: the same k-ordered fmaf chain the matrix pipe evaluates, one base row at a time
        const float *x = p.xq + q * (size_t)p.d;
        const float qn = p.qn[q];
        reservoir_scan<THREADS>(keys, cnt, tau, k, p.nb_first, p.nb_count, tid, [&](size_t col, float (&v)[SEL_COLS]) {
            for (int e = 0; e < SEL_COLS; ++e) {
                if (col + e >= p.nb_count) break;
                const float *y = p.xb + (p.nb_first + col + e) * (size_t)p.d;
                float acc = 0.f;
                for (uint32_t t = 0; t < p.d; ++t) acc = fmaf(x[t], y[t], acc);
                const float dist = fmaf(-2.f, acc, qn + p.bn[p.nb_first + col + e]);
                v[e] = dist < 0.f ? 0.f : dist;
            }
        });
    }
    if (THREADS == 1024 && merge && nc <= 1024) {
        // Few queries (wide workgroups; with 256 threads the 55 sort steps are cheaper).  Merge by enumeration: the state is sorted and keys are unique, so the final position of a key is its rank among
        // the state (its index, or a binary search) plus the number of candidates below it -- counted with broadcast LDS
        // reads, no barrier, no sort.  (c0 + nc) * nc comparisons over the workgroup: a few microseconds at the usual
        // few hundred candidates, against 55 barrier-separated sort steps.
        const uint32_t n = c0 + nc, total = n < k ? n : k;
        for (uint32_t e = tid; e < n; e += THREADS) {
            const uint64_t key = keys[e];
            uint32_t rank = e;
            if (e >= c0) {
                uint32_t lo = 0, hi = c0;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
                rank = lo;
            }
            uint32_t j = 0;
            for (; j + 4 <= nc; j += 4) {
                const uint64_t a = keys[c0 + j], b = keys[c0 + j + 1], c = keys[c0 + j + 2], d = keys[c0 + j + 3];
                rank += (a < key) + (b < key) + (c < key) + (d < key);
            }
            for (; j < nc; ++j) rank += keys[c0 + j] < key;
            if (rank >= k) continue;
            if (p.last) {
                emit_result(p, q * k + rank, true, key);
            } else {
                p.state[q * k + rank] = key;
                if (rank == k - 1) p.tau[q] = __uint_as_float((uint32_t)(key >> 32));
            }
        }
        if (p.last) {
            for (uint32_t i = total + tid; i < k; i += THREADS) emit_result(p, q * k + i, false, 0);
        } else if (tid == 0) {
            p.state_cnt[q] = total;
            if (total < k) p.tau[q] = INFINITY;
            p.cand_cnt[q] = 0;
        }
        return;
    }
    // sort, keep k, carry or emit (a merge usually holds far fewer than SEL_CAP keys: sort only what is there)
    if (merge && cnt > k) {                                     // workgroup-uniform (cnt is stable: a barrier follows every add)
        uint32_t with = 64, without = 64;
        while (with < k) with <<= 1;
        while (without < cnt) without <<= 1;
        if (with < without) radix_cut<THREADS>(keys, cnt, hist, ctl, k, tid);
    }
    uint32_t n_sort = 64;
    while (n_sort < cnt) n_sort <<= 1;                          // cnt is stable: the scan ends with a barrier
    bitonic_sort<THREADS>(keys, tid, n_sort);
    const uint32_t total = cnt < k ? cnt : k;
    if (p.last) {
        for (uint32_t i = tid; i < k; i += THREADS) emit_result(p, q * k + i, i < total, keys[i]);
    } else {
        for (uint32_t i = tid; i < total; i += THREADS) p.state[q * k + i] = keys[i];
        if (tid == 0) {
            p.state_cnt[q] = total;
            p.tau[q] = total == k ? __uint_as_float((uint32_t)(keys[k - 1] >> 32)) : INFINITY;
            p.cand_cnt[q] = 0;
            // Bootstrap, inexact operands: will the bf16 tiles filter for this query?  Their threshold sits a margin m = 1.05 x 2^-8
            // (|x|^2 + |y|^2) above the k-th distance (in inner-product units; 2 m in distance).  The sorted results give the
            // density of base rows there -- k / 2 rows between the distances of rank k / 2 and k, per bootstrap chunk -- and with it
            // the rows the margin lets through on top of the k a chunk is sized for.  More than 1.5 k of them (distances packed far
            // closer than the operands' rounding resolves: e.g. every row at almost the same distance from the query) and the
            // tile takes fp32 tiles from the first chunk on instead of finding out by overflowing a candidate list.
            if (p.mode == 0 && p.first && p.q_flags && total == k && k >= 8 && (!p.base_exact || (p.q_flags[q / 128] & 1u))) {
                const float dk = __uint_as_float((uint32_t)(keys[k - 1] >> 32)), dh = __uint_as_float((uint32_t)(keys[k / 2 - 1] >> 32));
                const float window = 2.f * BF16_MARGIN * (p.qn[q] + p.bn_max);
                if (window * (0.5f * (float)k) > 1.5f * (float)k * (dk - dh)) atomicOr(&p.q_flags[q / 128], 2u);
            }
        }
    }
}

template <uint32_t THREADS>
__global__ void __launch_bounds__(THREADS) k_select(SelArgs p) { select_one<THREADS>(p, blockIdx.x); }

// ---- merge by one WAVE per query (batches) -------------------------------------------------------------------------
// The merge of a chunk's candidates into the running top-k is the step between two tile launches: with a 256-thread
// workgroup per query its ~40 barrier-separated phases (histogram passes, sort steps) cost 19 us per call although the
// work is a few hundred keys -- waves spent 58 % of their cycles parked (PMC).  Here a wave owns a query: the keys sit in
// registers, the k-th smallest KEY (distance, then id: keys are unique, so exactly k survive and no plateau of ties needs a
// special case) is found by radix selection over the bytes that actually differ (wave min / max first), and the survivors
// go back to the state by ballot -- unsorted, no workgroup barrier anywhere.  (A first version also sorted them, in LDS:
// one wave cannot hide the LDS round trip of 36 dependent sort stages and the kernel took 27 us.  Nothing needs the order
// before the last chunk, whose merge sorts in select_one.)
// A workgroup takes four queries; if any of them cannot go this way (candidate list overflowed -> exact rescan, first or
// last chunk) the whole workgroup runs select_one() on its four queries in turn.
constexpr uint32_t MW_VPT = SEL_CAP / 64;
// The k smallest of n unique 64-bit keys held by one wave (slot e of lane l is element e * 64 + l < n; hi_at(e) / lo_at(e) yield
// the distance word and the id word of its key) go to the state of query q, unordered; the k-th distance becomes the query's
// threshold.  All tests run on the 32-bit halves (a pass over distance bytes never touches the ids).
// `opaque`: a register lo_at() may depend on, made opaque once per pass (ids computed from the slot number would otherwise all be
// formed ahead of the pass loop and kept).  (Tried for the 8192-row bootstrap as well, 128 computed keys per lane: 40 us against
// the workgroup version's 32 -- one wave serialises on the few histogram bins the leading bytes fall into.)
template <uint32_t VPT, class HiAt, class LoAt>
__device__ __forceinline__ void wave_keep_k_smallest(const SelArgs &p, const size_t q, uint32_t n, uint32_t *hist, int lane, uint32_t &opaque,
                                                     HiAt &&hi_at, LoAt &&lo_at) {
    const uint32_t k = p.k;
    uint32_t Thi = (uint32_t)((KEY_INF - 1) >> 32), Tlo = (uint32_t)(KEY_INF - 1);     // keep every real key when there are no more than k
    if (n > k) {
        // bits in which the keys differ at all: OR over (key ^ one of the keys); the bytes above the first of them are common
        const uint32_t h_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi_at(0)), l_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo_at(0));
        uint32_t dl = 0, dh = 0;                                    // (lane 0 holds a real key: n > k >= 1)
#pragma unroll
        for (uint32_t e = 0; e < VPT; ++e)
            if (e * 64 < n && e * 64 + lane < n) { dh |= hi_at(e) ^ h_first; dl |= lo_at(e) ^ l_first; }
        dh = __reduce_or_sync(~0ull, dh); dl = __reduce_or_sync(~0ull, dl);            // not both zero: keys are unique
        int shift = dh ? 32 + ((31 - __builtin_clz(dh)) / 8) * 8 : ((31 - __builtin_clz(dl | 1u)) / 8) * 8;
        // prefix / mask of the bytes already fixed, as (distance word, id word)
        uint32_t p_hi, m_hi, p_lo = 0, m_lo = 0;
        if (shift >= 32) { const int s2 = shift - 32; m_hi = s2 >= 24 ? 0u : ~0u << (s2 + 8); p_hi = h_first & m_hi; }
        else { m_hi = ~0u; p_hi = h_first; m_lo = shift >= 24 ? 0u : ~0u << (shift + 8); p_lo = l_first & m_lo; }
        uint32_t need = k;
        for (;; shift -= 8) {
            asm volatile("" : "+v"(opaque));
#pragma unroll
            for (int b = 0; b < 4; ++b) hist[4 * lane + b] = 0;
            wave_sync();
            if (shift >= 32) {                                      // wave-uniform: a byte of the distance
                const int s2 = shift - 32;
#pragma unroll
                for (uint32_t e = 0; e < VPT; ++e)
                    if (e * 64 < n && e * 64 + lane < n) { const uint32_t h = hi_at(e); if ((h & m_hi) == p_hi) atomicAdd(&hist[(h >> s2) & 255u], 1u); }
            } else {                                                // a byte of the id: only among keys of the k-th distance
#pragma unroll
                for (uint32_t e = 0; e < VPT; ++e)
                    if (e * 64 < n && e * 64 + lane < n && hi_at(e) == p_hi) { const uint32_t l = lo_at(e); if ((l & m_lo) == p_lo) atomicAdd(&hist[(l >> shift) & 255u], 1u); }
            }
            wave_sync();
            const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const uint32_t incl = wave_incl_scan(h0 + h1 + h2 + h3);
            const int L = __builtin_ctzll(__ballot(incl >= need));          // the lane whose four bins hold the need-th key (wave-uniform)
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)h0, L), b1 = (uint32_t)__builtin_amdgcn_readlane((int)h1, L),
                           b2 = (uint32_t)__builtin_amdgcn_readlane((int)h2, L), b3 = (uint32_t)__builtin_amdgcn_readlane((int)h3, L);
            uint32_t cum = (uint32_t)__builtin_amdgcn_readlane((int)incl, L) - (b0 + b1 + b2 + b3);
            uint32_t bin = 4 * L, cnt_bin = b0;
            if (cum + b0 < need) { cum += b0; ++bin; cnt_bin = b1;
                if (cum + b1 < need) { cum += b1; ++bin; cnt_bin = b2;
                    if (cum + b2 < need) { cum += b2; ++bin; cnt_bin = b3; } } }
            need -= cum;
            if (shift >= 32) { p_hi |= bin << (shift - 32); m_hi |= 0xFFu << (shift - 32); }
            else { p_lo |= bin << shift; m_lo |= 0xFFu << shift; }
            if (cnt_bin == need || shift == 0) {                   // the whole bin is wanted (always so at the last byte: keys are unique)
                Thi = p_hi | ~m_hi; Tlo = p_lo | ~m_lo;
                break;
            }
        }
    }
    asm volatile("" : "+v"(opaque));
    // compaction: keys <= T go to the state, by ballot -- UNSORTED (nothing between two chunks needs the order: the next
    // merge selects again, the tile kernel only wants the k-th distance; the last chunk's merge sorts, in select_one)
    uint32_t total = 0, dmax = 0;
#pragma unroll
    for (uint32_t e = 0; e < VPT; ++e) {
        if (e * 64 < n) {
            const uint32_t h = hi_at(e), l = lo_at(e);
            const bool keep = e * 64 + lane < n && (h < Thi || (h == Thi && l <= Tlo));
            const uint64_t m = __ballot(keep);
            const uint32_t pos = total + (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (keep && pos < k) {
                p.state[q * k + pos] = ((uint64_t)h << 32) | l;
                dmax = h > dmax ? h : dmax;
            }
            total += (uint32_t)__popcll(m);
        }
    }
    total = total < k ? total : k;                                  // (unique keys: exactly min(n, k))
    // the k-th distance: largest kept one (six ds_bpermute steps on one word, once)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(dmax, d); dmax = o > dmax ? o : dmax; }
    if (lane == 0) {
        p.state_cnt[q] = total;
        p.tau[q] = total == k ? __uint_as_float(dmax) : INFINITY;   // distances are >= 0: their bit patterns order like the values
        p.cand_cnt[q] = 0;
    }
}

// merge of a chunk's candidates into the running state
__device__ __forceinline__ void merge_wave(const SelArgs &p, const size_t q, uint32_t *hist, int lane) {
    const uint32_t k = p.k, c0 = p.state_cnt[q], nc = p.cand_cnt[q], n = c0 + nc;
    uint64_t v[MW_VPT];
#pragma unroll
    for (uint32_t e = 0; e < MW_VPT; ++e) {
        v[e] = KEY_INF;
        if (e * 64 < n) {                                          // wave-uniform
            const uint32_t i = e * 64 + lane;
            if (i < c0) v[e] = p.state[q * k + i];
            else if (i < n) v[e] = p.cand[q * p.cap + (i - c0)];
        }
    }
    uint32_t unused = 0;
    wave_keep_k_smallest<MW_VPT>(p, q, n, hist, lane, unused, [&](uint32_t e) { return (uint32_t)(v[e] >> 32); }, [&](uint32_t e) { return (uint32_t)v[e]; });
}

__global__ void __launch_bounds__(256) k_merge4(SelArgs p, uint32_t nq) {
    __shared__ uint32_t hist[4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t q0 = (size_t)blockIdx.x * 4;
    bool fast = p.mode == 1 && !p.first && !p.last;
    for (uint32_t j = 0; j < 4 && fast; ++j)
        if (q0 + j < nq) fast = p.cand_cnt[q0 + j] <= p.cap && p.state_cnt[q0 + j] + p.cand_cnt[q0 + j] <= SEL_CAP;      // workgroup-uniform
    if (fast) {
        if (q0 + wave < nq) merge_wave(p, q0 + wave, hist[wave], lane);
        return;
    }
    for (uint32_t j = 0; j < 4; ++j) {
        if (q0 + j < nq) select_one<256>(p, q0 + j);               // workgroup-uniform
        __syncthreads();
    }
}

// ---- Server::preciseSearch: exact gathered distances ---------------------------------------------
// float dist = 0; dist += std::pow(row[k] - q[k], 2)  ==  dist = (float)((double)dist + (double)diff*(double)diff)
// with diff an fp32 subtraction (server_lib.cpp:151-162).  One lane per (query, candidate); the chain is
// inherently sequential, the row read is 512 contiguous bytes per lane.
__global__ void __launch_bounds__(256) k_l2_gathered(const float *__restrict__ xb, size_t nb, uint32_t d, const float *__restrict__ xq,
                                                      const int64_t *__restrict__ ids, size_t nq, uint32_t c, float *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nq * c) return;
    const size_t qi = t / c;
    const int64_t id = ids[t];
    if (id < 0 || (size_t)id >= nb) { out[t] = INFINITY; return; }
    const float *row = xb + (size_t)id * d, *qv = xq + qi * d;
    float dist = 0.f;
    for (uint32_t k = 0; k < d; ++k) {
        const float diff = row[k] - qv[k];
        dist = (float)((double)dist + (double)diff * (double)diff);
    }
    out[t] = dist;
}

// out[i][:] = xb[ids[i]][:]; one wave per row, 16 B per lane where the row allows
__global__ void __launch_bounds__(256) k_gather_rows(const float *__restrict__ xb, size_t nb, uint32_t d, const int64_t *__restrict__ ids,
                                                      size_t n_ids, float *__restrict__ out) {
    const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_ids) return;
    const int lane = threadIdx.x & 63;
    const int64_t id = ids[r];
    const bool ok = id >= 0 && (size_t)id < nb;
    const float *src = xb + (ok ? (size_t)id : 0) * d;
    float *dst = out + r * d;
    if ((d & 3) == 0) {
        for (uint32_t k = lane * 4; k < d; k += 256) {
            float4 v = ok ? *reinterpret_cast<const float4 *>(src + k) : make_float4(NAN, NAN, NAN, NAN);
            *reinterpret_cast<float4 *>(dst + k) = v;
        }
    } else {
        for (uint32_t k = lane; k < d; k += 64) dst[k] = ok ? src[k] : NAN;
    }
}

}  // namespace pf

using namespace pf;

struct pf_flat {
    int device = 0;
    size_t nb = 0;
    uint32_t d = 0;
    float *xb = nullptr, *bn = nullptr;
    uint16_t *xb16 = nullptr;     // bf16 image of the base matrix (d = 64 or 128): rows of d values + AUX16 threshold words, nearest-even
    bool exact16 = false;         // EVERY value of the base passed the on-device exactness check: the image is the matrix itself
    float bn_max = 0.f;           // largest row norm (the margin of the bf16 tiles as a filter over inexact operands)
    bool use16 = true;            // pf_flat_exact16: the caller may switch the 16-bit operand path off
    int8_t *xb8 = nullptr;        // 8-bit data only (every value an integer in [0, 255]; d a multiple of 32 up to 128): rows of d values - 128 + AUX8 threshold bytes
    bool use8 = true;             // pf_flat_operands8: the caller may switch the int8 tiles off (the bf16 tiles then run on the same data)
    // workspace (grown outside graph capture)
    void *ws = nullptr;
    size_t ws_bytes = 0;
    size_t wg_slots = 1024;   // workgroups of the tile kernel resident on the device at once (CUs x occupancy)
    size_t num_cus = 256;
    // tuning knobs, read from the environment ONCE when the index is created (experiments; never on the search path)
    size_t b16_min_nq = 1;    // PF_FLAT_B16_MIN_NQ: smallest batch that takes the bf16 tiles
    size_t group_cap = 64;    // PF_FLAT_GROUP_CAP: most column tiles one workgroup of k_l2_tile16 walks
    double growth_div = 0.0;  // PF_FLAT_GROWTH_DIV: survivors per chunk as a fraction of the candidate capacity (0: the defaults)
};

namespace {

#ifndef PF_BOOT_ROWS
#define PF_BOOT_ROWS 8192
#endif
constexpr size_t BOOT_ROWS = PF_BOOT_ROWS;  // bootstrap chunk (slab path); at most 8192 (radix_bootstrap keeps the chunk in registers)

struct WsPlan { size_t boot, slab_ld, cap, off_qn, off_tau, off_cnt, off_scnt, off_state, off_cand, off_slab, off_q16, off_q8, off_qbad, total; };

WsPlan plan_ws(size_t nb, size_t nq, uint32_t k, uint32_t d) {
    WsPlan w{};
    const size_t nb_pad = (nb + 127) / 128 * 128;
    w.boot = BOOT_ROWS < nb_pad ? BOOT_ROWS : (nb_pad ? nb_pad : 128);
    w.slab_ld = w.boot;
    w.cap = SEL_CAP - k;                    // state (<= k keys) + candidates fit one sort
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t o = 0;
    w.off_qn = o; o += up(nq * 4);
    w.off_tau = o; o += up(nq * 4);
    w.off_cnt = o; o += up(nq * 4);
    w.off_scnt = o; o += up(nq * 4);
    w.off_state = o; o += up(nq * (size_t)k * 8);
    w.off_cand = o; o += up(nq * w.cap * 8);
    w.off_slab = o; o += up(nq * w.slab_ld * 4);
    w.off_q16 = o; o += up(nq * (size_t)d * 2);
    w.off_q8 = o; o += up(nq * (size_t)d);
    w.off_qbad = o; o += up(((nq + 127) / 128) * 4);
    w.total = o;
    return w;
}

pf_status ensure_ws(pf_flat *f, size_t bytes) {
    if (bytes <= f->ws_bytes) return PF_OK;
    if (f->ws) { PF_HIP(hipFree(f->ws)); f->ws = nullptr; f->ws_bytes = 0; }
    PF_HIP(hipMalloc(&f->ws, bytes));
    f->ws_bytes = bytes;
    return PF_OK;
}

}  // namespace

namespace pf {
const float *flat_base_device(const pf_flat *f, size_t *nb, uint32_t *d, int *device) {
    if (!f) return nullptr;
    if (nb) *nb = f->nb;
    if (d) *d = f->d;
    if (device) *device = f->device;
    return f->xb;
}
}  // namespace pf

extern "C" {

pf_status pf_flat_destroy(pf_flat *f) {
    if (!f) return PF_OK;
    {
        DeviceGuard g(f->device);
        if (f->xb) (void)hipFree(f->xb);
        if (f->xb16) (void)hipFree(f->xb16);
        if (f->xb8) (void)hipFree(f->xb8);
        if (f->bn) (void)hipFree(f->bn);
        if (f->ws) (void)hipFree(f->ws);
    }
    delete f;
    return PF_OK;
}

pf_status pf_flat_create(pf_flat **out, int device, const float *xb, size_t nb, uint32_t d) {
    if (!out || (!xb && nb) || d == 0) return fail(PF_ERR_INVALID_ARG, "null argument or d == 0");
    *out = nullptr;
    if (nb >= (1ull << 32) - 1) return fail(PF_ERR_UNSUPPORTED, "nb must be below 2^32-1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(PF_ERR_NO_DEVICE, "no such HIP device");
    PF_GUARD(device);
    pf_flat *f = new pf_flat;
    f->device = device; f->nb = nb; f->d = d;
    if (const char *v = getenv("PF_FLAT_B16_MIN_NQ")) f->b16_min_nq = (size_t)atoi(v);
    if (const char *v = getenv("PF_FLAT_GROUP_CAP")) { if (atoi(v) > 0) f->group_cap = (size_t)atoi(v); }
    if (const char *v = getenv("PF_FLAT_GROWTH_DIV")) f->growth_div = atof(v);
    const size_t bytes = (nb ? nb : 1) * (size_t)d * 4;
    hipError_t e = hipMalloc((void **)&f->xb, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&f->bn, (nb ? nb : 1) * 4);
    if (e == hipSuccess && nb) e = hipMemcpy(f->xb, xb, nb * (size_t)d * 4, hipMemcpyDefault);
    if (e == hipSuccess && nb) {
        // row norms; and, where the shape allows the bf16 loop, the 16-bit image with its value-by-value exactness check
        uint32_t *flag = nullptr;
        // every row length that is a multiple of the matrix instruction's k-step up to 256 (k_rows_prep stages whole rows in LDS
        // up to PREP_MAX_D; beyond 128 values the tiles are 128 x 64: Geo16W)
        const bool try16 = d % 16 == 0 && d <= 256 && getenv("PF_FLAT_NO_BF16") == nullptr;
        // image rows carry AUX16 threshold words behind their d values; one tile of zero rows pads the end (k_l2_tile16 copies whole tiles)
        const size_t bytes16 = (nb + 128) * (size_t)(d + AUX16) * 2;
        if (try16 && (hipMalloc((void **)&f->xb16, bytes16) != hipSuccess || hipMemset(f->xb16, 0, bytes16) != hipSuccess ||
                      hipMalloc((void **)&flag, 4) != hipSuccess || hipMemset(flag, 0, 4) != hipSuccess)) {
            (void)hipGetLastError();                                  // no room for the image: the fp32 path needs none
            if (f->xb16) { (void)hipFree(f->xb16); f->xb16 = nullptr; }
            if (flag) { (void)hipFree(flag); flag = nullptr; }
        }
        // 8-bit data (SIFT, the reference's dataset): an int8 image for the integer matrix instruction where the rows are whole 32-deep k-steps
        // (same zero-row padding; rows of d values - 128 + AUX8 bytes).  Kept only if EVERY value is an integer in [0, 255] (flag bit 2).
        const size_t bytes8 = (nb + 128) * (size_t)(d + AUX8);
        if (f->xb16 && d % 32 == 0 && d <= 128 && getenv("PF_FLAT_NO_I8") == nullptr &&
            (hipMalloc((void **)&f->xb8, bytes8) != hipSuccess || hipMemset(f->xb8, 0, bytes8) != hipSuccess)) {
            (void)hipGetLastError();
            if (f->xb8) { (void)hipFree(f->xb8); f->xb8 = nullptr; }
        }
        if (d <= 128) hipLaunchKernelGGL((k_rows_prep<64, 128>), dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, nullptr, f->xb, nb, d, f->bn, f->xb16, d + AUX16, true, flag, 0u,
                                         f->xb8, d + (uint32_t)AUX8);
        else if (d <= PREP_MAX_D) hipLaunchKernelGGL((k_rows_prep<32, PREP_MAX_D>), dim3((unsigned)((nb + 31) / 32)), dim3(64), 0, nullptr, f->xb, nb, d, f->bn, f->xb16, d + AUX16, true, flag, 0u);
        else hipLaunchKernelGGL(k_row_norms, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, nullptr, f->xb, nb, d, f->bn);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (flag) {
            uint32_t inexact = 1;
            if (e == hipSuccess) e = hipMemcpy(&inexact, flag, 4, hipMemcpyDeviceToHost);
            (void)hipFree(flag);
            f->exact16 = f->xb16 && !(inexact & 1u);       // inexact values: the image stays, as the operand of a conservative filter
            if (f->xb8 && (inexact & 5u)) { (void)hipFree(f->xb8); f->xb8 = nullptr; }     // some value is not an integer in [0, 255]
            if (e == hipSuccess && f->xb16 && !f->exact16) {
                hipLaunchKernelGGL(k_aux_margin, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, nullptr, f->xb16, f->bn, nb, d, d + AUX16);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipDeviceSynchronize();
            }
        }
        if (e == hipSuccess && f->xb16) {            // largest row norm; a norm that is not finite rules the filter out
            std::vector<float> norms(nb);
            e = hipMemcpy(norms.data(), f->bn, nb * 4, hipMemcpyDeviceToHost);
            bool finite = true;
            for (float v : norms) { finite = finite && std::isfinite(v); if (v > f->bn_max) f->bn_max = v; }
            if (!finite) { (void)hipFree(f->xb16); f->xb16 = nullptr; f->exact16 = false; if (f->xb8) { (void)hipFree(f->xb8); f->xb8 = nullptr; } }
        }
    }
    if (e != hipSuccess) { pf_flat_destroy(f); return fail(e == hipErrorOutOfMemory ? PF_ERR_OOM : PF_ERR_HIP, std::string("pf_flat_create: ") + hipGetErrorString(e)); }
    {
        hipDeviceProp_t prop{};
        int occ = 0;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_l2_tile<true, GeoBatch, true>, 256, 0) == hipSuccess && occ > 0)
            f->wg_slots = (size_t)prop.multiProcessorCount * (size_t)occ;
        if (prop.multiProcessorCount > 0) f->num_cus = (size_t)prop.multiProcessorCount;
        (void)hipGetLastError();
    }
    *out = f;
    return PF_OK;
}

pf_status pf_flat_exact16(pf_flat *f, int mode, int *active) {
    if (!f || mode < -1 || mode > 1) return fail(PF_ERR_INVALID_ARG, "pf_flat_exact16: index, and mode -1 (query), 0 (off) or 1 (on where exact)");
    if (mode >= 0) f->use16 = mode == 1;
    if (active) *active = f->xb16 && f->use16 ? (f->exact16 ? 2 : 1) : 0;
    return PF_OK;
}

pf_status pf_flat_operands8(pf_flat *f, int mode, int *active) {
    if (!f || mode < -1 || mode > 1) return fail(PF_ERR_INVALID_ARG, "pf_flat_operands8: index, and mode -1 (query), 0 (off) or 1 (on where the data is 8-bit)");
    if (mode >= 0) f->use8 = mode == 1;
    if (active) *active = f->xb8 && f->use8 && f->xb16 && f->use16 ? 1 : 0;
    return PF_OK;
}

pf_status pf_flat_info(const pf_flat *f, size_t *nb, uint32_t *d) {
    if (!f) return fail(PF_ERR_INVALID_ARG, "null index");
    if (nb) *nb = f->nb;
    if (d) *d = f->d;
    return PF_OK;
}

pf_status pf_flat_reserve(pf_flat *f, size_t nq_max, uint32_t k_max) {
    if (!f || nq_max == 0 || k_max == 0) return fail(PF_ERR_INVALID_ARG, "bad argument");
    PF_GUARD(f->device);
    return ensure_ws(f, plan_ws(f->nb, nq_max, k_max, f->d).total);
}

pf_status pf_flat_search(pf_flat *f, const float *xq, size_t nq, uint32_t k, float *D, int64_t *I, pf_stream stream) {
    if (nq && (!D || !I)) return fail(PF_ERR_INVALID_ARG, "null argument");
    return pf_flat_search_packed(f, xq, nq, k, D, I, nullptr, stream);
}

pf_status pf_flat_search_packed(pf_flat *f, const float *xq, size_t nq, uint32_t k, float *D, int64_t *I, uint32_t *packed, pf_stream stream) {
    if (!f) return fail(PF_ERR_INVALID_ARG, "null index");
    if (nq == 0) return PF_OK;
    if (!xq || (!packed && (!D || !I))) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (k == 0 || k > K_MAX) return fail(PF_ERR_UNSUPPORTED, "k must be in [1, 1024]");
    if (nq > (1u << 20)) return fail(PF_ERR_INVALID_ARG, "nq too large for one call (at most 2^20 queries)");
    PF_GUARD(f->device);
    hipStream_t s = as_stream(stream);
    const WsPlan w = plan_ws(f->nb, nq, k, f->d);
    pf_status st = ensure_ws(f, w.total);
    if (st != PF_OK) return st;
    char *base = static_cast<char *>(f->ws);
    float *qn = reinterpret_cast<float *>(base + w.off_qn);
    float *tau = reinterpret_cast<float *>(base + w.off_tau);
    uint32_t *ccnt = reinterpret_cast<uint32_t *>(base + w.off_cnt);
    uint32_t *scnt = reinterpret_cast<uint32_t *>(base + w.off_scnt);
    uint64_t *state = reinterpret_cast<uint64_t *>(base + w.off_state);
    uint64_t *cand = reinterpret_cast<uint64_t *>(base + w.off_cand);
    float *slab = reinterpret_cast<float *>(base + w.off_slab);
    // Any batch size: for a few queries the 128-row tiles are mostly padding, but the scan is then bound by the bytes of the
    // base it streams, and the bf16 image is half the fp32 matrix (1 query over 1M x 128: 0.22 -> 0.16 ms, 64 queries 0.36 -> 0.23)
    const bool b16 = f->xb16 && f->use16 && nq >= f->b16_min_nq;
    uint16_t *q16 = reinterpret_cast<uint16_t *>(base + w.off_q16);
    uint32_t *qbad = reinterpret_cast<uint32_t *>(base + w.off_qbad);
    if (b16) PF_HIP(hipMemsetAsync(qbad, 0, ((nq + 127) / 128) * 4, s));
    const bool b8 = b16 && f->xb8 && f->use8;                       // int8 tiles for the query tiles that turn out to be 8-bit too (flag bit 2, set by the kernel below)
    int8_t *q8 = reinterpret_cast<int8_t *>(base + w.off_q8);
    if (f->d <= PREP_MAX_D) hipLaunchKernelGGL((k_rows_prep<4, PREP_MAX_D>), dim3((unsigned)((nq + 3) / 4)), dim3(64), 0, s, xq, nq, f->d, qn, b16 ? q16 : nullptr, f->d,
                                               false, b16 ? qbad : nullptr, 128u, b8 ? q8 : nullptr, f->d);
    else hipLaunchKernelGGL(k_row_norms, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, xq, nq, f->d, qn);
    TileArgs t{};
    t.xq16 = q16; t.xb16 = f->xb16; t.q_inexact = qbad; t.base_exact = f->exact16 ? 1u : 0u; t.bn_max = f->bn_max;
    t.xq8 = b8 ? q8 : nullptr; t.xb8 = b8 ? f->xb8 : nullptr;
    t.xq = xq; t.xb = f->xb; t.qn = qn; t.bn = f->bn; t.slab = slab; t.nq = (uint32_t)nq; t.d = f->d; t.slab_ld = (uint32_t)w.slab_ld;
    t.tau = tau; t.cand_cnt = ccnt; t.cand = cand; t.cap = (uint32_t)w.cap;
    SelArgs a{};
    a.slab = slab; a.slab_ld = (uint32_t)w.slab_ld; a.state = state; a.state_cnt = scnt; a.tau = tau; a.cand_cnt = ccnt; a.cand = cand;
    a.cap = (uint32_t)w.cap; a.xq = xq; a.xb = f->xb; a.qn = qn; a.bn = f->bn; a.d = f->d; a.k = k; a.D = D; a.I = I; a.packed = packed; a.q_flags = b16 ? qbad : nullptr; a.bn_max = f->bn_max; a.base_exact = f->exact16 ? 1u : 0u;
    // tile geometry by batch size: 128-row query tiles for batches, 32 / 64-row tiles when a 128-row tile would be
    // mostly padding (the scan of the base is then HBM-bound instead of MFMA-bound)
    const int geo = b16 ? 2 : nq <= 32 ? 0 : nq <= 64 ? 1 : 2;
    const bool wide16 = b16 && f->d > 128;                                             // Geo16W
    const size_t TM = geo == 0 ? 32 : geo == 1 ? 64 : 128, TN = b16 ? (size_t)(wide16 ? Geo16W::TN : Geo16::TN) : geo == 2 ? 128 : 256;
    const size_t slots = b16 ? f->num_cus * (size_t)(wide16 ? 2 : B16_WG_PER_CU) : f->wg_slots;       // workgroups of the tile kernel resident at once
    t.n_qtiles = (uint32_t)((nq + TM - 1) / TM);
    auto launch_tile = [&](bool filter, size_t cols) {
        const size_t nct = (cols + TN - 1) / TN;
        if (b16) {
            // Column tiles per workgroup: the launch should take the fewest whole rounds of resident workgroups (2 per CU) that walks of
            // at most group_cap tiles allow, and fill them -- a walk pays a prologue and a flush (about three tiles' worth), and a
            // round that is a quarter full takes as long as a full one.  (Before: two rounds whatever the chunk and walks of at most 8
            // tiles; 16 k columns ran as 2 x 1 tile, 213 k as 3.25 rounds of 8.  Measured over the cap: 8 0.555, 16 0.537, 32 0.520,
            // 64 0.514, 128 0.510 ms per search -- most chunks are then one round of workgroups that walk their whole share.)
            const size_t group_cap = f->group_cap;
            const size_t per_round = slots / t.n_qtiles ? slots / t.n_qtiles : 1;       // column groups of one round
            const size_t rounds = (nct + per_round * group_cap - 1) / (per_round * group_cap);
            size_t group = (nct + per_round * rounds - 1) / (per_round * rounds);
            group = group < 1 ? 1 : group;
            const size_t n_groups = (nct + group - 1) / group;
            const dim3 grid16((unsigned)(((n_groups + 7) / 8) * 8 * t.n_qtiles));
            const uint32_t g32 = (uint32_t)group, n32 = (uint32_t)n_groups;
            switch (f->d) {
#define PF_T16(DD) case DD: if (filter) hipLaunchKernelGGL((k_l2_tile16<true, DD>), grid16, dim3(256), 0, s, t, g32, n32); \
                            else hipLaunchKernelGGL((k_l2_tile16<false, DD>), grid16, dim3(256), 0, s, t, g32, n32); break;
                PF_T16(16) PF_T16(32) PF_T16(48) PF_T16(64) PF_T16(80) PF_T16(96) PF_T16(112) PF_T16(128)
                PF_T16(144) PF_T16(160) PF_T16(176) PF_T16(192) PF_T16(208) PF_T16(224) PF_T16(240) PF_T16(256)
#undef PF_T16
                default: break;                                       // (pf_flat_create keeps an image for these row lengths only)
            }
            return;
        }
        const dim3 grid((unsigned)(((nct + 7) / 8) * 8 * t.n_qtiles));
        const bool fast = f->d % TK == 0;
#define PF_TILE(FILTER, GEO, AGG) do { if (fast) hipLaunchKernelGGL((k_l2_tile<FILTER, GEO, true, AGG>), grid, dim3(256), 0, s, t); \
                                       else hipLaunchKernelGGL((k_l2_tile<FILTER, GEO, false, AGG>), grid, dim3(256), 0, s, t); } while (0)
        // per-workgroup aggregation of survivors pays once several queries share the hot counters (measured: nq = 1 0.27 ->
        // 0.25 ms without it, nq = 32 0.30 -> 0.33 ms without it)
        const bool agg = nq > 4;
        switch (geo * 2 + (filter ? 1 : 0)) {
            case 0: PF_TILE(false, GeoSmall32, false); break;
            case 1: if (agg) PF_TILE(true, GeoSmall32, true); else PF_TILE(true, GeoSmall32, false); break;
            case 2: PF_TILE(false, GeoSmall64, false); break;
            case 3: PF_TILE(true, GeoSmall64, true); break;
            case 4: PF_TILE(false, GeoBatch, false); break;
            default: PF_TILE(true, GeoBatch, false); break;
        }
#undef PF_TILE
    };
    auto launch_select = [&]() {
        if (nq <= 256) hipLaunchKernelGGL(k_select<1024>, dim3((unsigned)nq), dim3(1024), 0, s, a);
        else if (a.mode == 1 && !a.first && !a.last) hipLaunchKernelGGL(k_merge4, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, a, (uint32_t)nq);   // a wave per query
        else hipLaunchKernelGGL(k_select<256>, dim3((unsigned)nq), dim3(256), 0, s, a);
    };
    // bootstrap chunk through the slab
    const size_t boot = f->nb < w.boot ? f->nb : w.boot;
    t.nb_first = 0; t.nb_count = boot;
    if (boot) launch_tile(false, boot);
    a.nb_first = 0; a.nb_count = boot; a.mode = 0; a.first = 1; a.last = boot == f->nb;
    launch_select();
    // streaming chunks: sized so that the expected survivors per query, k * chunk / rows_seen, stay at a fraction 1/div of
    // the candidate capacity
    const double growth_div = f->growth_div;
    size_t pos = boot;
    // Batches: rows_seen may grow by at most g_max = 1 + cap / (div * k) per chunk.  Taking g_max every time ends in a short
    // last chunk that still costs a launch and a merge; instead the number of chunks n is the smallest that g_max allows and
    // all of them grow by the same ratio (nb / boot)^(1/n) -- 1M rows after 8192: four chunks of ratio 3.3 at div = 3 carry
    // the survivors per chunk that five greedy ones at div = 4 carried.
    double ratio = 0.0;
    int n_chunks = 0, chunk_no = 0;
    if (geo == 2 && f->nb > boot) {
        const double div = growth_div > 0.0 ? growth_div : 3.0, g_max = 1.0 + (double)w.cap / (div * (double)k), span = (double)f->nb / (double)boot;
        n_chunks = (int)ceil(log(span) / log(g_max) - 1e-9);
        if (n_chunks < 1) n_chunks = 1;
        ratio = pow(span, 1.0 / n_chunks);
    }
    while (pos < f->nb) {
        size_t chunk;
        if (geo == 2) {
            ++chunk_no;
            const double end = (double)boot * pow(ratio, chunk_no);
            chunk = chunk_no >= n_chunks || end >= (double)f->nb ? f->nb - pos : ((size_t)end - pos) / 1024 * 1024;
            if (chunk < 4096) chunk = 4096;
            // whole rounds of resident workgroups: the short early chunks take as long as their rounds, however full the last one is
            const size_t round_cols = (slots / t.n_qtiles ? slots / t.n_qtiles : 1) * TN;
            if (chunk > round_cols && chunk < f->nb - pos) chunk = (chunk + round_cols / 2) / round_cols * round_cols;
            if (chunk > f->nb - pos || f->nb - pos - chunk < 4096) chunk = f->nb - pos;
        } else {
            const double div = growth_div > 0.0 ? growth_div : 4.0;
            chunk = (size_t)((double)pos * (double)w.cap / (div * (double)k));
            chunk = chunk / 256 * 256;
            if (chunk < 4096) chunk = 4096;
            // whole rounds of resident workgroups: a chunk that fills the device 1.2 times takes as long as one that fills it twice
            const size_t round_cols = (f->wg_slots / t.n_qtiles ? f->wg_slots / t.n_qtiles : 1) * TN;
            if (chunk > round_cols) chunk = chunk / round_cols * round_cols;
            if (chunk > f->nb - pos) chunk = f->nb - pos;
            // a short tail is not worth a launch and a merge of its own (the 4x margin on the candidate capacity absorbs it)
            if (f->nb - pos - chunk < chunk / 2) chunk = f->nb - pos;
        }
        t.nb_first = pos; t.nb_count = chunk;
        launch_tile(true, chunk);
        a.nb_first = pos; a.nb_count = chunk; a.mode = 1; a.first = 0; a.last = pos + chunk == f->nb;
        launch_select();
        pos += chunk;
    }
    PF_HIP(hipGetLastError());
    return PF_OK;
}

pf_status pf_l2_gathered(pf_flat *f, const float *xq, const int64_t *ids, size_t nq, uint32_t c, float *D, pf_stream stream) {
    if (!f) return fail(PF_ERR_INVALID_ARG, "null index");
    if (nq == 0 || c == 0) return PF_OK;
    if (!xq || !ids || !D) return fail(PF_ERR_INVALID_ARG, "null argument");
    PF_GUARD(f->device);
    const size_t total = nq * (size_t)c;
    hipLaunchKernelGGL(k_l2_gathered, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), f->xb, f->nb, f->d, xq, ids, nq, c, D);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

pf_status pf_gather_rows(pf_flat *f, const int64_t *ids, size_t n_ids, float *out, pf_stream stream) {
    if (!f) return fail(PF_ERR_INVALID_ARG, "null index");
    if (n_ids == 0) return PF_OK;
    if (!ids || !out) return fail(PF_ERR_INVALID_ARG, "null argument");
    PF_GUARD(f->device);
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n_ids + 3) / 4)), dim3(256), 0, as_stream(stream), f->xb, f->nb, f->d, ids, n_ids, out);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

}  // extern "C"

#ifdef PF_FLAT_STAMPS
extern "C" int pf_flat_debug_stamps(unsigned long long *out, size_t n) {
    const size_t have = sizeof(pf::pf_flat_stamp_buf) / 8;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::pf_flat_stamp_buf), (n < have ? n : have) * 8);
}
extern "C" int pf_flat_debug_flush_stamps(unsigned long long *out, size_t n) {
    const size_t have = sizeof(pf::pf_flat_flush_stamp_buf) / 8;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::pf_flat_flush_stamp_buf), (n < have ? n : have) * 8);
}
#endif
