// flat_tile16.hpp -- bf16 / int8 tile walk on the matrix pipe with the threshold inside the accumulators (k_l2_tile16)
// (part of the pre-filter translation unit pf_flat.hip: included there, in order; gfx950 only)
#pragma once
#include "flat_tile_f32.hpp"
#include "flat_flush16.hpp"

namespace pf {

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
template <bool FILTER, int D, bool I8, bool PADDED = false>         // D = row length (a multiple of 16 up to 256): every loop below is compile-time; PADDED: p.d < D (flat_flush16.hpp)
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
            if constexpr (PADDED) {
                if (p.d % TK == 0) l2_tile_f32<FILTER, GEO, true, false>(p, smem, qt, ct);    // workgroup-uniform (a padded row length has the bounds-checked slab fetch)
                else l2_tile_f32<FILTER, GEO, false, false>(p, smem, qt, ct);
            } else l2_tile_f32<FILTER, GEO, true, false>(p, smem, qt, ct);
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
                        const int R = (int)row_qn - (int)ceilf(row_tau) - 256 * sx - 32768 * D;     // ceil: a fractional tau (none is produced today) keeps the filter a superset
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
            // (the walk's LAST flush is issued behind the loop, where the query fragments and thresholds are dead: it may use their registers)
            if (u == MT - 1 && ct + 1 != ct1) {                         // workgroup-uniform
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
                pend16_flush<D, MT, NJ, TN, I8, false, false, PADDED>(p, pend, stage, q0, tid, surv, ct - u, wm, wn, approx && !abl_exact, buf_cur, q_valid, false, (ct - ct0) / MT);
#pragma unroll
                for (int e = 0; e < MT * NJ; ++e) sv[e] = 0;
            }
        }
    }
    if constexpr (FILTER && !I8) {
        if (ct1 > ct0) {                                                // workgroup-uniform: the walk's last flush (both tile buffers are free: nothing in flight, nobody reads them)
            const uint32_t u_last = (ct1 - 1 - ct0) % MT;
            uint32_t surv[MT][NJ];
#pragma unroll
            for (int e = 0; e < MT * NJ; ++e) surv[e / NJ][e % NJ] = sv[e];
#ifdef PF_ABL_EXACTFLUSH
            constexpr bool abl_exact2 = true;
#else
            constexpr bool abl_exact2 = false;
#endif
            pend16_flush<D, MT, NJ, TN, I8, false, true, PADDED>(p, pend, stage, q0, tid, surv, ct1 - 1 - u_last, wm, wn, approx && !abl_exact2, smem, q_valid, true, (ct1 - 1 - ct0) / MT);
        }
    }
}

// the filtered int8 walk (flat_tile8.hpp); TILE8_LDS: the LDS its four waves carve out of the kernel's tile buffers
constexpr size_t TILE8_LDS = 4 * 17024;
template <int D, size_t SMEM_BYTES>
__device__ __forceinline__ void tile8_walk(const TileArgs &p, const uint32_t group, char *smem, float *stage, Pend16 &pend, const uint32_t qt, const uint32_t grp);

// WITH8: the instantiation that carries the streamed int8 walk (8-bit bases).  Bases that are not 8-bit launch the one without it: the walk's mere
// presence in the kernel changes the code hipcc builds for the bf16 loop (an s_waitcnt vmcnt(0) behind the tile's first copy request: +8 % on the
// long chunk of an N(0,1) search).
template <bool FILTER, int D, bool WITH8 = false, bool PADDED = false>
__global__ void __launch_bounds__(256, Geo16Of<D>::WG_PER_CU) k_l2_tile16(TileArgs p, uint32_t group, uint32_t n_groups) {
    using GEO = typename Geo16Of<D>::type;
    constexpr int TM = GEO::TM, TN = GEO::TN, PITCH = (D + (int)AUX16) * 2;
    constexpr size_t SMEM_T = 2 * (size_t)TN * PITCH > F32_TILE_LDS<GEO> ? 2 * (size_t)TN * PITCH : F32_TILE_LDS<GEO>;   // the fp32 fallback borrows this LDS
    constexpr size_t SMEM = (WITH8 && SMEM_T < TILE8_LDS) ? TILE8_LDS : SMEM_T;                                         // ... and so do the four waves of the streamed int8 walk
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
        if (p.xb8 && !(qflags & 7u)) {
            if constexpr (WITH8) {
                if (!p.i8_old) { tile8_walk<D, SMEM>(p, group, smem, stage, pend, qt, grp); return; }
            }
            tile16_walk<FILTER, D, true, PADDED>(p, group, n_groups, smem, stage, pend, qt, grp, qflags);
            return;
        }
    }
#ifdef PF_ABL_I8ONLY   // experiment (register count of the int8 walk on its own; other query tiles are not processed: wrong results for them)
    if constexpr (!(D % 32 == 0 && D <= 128))
#endif
    tile16_walk<FILTER, D, false, PADDED>(p, group, n_groups, smem, stage, pend, qt, grp, qflags);
}

}  // namespace pf
