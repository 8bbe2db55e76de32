// flat_tile8.hpp -- the filtered int8 walk of the pre-filter (8-bit data: every value of the base and of the query tile an integer in [0, 255])
// (part of the pre-filter translation unit pf_flat.hip: included there, in order; gfx950 only)
//
// The integer filter is flat_tile16.hpp's (x' = x - 128, y' = y - 128, S = sum x'y' accumulated exactly by the int8 matrix instruction;
// dist < tau <=> S + r0 + c0 >= 0 with the row half r0 = -(floor(R / 2) + 1), the column half c0 = -floor(C / 2); every survivor evaluated exactly by
// v_dot4_u32_u8).  What changed in round 4 is how the walk is run.  Phase stamps of round 3's walk (profiles/r03_z_flat_stamps.txt) had a 128 x 128
// tile at 3 854 cycles of which the matrix instructions' own time is 512: the rest was the machinery around them -- LDS-DMA requests (60-100 cycles of
// wave issue each), fragment reads, a workgroup barrier and a vmcnt(0) per tile, an initialisation pass and a sweep that ran AFTER the matrix
// instructions, in lockstep with the partner wave of the SIMD.  Now:
//   * EVERY WAVE WALKS ON ITS OWN.  The base is kept in matrix-fragment order (flat_common.hpp: frag8_offset): the 1 KiB a wave loads with one
//     global_load_dwordx4 IS the B operand of v_mfma_i32_16x16x64_i8.  No LDS staging, no LDS-DMA, no barrier anywhere in the walk; the four waves of a
//     workgroup take a quarter of its columns each, keep all 128 query rows of the tile in registers (64 VGPRs) and prefetch the next step's 4 KiB
//     while they work on this one.  Each 1 KiB piece is read by 8 waves (the 8 query tiles of a batch of 1024), as with the LDS tiles.
//   * The row halves never touch the vector pipe: the first matrix instruction of a block takes them as its C operand (32 VGPRs in the 16 x 16
//     accumulator layout, loaded once), so the accumulators come out as S + r0.  The sweep is then add c0, shift the sign bit into the lane's word.
//   * Matrix instructions and sweep are interleaved in ONE wave: while the matrix pipe works on one 16-column block (16 instructions into 32
//     accumulator registers), the vector pipe sweeps the other block's finished accumulators, two values behind every matrix instruction.
//   * Survivors are handled by the wave that found them: verdict words go to a wave-private ring in LDS; when it fills, or the walk ends, the wave
//     decodes them, reserves its rows' ranges of the candidate lists and evaluates the survivors, 8 lanes per survivor -- while the other wave of the
//     SIMD keeps the pipes busy.  (Before, a flush stopped the whole workgroup behind three barriers.)
#pragma once
#ifndef PF_W8_AHEAD
#define PF_W8_AHEAD 3           // steps ahead of the walk at which a step's operands are pulled into L2 (0: off)
#endif
#include <type_traits>
#include "flat_tile16.hpp"

namespace pf {

using i32x4w = __attribute__((ext_vector_type(4))) int;

// LDS of one wave of the walk (carved out of the kernel's tile buffers, which this walk does not use).  A RECORD is what one lane found in one
// 16-column block of a step: (meta, verdict word, the column's C, pad, its 32 accumulator values) = 36 dwords.  The accumulators are S + r0, exact
// integers, so the flush forms the survivors' distances from them -- no row is ever fetched again.
struct Walk8Lds {
    static constexpr uint32_t RDW = 36, RCAP = 104;                     // dwords per record; records (a half-step adds at most 64)
    __attribute__((aligned(16))) uint32_t ring[RCAP * RDW];
    uint32_t rcnt[128], rbase[128];
    int rq[128];                                                        // |x|^2 - 256 sum x' - 32768 d of the tile's rows
    int r0[128];                                                        // the rows' threshold halves
};
static_assert(4 * sizeof(Walk8Lds) <= TILE8_LDS, "four of them fit the tile buffers of k_l2_tile16 at every row length");

template <int D>
struct Walk8 {
    static constexpr int KP = (D + 63) / 64 * 64, NKS = KP / 64;       // rows padded to whole 64-deep k-steps (zeros: value - 128 = 0)
    static constexpr int NI = 8;                                        // 16-row blocks of the 128-query tile
    static constexpr uint32_t STEP_BYTES = 2 * NKS * 1024;              // a step = 32 base rows = two 16-row blocks x NKS pieces of 1 KiB
};

// Survivors of one wave.  Every lane walks the verdict word of ITS record, one survivor per lane and round; a survivor's distance is
//   dist = |x|^2 + |y|^2 - 2 x.y = (|x|^2 - 256 sum x' - 32768 d) + (|y|^2 - 256 sum y') - 2 S = rq[row] + C - 2 (acc - r0[row]),
// all exact integers below 2^24 -- the number the fp32 expression fmaf(-2, x.y, |x|^2 + |y|^2) of every other path yields, bit for bit.  Two passes
// over the ring: count per query row, reserve the rows' ranges of the candidate lists (one returning atomic per row with survivors), write the
// keys.  Wave-private: no barrier; the LDS traffic of a wave executes in order.
template <int D>
__device__ __forceinline__ void walk8_flush(const TileArgs &p, Walk8Lds &L, const size_t q0, const uint32_t q_valid, const int lane, const uint32_t rc,
                                            const uint32_t step0) {
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t rb = 0; rb < rc; rb += 64) {
            const uint32_t rec = rb + (uint32_t)lane < rc ? rb + (uint32_t)lane : rc - 1;
            const uint32_t *r = &L.ring[rec * Walk8Lds::RDW];
            const uint32_t meta = r[0];
            uint32_t cur = rb + (uint32_t)lane < rc ? r[1] : 0u;
            const int C = (int)r[2];
            const uint32_t ls = meta & 63u, cb = (meta >> 6) & 1u, srel = meta >> 7;
            const uint32_t col = srel * 32u + cb * 16u + (ls & 15u);   // relative to the wave's first step
            if ((size_t)step0 * 32 + col >= p.nb_count) cur = 0;       // columns past the end of the chunk belong to the next one: dropped here, not in the walk
            const uint32_t id = (uint32_t)(p.nb_first + (size_t)step0 * 32 + col);
            while (__ballot(cur != 0)) {
                if (cur != 0) {
                    const int b = 31 - __builtin_clz(cur);
                    cur &= ~(1u << b);
                    const uint32_t v = 31u - (uint32_t)b;               // accumulator value 4 i + r of the lane that wrote the record
                    const uint32_t row = 16u * (v >> 2) + 4u * (ls >> 4) + (v & 3u);
                    if (pass == 0) atomicAdd(&L.rcnt[row], 1u);
                    else {
                        const uint32_t pos = atomicAdd(&L.rbase[row], 1u);
                        if (pos < p.cap) {                              // (at or past cap: the list overflowed, k_select rescans the chunk)
                            const int S = (int)r[4 + v] - L.r0[row];
                            const int dist = L.rq[row] + C - 2 * S;
                            p.cand[(q0 + row) * p.cap + pos] = make_key((float)(dist < 0 ? 0 : dist), id);
                        }
                    }
                }
            }
        }
        if (pass == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int h = 0; h < 2; ++h) {                               // lane l: rows l and l + 64
                const uint32_t row = (uint32_t)lane + 64u * h;
                const uint32_t c = L.rcnt[row];
                L.rbase[row] = (c && row < q_valid) ? atomicAdd(&p.cand_cnt[q0 + row], c) : 0u;
                L.rcnt[row] = 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <int D, size_t SMEM_BYTES>
__device__ __forceinline__ void tile8_walk(const TileArgs &p, const uint32_t group, char *smem, float *stage, Pend16 &pend, const uint32_t qt, const uint32_t grp) {
    (void)stage; (void)pend;
    constexpr int KP = Walk8<D>::KP, NKS = Walk8<D>::NKS, NI = Walk8<D>::NI;
    constexpr uint32_t STEP_BYTES = Walk8<D>::STEP_BYTES;
    static_assert(D % 16 == 0 && D <= 128 && SMEM_BYTES >= 4 * sizeof(Walk8Lds), "rows of whole 16-byte lanes; the waves' LDS fits the kernel's tile buffers");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Walk8Lds &L = reinterpret_cast<Walk8Lds *>(smem)[wave];
    const size_t q0 = (size_t)qt * 128;
    const uint32_t q_valid = (uint32_t)(p.nq - q0 < 128 ? p.nq - q0 : 128);
    // this wave's steps of 32 columns: a quarter of the workgroup's column tiles (4 steps each), clipped to the chunk
    const uint32_t n_steps = (uint32_t)((p.nb_count + 31) / 32);
    const uint32_t wg0 = grp * group * 4u, wg1 = wg0 + group * 4u < n_steps ? wg0 + group * 4u : n_steps;
    if (wg0 >= wg1) return;
    const uint32_t per = (wg1 - wg0 + 3u) / 4u;
    const uint32_t s0 = wg0 + (uint32_t)wave * per, s1 = s0 + per < wg1 ? s0 + per : wg1;
    if (s0 >= s1) return;                                               // (no barrier below: a wave without columns just leaves)
    // base of step s: 32 rows = STEP_BYTES consecutive bytes of the image; lane l reads 16 bytes at l * 16 of each 1 KiB piece
    const char *const img = reinterpret_cast<const char *>(p.xb8f) + (size_t)(p.nb_first / 32) * STEP_BYTES;
    const int *const cimg = p.c0f + p.nb_first;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    // ---- the query operand: lane l holds row l & 15 of each 16-row block, 16 consecutive k of every 64-deep step starting at 16 (l >> 4)
    i32x4w afrag[NI][NKS];
    auto load_afrag = [&]() {
        const char *qimg = reinterpret_cast<const char *>(p.xq8);
        asm volatile("" : "+s"(qimg));                                  // (opaque per call: otherwise the eight row addresses are kept in 16 registers through the walk for the rare reload)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const uint32_t r = 16u * i + ((uint32_t)lane & 15u);
            const char *row = qimg + (q0 + (r < q_valid ? r : q_valid - 1)) * (size_t)D;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const uint32_t k = 64u * ks + 16u * ((uint32_t)lane >> 4);
                if (KP == D || k + 16 <= (uint32_t)D) afrag[i][ks] = *reinterpret_cast<const i32x4w *>(row + k);
                else afrag[i][ks] = i32x4w{0, 0, 0, 0};                 // the padding of the last k-step
            }
        }
    };
    load_afrag();
    // ---- row halves of the thresholds: r0 = -(floor(R / 2) + 1), R = rq - ceil(tau), rq = |x|^2 - 256 sum x' - 32768 d (lane l: rows l and l + 64).
    // |S| < 2^21, |r0| < 2^26, |C| < 2^25: the sentinels +-2^29 of "everything passes" / "nothing passes" stay clear of every sum.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t r = (uint32_t)lane + 64u * h;
        int r0 = -(1 << 29), rq = 0;                                   // rows past nq: nothing passes
        if (r < q_valid) {
            rq = (int)p.qn[q0 + r] - 256 * p.qsx8[q0 + r] - 32768 * D;
            const float row_tau = p.tau[q0 + r];
            if (row_tau == INFINITY) r0 = 1 << 29;                     // fewer than k results so far: everything passes
            else r0 = -((rq - (int)ceilf(row_tau)) >> 1) - 1;          // (>> of a negative int: floor)
        }
        L.rq[r] = rq;
        L.r0[r] = r0;
        L.rcnt[r] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // D[row = 4 (lane >> 4) + r][column = lane & 15] of a 16 x 16 block: the row halves in the accumulators' own layout
    i32x4w r0t[NI];
    auto load_r0t = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) r0t[i] = *reinterpret_cast<const i32x4w *>(&L.r0[16 * i + 4 * (lane >> 4)]);
    };
    load_r0t();
    // ---- the walk
    uint32_t rc = 0;                                                    // records in this wave's ring (wave-uniform)
    i32x4w acc[NI][2];                                                  // accumulators of the two 16-column blocks of a step
#pragma unroll
    for (int i = 0; i < NI; ++i) {                                      // (the first half sweeps block 1 of a step that does not exist: "nothing passes" there)
        acc[i][0] = r0t[i];
        acc[i][1] = i32x4w{-(1 << 29), -(1 << 29), -(1 << 29), -(1 << 29)};
    }
    // one B operand set = the 2 x NKS pieces of a step + the two columns' C; two sets alternate (the loop below is written out for both parities)
    i32x4w bA[2][NKS], bB[2][NKS];
    int cA[2], cB[2];
    auto fetch = [&](i32x4w (&b)[2][NKS], int (&c)[2], uint32_t s) {
#ifdef PF_ABL_W8_HOTB   // ablation (timing only, wrong results): every wave reads the same 256 steps of the image -- what the walk costs when no request misses
        const char *src = img + (size_t)(s & 255u) * STEP_BYTES + lane16;
#else
        const char *src = img + (size_t)s * STEP_BYTES + lane16;       // wave-uniform base + lane offset
#endif
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) b[cb][ks] = *reinterpret_cast<const i32x4w *>(src + (cb * NKS + ks) * 1024);
        const int2 cc = *reinterpret_cast<const int2 *>(cimg + (size_t)s * 32 + 2 * (lane & 15));
        c[0] = cc.x; c[1] = cc.y;
    };
    // two accumulator values of the sweep: the value passes where S + r0 + c0 >= 0, c0 = -floor(C / 2); its sign bit is shifted into the lane's word
    // (value 4 i + r ends up in bit 31 - (4 i + r); set = fails)
    auto sweep2 = [&](const i32x4w &a, int c0, uint32_t &fail, int r2) {
        fail = __builtin_amdgcn_alignbit(fail, (uint32_t)(a[2 * r2] + c0), 31);
        fail = __builtin_amdgcn_alignbit(fail, (uint32_t)(a[2 * r2 + 1] + c0), 31);
    };
    // the 8 x NKS matrix instructions of column block CB of a step beside the sweep of the other block's accumulators (its column's C) into `fail`;
    // `pre`: issued inside the scheduled region (the next step's operand requests, placed between the matrix instructions instead of in a burst)
    auto half = [&](auto CBc, const i32x4w (&b)[2][NKS], int Cs, uint32_t &fail, auto &&pre) {
        constexpr int CB = decltype(CBc)::value;
        constexpr int NM = NI * NKS;                                    // matrix instructions of the half; 16 sweep pairs spread over them
        const int c0s = -(Cs >> 1);
        __builtin_amdgcn_sched_barrier(0);
        pre();
        int g = 0;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                acc[i][CB] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag[i][ks], b[CB][ks], ks == 0 ? r0t[i] : acc[i][CB], 0, 0, 0);
                const int m = ks * NI + i, g_end = (16 * (m + 1) + NM - 1) / NM;       // pairs due after matrix instruction m
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    if (g + e < g_end) sweep2(acc[(g + e) >> 1][1 - CB], c0s, fail, (g + e) & 1);
                g = g_end;
            }
        asm volatile("" : "+v"(fail));                                  // (the word is complete HERE: without this the compiler sinks the sweep down to the branch that reads it)
        // the order the scheduler is to emit: a matrix instruction, its share of the sweep's 64 vector instructions, and -- in the half that carries
        // the requests -- one load behind every third matrix instruction
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (CB == 0 && m % 2 == 1 && m / 2 < 2 * NKS + 1 + (PF_W8_AHEAD ? 1 : 0)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 64 / NM, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto nothing = []() {};
    // The operands are requested one step ahead into registers; that hides an L2 hit, not a miss -- and the eight waves that share a piece (one per
    // query tile) ask for it at about the same time, so all of them would wait for the one request that goes to memory.  A step's 2 x NKS KiB are
    // therefore touched PF_W8_AHEAD steps ahead by ONE load per wave (a dword of each 64-byte line: STEP_BYTES / 64 lanes), whose value is only
    // looked at two steps later (folded into `sink`, which nothing depends on).
    uint32_t sink = 0;
    auto touch = [&](uint32_t s) -> uint32_t {
        const uint32_t l = (uint32_t)lane < STEP_BYTES / 64 ? (uint32_t)lane : 0u;
        return *reinterpret_cast<const volatile uint32_t *>(img + (size_t)(s < s1 ? s : s1 - 1) * STEP_BYTES + l * 64u);
    };
    // the lanes that found something in the block just swept (accumulators still in place: the next half overwrites them) leave a record
    auto append = [&](auto CBc, uint32_t fail, int Cs, uint32_t s_of_block) {
        constexpr int CB = decltype(CBc)::value;                        // the block that was SWEPT
        const uint32_t w = ~fail;
        const uint64_t m = __ballot(w != 0);
        if (m == 0) return;                                             // wave-uniform
        const uint32_t n = (uint32_t)__popcll(m);
        if (rc + n > Walk8Lds::RCAP) {                                  // wave-uniform, rare: the ring is worked off first
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            walk8_flush<D>(p, L, q0, q_valid, lane, rc, s0);
            rc = 0;
            load_afrag();                                               // (registers that need not live across the flush: fetched again)
            load_r0t();
        }
        if (w) {
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));      // lanes below this one in m
            uint32_t *r = &L.ring[(rc + before) * Walk8Lds::RDW];
            *reinterpret_cast<u32x4 *>(r) = u32x4{((s_of_block - s0) << 7) | ((uint32_t)CB << 6) | (uint32_t)lane, w, (uint32_t)Cs, 0u};
#pragma unroll
            for (int i = 0; i < NI; ++i) *reinterpret_cast<i32x4w *>(r + 4 + 4 * i) = acc[i][CB];
        }
        rc += n;
    };
#ifdef PF_FLAT_STAMPS
    const bool fs_on = p.nb_count >= 400000 && blockIdx.x >= 256 && blockIdx.x < 256 + PF_FS_WGS;
    const uint32_t ct0 = s0;
#endif
    int Cp = 0;                                                         // C of the step before's block 1
    // One step: block 0's matrix instructions beside the sweep of the step before's block 1, then block 1's beside the sweep of block 0
    auto step = [&](const i32x4w (&b)[2][NKS], const int (&c)[2], i32x4w (&bn)[2][NKS], int (&cn)[2], uint32_t s, uint32_t s_next, uint32_t &ahead) {
        uint32_t f1 = 0, f0 = 0;
        half(std::integral_constant<int, 0>{}, b, Cp, f1, [&]() {
            fetch(bn, cn, s_next);
#if PF_W8_AHEAD
            sink ^= ahead;                                              // (the touch of two steps ago: long back)
            ahead = touch(s + PF_W8_AHEAD);
#endif
        });
        append(std::integral_constant<int, 1>{}, f1, Cp, s - 1);       // (s - 1 underflows for the walk's first step: its word is empty)
        half(std::integral_constant<int, 1>{}, b, c[0], f0, nothing);
        append(std::integral_constant<int, 0>{}, f0, c[0], s);
        Cp = c[1];
    };
    fetch(bA, cA, s0);
    uint32_t aheadA = 0, aheadB = 0;
#if PF_W8_AHEAD
    for (uint32_t a = 1; a < PF_W8_AHEAD; ++a) sink ^= touch(s0 + a);  // (the steps the loop's touches start behind)
#endif
    const uint32_t last = s1 - 1;
    for (uint32_t s = s0; s < s1; s += 2) {
        // operand sets A, B alternate; the set not in use is requested one step ahead (clamped to the walk's last step: no branch around the requests)
#ifdef PF_FLAT_STAMPS
        const uint32_t ct = s0 + (s - s0) / 2;                         // (a stamped period = two steps)
#endif
        PF_FSTAMP(0);
        step(bA, cA, bB, cB, s, s + 1 < last ? s + 1 : last, aheadA);
        PF_FSTAMP(1);
        if (s + 1 >= s1) break;                                         // wave-uniform
        step(bB, cB, bA, cA, s + 1, s + 2 < last ? s + 2 : last, aheadB);
        PF_FSTAMP(2);
    }
    // the last step's second block (its sweep has nothing beside it), then whatever the ring holds
    {
        uint32_t fl = 0;
        const int c0s = -(Cp >> 1);
#pragma unroll
        for (int g = 0; g < 16; ++g) sweep2(acc[g >> 1][1], c0s, fl, g & 1);
        append(std::integral_constant<int, 1>{}, fl, Cp, s1 - 1);
        if ((sink ^ aheadA ^ aheadB) == 0x5EEDFACEu && p.nq == 0xFFFFFFFFu) L.rcnt[0] = sink;      // (never true: keeps the touches alive)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        walk8_flush<D>(p, L, q0, q_valid, lane, rc, s0);
    }
}

}  // namespace pf
