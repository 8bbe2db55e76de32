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
// What it bought, measured against round 3's library on one lease (EXPERIMENTS.md): the long chunk of a 1024-query search takes what it took (141
// against 139 us); the gain is the cheaper survivors -- three chunks instead of four, one merge less -- 3.5-5 % of a search.  And it needs readers:
// with one or two query tiles a piece of the base is asked for by one or two waves and every step is a trip to memory, so batches of up to 256
// queries keep the LDS-tiled walk (pf_flat.hip: STREAMED_WALK_MIN_NQ), which is up to 1.7x faster there.
#pragma once
#ifndef PF_W8_WARM
#define PF_W8_WARM 0            // steps behind the first whose operands are pulled into L2 while the prologue runs (6: 0.347 -> 0.355 ms per search: off)
#endif
#ifndef PF_W8_FLUSH_U
#ifndef PF_W8_FLUSH_PIPE
#define PF_W8_FLUSH_PIPE 1
#endif
#define PF_W8_FLUSH_U (PF_W8_FLUSH_PIPE ? 4 : 8)
#endif
#include <type_traits>
#include "flat_tile16.hpp"

namespace pf {

using i32x4w = __attribute__((ext_vector_type(4))) int;

// LDS of one wave of the walk (carved out of the kernel's tile buffers, which this walk does not use)
struct Walk8Lds {
    static constexpr uint32_t RCAP = 1280, LCAP = 1024;                 // verdict records (8 bytes), decoded survivors (4 bytes)
    uint2 ring[RCAP];
    uint32_t list[LCAP];
    uint32_t rcnt[128], rbase[128];
    float qn[128];
    int r0[128];
};
static_assert(4 * sizeof(Walk8Lds) <= TILE8_LDS, "four of them fit the tile buffers of k_l2_tile16 at every row length");

template <int D>
struct Walk8 {
    static constexpr int KP = (D + 63) / 64 * 64, NKS = KP / 64;       // rows padded to whole 64-deep k-steps (zeros: value - 128 = 0)
    static constexpr int NI = 8;                                        // 16-row blocks of the 128-query tile
    static constexpr uint32_t STEP_BYTES = 2 * NKS * 1024;              // a step = 32 base rows = two 16-row blocks x NKS pieces of 1 KiB
};

// Survivors of one wave: the verdict records of its ring are decoded into (row, column) entries, every row with survivors reserves its range of the
// query's candidate list with one returning atomic, then 8 lanes per survivor recompute x.y from the two int8 rows (v_dot4_u32_u8: exact) and the
// group's first lane writes the key.  Wave-private: no barrier; LDS traffic of a wave executes in order.
template <int D>
__device__ __forceinline__ void walk8_flush(const TileArgs &p, Walk8Lds &L, const size_t q0, const uint32_t q_valid, const int lane, const uint32_t rc,
                                            const uint32_t step0, const uint32_t flush_no = 0) {
#ifdef PF_FLAT_STAMPS      // (tools/flat_stamps.py) stamps 0 entry | 1 first decode done | 2 first rows + reservations requested | 3 exit; 6 / 7: records, list entries
#define PF_F8STAMP(k, v) do { if (p.nb_count >= 400000 && blockIdx.x >= 256 && blockIdx.x < 256 + PF_FS_WGS && lane == 0 && flush_no < 8) \
    pf_flat_flush_stamp_buf[(((blockIdx.x - 256) * 4 + (threadIdx.x >> 6)) * 8 + flush_no) * 8 + (k)] = (v); } while (0)
    bool first_round = true;
#else
#define PF_F8STAMP(k, v) do { } while (0)
    (void)flush_no;
#endif
    PF_F8STAMP(0, __builtin_readcyclecounter());
    PF_F8STAMP(6, rc);
    constexpr uint32_t LU = D / 16;                                     // lanes of a group of 8 that hold 16 bytes of both rows
    constexpr int U = PF_W8_FLUSH_U;                                    // passes in flight (8 survivors each)
    uint32_t rb = 0, cur = 0, meta = 0;                                 // next batch of records; what is left of this lane's record
    for (;;) {
        // ---- decode: one survivor per lane and round until the ring is empty or the list could not take another round
        uint32_t ln = 0;                                                // entries in the list (wave-uniform)
        for (;;) {
            if (__ballot(cur != 0) == 0) {
                if (rb >= rc) break;                                    // wave-uniform
                const uint32_t idx = rb + (uint32_t)lane;
                const uint2 rec = idx < rc ? L.ring[idx] : make_uint2(0u, 0u);
                cur = rec.x; meta = rec.y;
                rb += 64;
                continue;
            }
            if (ln + 64 > Walk8Lds::LCAP) break;                        // the list is worked off first
            const bool has = cur != 0;
            const uint64_t m = __ballot(has);
            if (has) {
                const int b = 31 - __builtin_clz(cur);
                cur &= ~(1u << b);
                const uint32_t v = 31u - (uint32_t)b;                   // accumulator value 4 i + r of the lane that wrote the record
                const uint32_t ls = meta & 63u, cb = (meta >> 6) & 1u, srel = meta >> 7;
                const uint32_t row = 16u * (v >> 2) + 4u * (ls >> 4) + (v & 3u);
                const uint32_t col = srel * 32u + cb * 16u + (ls & 15u);            // relative to the wave's first step
                L.list[ln + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = (row << 24) | col;
                atomicAdd(&L.rcnt[row], 1u);
            }
            ln += (uint32_t)__popcll(m);
        }
        if (ln == 0) { PF_F8STAMP(3, __builtin_readcyclecounter()); return; }         // wave-uniform: nothing (left)
#ifdef PF_FLAT_STAMPS
        if (first_round) { PF_F8STAMP(1, __builtin_readcyclecounter()); PF_F8STAMP(7, ln); }
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // ---- evaluation: group g = lane >> 3 takes survivor e0 + 8 u + g, its lane l = lane & 7 holds bytes 16 l .. 16 l + 15 of both rows
        const uint32_t g = (uint32_t)lane >> 3, l = (uint32_t)lane & 7u;
        // Two register sets of U passes alternate: while one set's dot products run, the other set's rows are on their way (a flush is a chain of
        // round trips to memory with nothing else for the wave to do: with one set every batch exposed the whole trip).
        struct Set { u32x4 va[U], vb[U]; uint32_t row[U], id[U]; float bnv[U]; };
        auto request = [&](Set &S, uint32_t e0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t e = e0 + 8 * u + g < ln ? e0 + 8 * u + g : ln - 1;       // idle groups of the last pass repeat the last survivor
                const uint32_t ent = L.list[e];
                S.row[u] = ent >> 24;
                S.id[u] = (uint32_t)(p.nb_first + (size_t)step0 * 32 + (ent & 0xFFFFFFu));
                if (l < LU) {
                    S.va[u] = *reinterpret_cast<const u32x4 *>(p.xq8 + (q0 + (S.row[u] < q_valid ? S.row[u] : q_valid - 1)) * (size_t)D + 16 * l);
                    S.vb[u] = *reinterpret_cast<const u32x4 *>(p.xb8 + (size_t)S.id[u] * (D + AUX8) + 16 * l);     // (the row-major image: a row is 2-3 cache lines there, 8 in fragment order)
                } else {
                    S.va[u] = u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};  // lanes past the row: value 0 is stored as -128
                    S.vb[u] = S.va[u];
                }
                S.bnv[u] = p.bn[S.id[u]];
            }
        };
        auto finish = [&](const Set &S, uint32_t e0) {
            uint32_t pos[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pos[u] = (l == 0 && e0 + 8 * u + g < ln) ? atomicAdd(&L.rbase[S.row[u]], 1u) : ~0u;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t si = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t wa = S.va[u][w] ^ 0x80808080u, wb = S.vb[u][w] ^ 0x80808080u;  // value = stored byte with its top bit flipped, as an unsigned byte
                    si = __builtin_amdgcn_udot4(wa, wb, si, false);
                }
                si += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)si, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
                si += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)si, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
                si += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)si, 0x141, 0xf, 0xf, true);   // row_half_mirror
                if (pos[u] < p.cap) {                                   // (~0 for idle lanes and groups; at or past cap: the list overflowed, k_select rescans the chunk)
                    const float dist = fmaf(-2.f, (float)si, L.qn[S.row[u]] + S.bnv[u]);          // x.y below 2^24: exact; the fp32 expression of every other path
                    p.cand[(q0 + S.row[u]) * p.cap + pos[u]] = make_key(dist < 0.f ? 0.f : dist, S.id[u]);
                }
            }
        };
        Set SA, SB;
        request(SA, 0);
        // the rows' ranges of the candidate lists (lane l: rows l and l + 64): one returning atomic per row with survivors, issued behind the first
        // batch's row loads -- one round trip to memory for both
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t r = (uint32_t)lane + 64u * h;
            const uint32_t c = L.rcnt[r];
            L.rbase[r] = (c && r < q_valid) ? atomicAdd(&p.cand_cnt[q0 + r], c) : 0u;
            L.rcnt[r] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#ifdef PF_FLAT_STAMPS
        if (first_round) { PF_F8STAMP(2, __builtin_readcyclecounter()); first_round = false; }
#endif
#if PF_W8_FLUSH_PIPE
        for (uint32_t e0 = 0; e0 < ln; e0 += 16 * U) {
            if (e0 + 8 * U < ln) request(SB, e0 + 8 * U);               // wave-uniform
            finish(SA, e0);
            if (e0 + 8 * U >= ln) break;
            if (e0 + 16 * U < ln) request(SA, e0 + 16 * U);
            finish(SB, e0 + 8 * U);
        }
#else
        (void)SB;
        for (uint32_t e0 = 0; e0 < ln; e0 += 8 * U) {
            if (e0) request(SA, e0);
            finish(SA, e0);
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
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
    const int *const c0img = p.c0f + p.nb_first;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    // one B operand set = the 2 x NKS pieces of a step; two sets alternate (the loop below is written out for both parities)
    i32x4w bA[2][NKS], bB[2][NKS];
    int c0A[2], c0B[2];
    auto fetch = [&](i32x4w (&b)[2][NKS], int (&c0)[2], uint32_t s) {
        const char *src = img + (size_t)s * STEP_BYTES + lane16;       // wave-uniform base + lane offset
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) b[cb][ks] = *reinterpret_cast<const i32x4w *>(src + (cb * NKS + ks) * 1024);
        const int2 c = *reinterpret_cast<const int2 *>(c0img + (size_t)s * 32 + 2 * (lane & 15));
        c0[0] = -(c.x >> 1); c0[1] = -(c.y >> 1);                       // the image keeps C; the filter wants c0 = -floor(C / 2)
    };
    // The walk's first operands are requested before anything else (one round trip to memory for the whole prologue), and the PF_W8_WARM steps
    // behind them are touched (a dword of each 64-byte line, one load per step): every wave of the launch starts at once, and the eight waves that
    // share a piece ask for it together, so the first steps would otherwise each wait for a request that goes all the way to memory.
    fetch(bA, c0A, s0);
    uint32_t warm[PF_W8_WARM > 0 ? PF_W8_WARM : 1] = {};
#pragma unroll
    for (uint32_t a = 0; a < PF_W8_WARM; ++a) {
        const uint32_t st = s0 + 1 + a < s1 ? s0 + 1 + a : s1 - 1, l = (uint32_t)lane < STEP_BYTES / 64 ? (uint32_t)lane : 0u;
        warm[a] = *reinterpret_cast<const volatile uint32_t *>(img + (size_t)st * STEP_BYTES + l * 64u);
    }
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
    // ---- row halves of the thresholds: r0 = -(floor(R / 2) + 1), R = |x|^2 - ceil(tau) - 256 sum x' - 32768 d (lane l: rows l and l + 64).
    // |S| < 2^21, |r0| < 2^26, |c0| < 2^24: the sentinels +-2^29 of "everything passes" / "nothing passes" stay clear of every sum.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t r = (uint32_t)lane + 64u * h;
        int r0 = -(1 << 29);                                           // rows past nq: nothing passes
        float row_qn = 0.f;
        if (r < q_valid) {
            row_qn = p.qn[q0 + r];
            const float row_tau = p.tau[q0 + r];
            if (row_tau == INFINITY) r0 = 1 << 29;                     // fewer than k results so far: everything passes
            else {
                const int R = (int)row_qn - (int)ceilf(row_tau) - 256 * p.qsx8[q0 + r] - 32768 * D;
                r0 = -(R >> 1) - 1;                                    // (>> of a negative int: floor)
            }
        }
        L.qn[r] = row_qn;
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
    for (int i = 0; i < NI; ++i) { acc[i][0] = r0t[i]; acc[i][1] = r0t[i]; }      // (defined values for the first half's sweep, whose word is dropped)
    // two accumulator values of the sweep: the value passes where S + r0 + c0 >= 0; its sign bit is shifted into the lane's word (value 4 i + r ends
    // up in bit 31 - (4 i + r); set = fails)
    auto sweep2 = [&](const i32x4w &a, int c0, uint32_t &fail, int r2) {
        fail = __builtin_amdgcn_alignbit(fail, (uint32_t)(a[2 * r2] + c0), 31);
        fail = __builtin_amdgcn_alignbit(fail, (uint32_t)(a[2 * r2 + 1] + c0), 31);
    };
    // the 8 x NKS matrix instructions of column block CB of a step beside the sweep of the other block's accumulators (column half c0s) into `fail`
    auto half = [&](auto CBc, const i32x4w (&b)[2][NKS], int c0s, uint32_t &fail, auto &&pre) {
        constexpr int CB = decltype(CBc)::value;
        constexpr int NM = NI * NKS;                                    // matrix instructions of the half; 16 sweep pairs spread over them
        __builtin_amdgcn_sched_barrier(0);
        pre();                                                          // (the next step's operand requests: placed between the matrix instructions, not in a burst)
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
        asm volatile("" : "+v"(fail));                                  // (the word is complete HERE: without this the compiler sinks the sweep down to the append that reads it)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one matrix instruction ...
            if (CB == 0 && m % 3 == 1 && m / 3 < 2 * NKS + 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // ... a request behind every third (the half that carries them) ...
            __builtin_amdgcn_sched_group_barrier(0x002, 64 / NM, 0);    // ... then its share of the sweep's 64 vector instructions
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto nothing = []() {};
    // verdict words -> records in the wave's ring: w0 belongs to block 0 of step s, w1 to block 1 of step s - 1 (finished one half later)
    auto append = [&](uint32_t f0, uint32_t f1, uint32_t s, bool first) {
        const uint32_t col0 = s * 32u + ((uint32_t)lane & 15u), col1 = col0 - 16u;       // (block 1 of step s - 1 = columns 32 (s - 1) + 16 ..)
        const uint32_t w0 = col0 < p.nb_count ? ~f0 : 0u;               // columns past the end of the chunk belong to the next one
        const uint32_t w1 = (!first && col1 < p.nb_count) ? ~f1 : 0u;
        const uint64_t m0 = __ballot(w0 != 0), m1 = __ballot(w1 != 0);
        if ((m0 | m1) == 0) return;                                     // wave-uniform
        const uint32_t n0 = (uint32_t)__popcll(m0);
        if (w0) L.ring[rc + __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))] = make_uint2(w0, ((s - s0) << 7) | (uint32_t)lane);
        if (w1) L.ring[rc + n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))] = make_uint2(w1, ((s - 1 - s0) << 7) | 64u | (uint32_t)lane);
        rc += n0 + (uint32_t)__popcll(m1);
    };
    uint32_t nflush = 0;                                                // (flushes of this walk so far: only the stamped build looks at it)
    auto flush_if_full = [&]() {
        if (rc + 128 > Walk8Lds::RCAP) {                                // a step adds at most 128 records
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            walk8_flush<D>(p, L, q0, q_valid, lane, rc, s0, nflush++);
            rc = 0;
            load_afrag();                                               // (registers that need not live across the flush: fetched again)
            load_r0t();
        }
    };
#ifdef PF_FLAT_STAMPS
    const bool fs_on = p.nb_count >= 400000 && blockIdx.x >= 256 && blockIdx.x < 256 + PF_FS_WGS;
    const uint32_t ct0 = s0;
#endif
    uint32_t f1 = 0;                                                    // block 1's word of the step before (swept during this step's first half)
    int c0p = 0;                                                        // ... and its column half
    for (uint32_t s = s0; s < s1; s += 2) {
        // even position of the walk: operands in set A; set B is requested for step s + 1 (clamped: the last step asks for itself again -- no branch)
#ifdef PF_FLAT_STAMPS
        const uint32_t ct = s0 + (s - s0) / 2;                         // (every second step is stamped: a period below = two steps)
#endif
        PF_FSTAMP(0);
        PF_FSTAMP(1);
        f1 = 0;
        half(std::integral_constant<int, 0>{}, bA, c0p, f1, [&]() { fetch(bB, c0B, s + 1 < s1 ? s + 1 : s1 - 1); });
        PF_FSTAMP(2);
        uint32_t f0 = 0;
        half(std::integral_constant<int, 1>{}, bA, c0A[0], f0, nothing);
        PF_FSTAMP(3);
        append(f0, f1, s, s == s0);
        PF_FSTAMP(4);
        c0p = c0A[1];
        flush_if_full();
        PF_FSTAMP(5);
        if (s + 1 >= s1) break;                                         // wave-uniform
        f1 = 0;
        half(std::integral_constant<int, 0>{}, bB, c0p, f1, [&]() { fetch(bA, c0A, s + 2 < s1 ? s + 2 : s1 - 1); });
        f0 = 0;
        half(std::integral_constant<int, 1>{}, bB, c0B[0], f0, nothing);
        append(f0, f1, s + 1, false);
        c0p = c0B[1];
        flush_if_full();
    }
    // the last step's second block, then whatever the ring holds
    {
        uint32_t fl = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) sweep2(acc[g >> 1][1], c0p, fl, g & 1);
        const uint32_t col1 = (s1 - 1) * 32u + 16u + ((uint32_t)lane & 15u);
        const uint32_t w1 = col1 < p.nb_count ? ~fl : 0u;
        const uint64_t m1 = __ballot(w1 != 0);
        if (w1) L.ring[rc + (uint32_t)__popcll(m1 & ((1ull << lane) - 1))] = make_uint2(w1, ((s1 - 1 - s0) << 7) | 64u | (uint32_t)lane);
        rc += (uint32_t)__popcll(m1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        walk8_flush<D>(p, L, q0, q_valid, lane, rc, s0, nflush);
        uint32_t sink = 0;
#pragma unroll
        for (uint32_t a = 0; a < PF_W8_WARM; ++a) sink ^= warm[a];
        if (sink == 0x5EEDFACEu && p.nq == 0xFFFFFFFFu) L.rcnt[0] = sink;                  // (never true: keeps the warm-up touches alive)
    }
}

}  // namespace pf
