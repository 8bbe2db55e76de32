// flat_tile8.hpp -- the filtered int8 tile walk of the pre-filter (8-bit data: every value of the base and of the query tile an integer in [0, 255])
// (part of the pre-filter translation unit pf_flat.hip: included there, in order; gfx950 only)
//
// What the walk computes is flat_tile16.hpp's integer filter (x' = x - 128, y' = y - 128, S = sum x'y' accumulated by v_mfma_i32_32x32x32_i8;
// dist < tau <=> S + r0 + c0 >= 0 with the row half r0 = -(floor(R / 2) + 1), the column half c0 = -floor(C / 2); survivors evaluated exactly by
// v_dot4_u32_u8) -- restructured so that a tile costs the vector pipe ONE instruction per accumulator value instead of two:
//   * the row halves never touch the vector pipe: the first matrix instruction of a tile takes them as its C operand (two 16-register tuples per
//     lane, loaded once per walk), so the accumulators come out as S + r0 with no initialisation pass;
//   * the column half is the same for all 32 accumulator values of a lane and column block, so the verdict is ONE compare per value against
//     -c0 (v_cmp_ge_i32 writing the wave's 64-bit mask of that accumulator row into scalar registers); the masks are OR-ed on the scalar unit and
//     a block of sixteen rows without a survivor -- most of them in the long late chunks -- costs nothing more;
//   * a non-zero mask IS the list of survivors of one accumulator row: its lanes append (tile, query row, column) records to the wave's ring in
//     LDS, 4 bytes each, and the flush evaluates them straight from the rings -- no verdict words to decode, no list to build first.
// (Before: 64 additions to start the accumulators at r0 + c0, 64 v_alignbit for the sign bits, a ballot and a record per non-zero word, and a
// decode of the words into a list at the flush: 3 854 cycles per tile of which the 16 matrix instructions' own time is 512; profiles/r03_z_flat_stamps.txt.)
#pragma once
#include "flat_tile16.hpp"

namespace pf {

#ifndef PF_W8_NBUF
#define PF_W8_NBUF 2            // column tiles in LDS: 2 = the next one is requested while this one is worked on; 3 = two ahead (counted vmcnt, raw barrier)
#endif

// Survivors of the rings, evaluated exactly and appended to their queries' candidate lists.  rings: 4 x RCAP records (trel << 14 | local row << 7 |
// column inside the tile), cnt[w] of them valid in ring w (the same four numbers in every thread).  Barriers inside: call from all threads.
template <int D>
__device__ __forceinline__ void walk8_flush(const TileArgs &p, Pend16 &pd, const float *sA, const size_t q0, const int tid, const uint32_t *rings,
                                            const uint32_t RCAP, const uint32_t ct0, const uint32_t c0, const uint32_t c1, const uint32_t c2, const uint32_t c3) {
    constexpr uint32_t LU = D / 16, L = LU <= 2 ? 2 : LU <= 4 ? 4 : 8, G = 256 / L;       // lanes per survivor (16 bytes of both rows each), survivors per pass
    constexpr int U = PF_FLUSH_U;
    const uint32_t o1 = c0, o2 = o1 + c1, o3 = o2 + c2, total = o3 + c3;
    if (total == 0) return;                                             // workgroup-uniform
    auto rec_at = [&](uint32_t e) {
        const uint32_t w = (uint32_t)(e >= o1) + (uint32_t)(e >= o2) + (uint32_t)(e >= o3);
        const uint32_t base = w == 0 ? 0u : w == 1 ? o1 : w == 2 ? o2 : o3;
        return rings[w * RCAP + (e - base)];
    };
    for (uint32_t e = tid; e < total; e += 256) atomicAdd(&pd.rcnt[(rec_at(e) >> 7) & 127u], 1u);
    __syncthreads();
    const uint32_t g = (uint32_t)tid / L, l = (uint32_t)tid % L;
    for (uint32_t e0 = 0; e0 < total; e0 += G * U) {
        u32x4 va[U], vb[U];
        uint32_t loc[U], id[U];
        float bnv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = e0 + u * G + g < total ? e0 + u * G + g : total - 1;     // idle groups of the last pass repeat the last survivor
            const uint32_t rec = rec_at(e);
            loc[u] = (rec >> 7) & 127u;
            id[u] = (uint32_t)(p.nb_first + (size_t)(ct0 + (rec >> 14)) * 128 + (rec & 127u));
            if (LU == L || l < LU) {
                va[u] = *reinterpret_cast<const u32x4 *>(p.xq8 + (q0 + loc[u]) * (size_t)D + 16 * l);
                vb[u] = *reinterpret_cast<const u32x4 *>(p.xb8 + (size_t)id[u] * (D + AUX8) + 16 * l);
            } else {
                va[u] = u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};     // lanes past the row: value 0 is stored as -128
                vb[u] = va[u];
            }
            bnv[u] = p.bn[id[u]];
        }
        if (e0 == 0) {                                                  // workgroup-uniform: the rows' ranges of the candidate lists, one returning atomic per row with survivors,
            if (tid < 128) {                                            // travelling together with the first pass's row loads
                const uint32_t c = pd.rcnt[tid];
                pd.rbase[tid] = c ? atomicAdd(&p.cand_cnt[q0 + tid], c) : 0u;
                pd.rcnt[tid] = 0;
            }
            __syncthreads();
        }
        uint32_t pos[U];
#pragma unroll
        for (int u = 0; u < U; ++u) pos[u] = (l == 0 && e0 + u * G + g < total) ? atomicAdd(&pd.rbase[loc[u]], 1u) : ~0u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t si = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t wa = va[u][w] ^ 0x80808080u, wb = vb[u][w] ^ 0x80808080u;      // value = stored byte with its top bit flipped, as an unsigned byte
                si = __builtin_amdgcn_udot4(wa, wb, si, false);
            }
            si += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)si, 0xB1, 0xf, 0xf, true);                            // quad_perm [1,0,3,2]
            if constexpr (L >= 4) si += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)si, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
            if constexpr (L >= 8) si += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)si, 0x141, 0xf, 0xf, true);     // row_half_mirror
            if (pos[u] < p.cap) {                                       // (~0 for idle lanes and groups; at or past cap: the list overflowed, k_select rescans)
                const uint32_t row = loc[u];
                const float dist = fmaf(-2.f, (float)si, sA[2 * row] + bnv[u]);          // x.y below 2^24: exact; the same fp32 expression as every other path
                p.cand[(q0 + row) * p.cap + pos[u]] = make_key(dist < 0.f ? 0.f : dist, id[u]);
            }
        }
    }
    __syncthreads();                                                    // the rings may be written again
}

template <int D, size_t SMEM_BYTES>
__device__ __forceinline__ void tile8_walk(const TileArgs &p, const uint32_t group, char *smem, float *stage, Pend16 &pend, const uint32_t qt, const uint32_t grp) {
    constexpr int TM = 128, TN = 128, MI = 2, NJ = 2, STEPS = D / 32, PITCH = D + AUX8;
    constexpr uint32_t BUFB = TN * PITCH, PIECES = BUFB / 16, SWEEPS = PIECES / 256, REM = PIECES % 256;       // D = 128: 4 x 256 + 128 sixteen-byte pieces per tile
    static_assert(D % 32 == 0 && D <= 128 && PITCH % 32 == 16 && REM % 64 == 0, "rows of whole 32-deep k-steps, odd pitch in 16-byte units");
    constexpr uint32_t RING_ROOM = (uint32_t)((SMEM_BYTES - 2 * (size_t)BUFB) / (4 * sizeof(uint2)));
    constexpr uint32_t RCAP = RING_ROOM >= 1024 ? 1024u : RING_ROOM >= 512 ? 512u : 256u;          // records per wave
    static_assert(RING_ROOM >= 256, "a ring takes at least two tiles' worth of records");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;            // 2 x 2 waves of 64 query rows x 64 columns (2 x 2 blocks of 32 x 32)
    const uint32_t n_ct = (uint32_t)((p.nb_count + TN - 1) / TN);
    const uint32_t ct0 = grp * group, ct1 = ct0 + group < n_ct ? ct0 + group : n_ct;
    const size_t q0 = (size_t)qt * TM;
    const uint32_t q_valid = (uint32_t)(p.nq - q0 < (size_t)TM ? p.nq - q0 : (size_t)TM);
    uint2 *const ring = reinterpret_cast<uint2 *>(smem + 2 * (size_t)BUFB) + (size_t)wave * RCAP;
    if (tid < TM) pend.rcnt[tid] = 0;
    if (tid == 0) pend.n = 0;
    // (buffers picked by selects on ONE condition, with constant offsets: that is what lets the compiler's alias analysis see that the copies in flight
    // never touch the tile being read -- a computed buffer index costs an s_waitcnt vmcnt(0), i.e. the whole copy, in front of every fragment read)
    char *const B0 = smem, *const B1 = smem + BUFB;
    // a column tile is PIECES consecutive 16-byte pieces of the image (padded by one tile of zero rows) copied as such by LDS-DMA: lane t moves pieces
    // t, t + 256, ...; one wave-instruction fills 1 KiB of LDS from its wave-uniform base.  The source is a wave-uniform base plus a 32-bit lane
    // offset (scalar-base addressing: no 64-bit vector arithmetic per copy).
    const char *const img = reinterpret_cast<const char *>(p.xb8) + p.nb_first * (size_t)PITCH;
    const uint32_t lane_off = (uint32_t)tid * 16u;
    // (the remainder sweep covers REM pieces: every wave issues it, the upper waves repeating the lower ones' pieces -- the same bytes to the same place --
    // so that no copy sits behind a branch: a branch ends the scheduling region the matrix instructions and the sweep are interleaved in)
    auto stage_sweep = [&](uint32_t ct, char *buf, uint32_t it) {
        const char *src = img + (size_t)ct * BUFB + 4096u * it;         // wave-uniform
        if (it < SWEEPS) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + lane_off),
                                             (__attribute__((address_space(3))) void *)(buf + wave * 1024 + 4096 * it), 16, 0, 0);
        } else if (REM) {
            const int w = wave % (int)(REM / 64);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (uint32_t)(w * 1024 + lane * 16)),
                                             (__attribute__((address_space(3))) void *)(buf + w * 1024 + 4096 * it), 16, 0, 0);
        }
    };
    // the walk's first tile is requested before anything else: one round trip to memory for the whole prologue
#pragma unroll
    for (uint32_t it = 0; it <= SWEEPS; ++it) stage_sweep(ct0, B0, it);
    // the query operand never changes during the walk: lane l holds row l & 31 of each 32-row block, 16 consecutive k of every 32-deep step
    i32x4v afrag[MI][STEPS];
    auto load_afrag = [&]() {
        const char *abase = reinterpret_cast<const char *>(p.xq8 + q0 * (size_t)D);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const uint32_t r = wm + 32 * i + (lane & 31);
            const char *row = abase + (size_t)(r < q_valid ? r : q_valid - 1) * D + (lane >> 5) * 16;
#pragma unroll
            for (int ks = 0; ks < STEPS; ++ks) afrag[i][ks] = *reinterpret_cast<const i32x4v *>(row + ks * 32);
        }
    };
    load_afrag();
    // row halves of the thresholds (flat_tile16.hpp: tile16_walk): r0 = -(floor(R / 2) + 1), R = |x|^2 - ceil(tau) - 256 sum x' - 32768 d.
    // |S| < 2^21, |r0| < 2^26, |c0| < 2^24: the sentinels +-2^29 of "everything passes" / "nothing passes" stay clear of every sum.
    if (tid < TM) {
        int r0 = -(1 << 29);                                           // rows past nq: nothing passes
        float row_qn = 0.f;
        if (q0 + tid < p.nq) {
            row_qn = p.qn[q0 + tid];
            const float row_tau = p.tau[q0 + tid];
            if (row_tau == INFINITY) r0 = 1 << 29;                     // fewer than k results so far: everything passes
            else {
                const uint32_t *w = reinterpret_cast<const uint32_t *>(p.xq8 + (q0 + tid) * (size_t)D);
                int sx = 0;
#pragma unroll 8
                for (int t = 0; t < D / 4; ++t) sx = __builtin_amdgcn_sdot4((int)w[t], 0x01010101, sx, false);
                const int R = (int)row_qn - (int)ceilf(row_tau) - 256 * sx - 32768 * D;
                r0 = -(R >> 1) - 1;                                    // (>> of a negative int: floor)
            }
        }
        stage[2 * tid] = row_qn;                                       // the flush's |x|^2
        reinterpret_cast<int *>(stage)[3 * TM + tid] = r0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // C[row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][column = lane & 31] of a 32 x 32 block: the row halves in the accumulators' own layout -- the
    // first matrix instruction of a tile takes them as its C operand, so the accumulators come out as S + r0 with no initialisation pass
    i32x16v r0t[MI];
    auto load_r0t = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) r0t[i][r] = reinterpret_cast<const int *>(stage)[3 * TM + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
    };
    load_r0t();
#ifdef PF_FLAT_STAMPS
    const bool fs_on = p.nb_count >= 400000 && blockIdx.x >= 256 && blockIdx.x < 256 + PF_FS_WGS;
#endif
    uint32_t rc = 0;                                                    // records in this wave's ring (wave-uniform)
    // The walk is software-pipelined by HALF tiles: while the matrix pipe works on one column block of the wave's tile (8 matrix instructions into 32
    // accumulator registers), the vector pipe sweeps the other block's finished accumulators -- one matrix instruction, then the eight vector
    // instructions of four accumulator values, eight times per half -- so that ONE wave keeps both pipes busy with the 64 accumulator registers it
    // always had.  (Round 3's walk ran the two phases one after the other and relied on its partner wave on the SIMD for the overlap; phase stamps
    // showed the partners in lockstep: 3 854 cycles per tile for 512 cycles of matrix work.)
    //   half 0 of tile t:  matrix instructions -> acc[.][0]   beside   sweep of acc[.][1] = tile t - 1's second column block
    //   half 1 of tile t:  matrix instructions -> acc[.][1]   beside   sweep of acc[.][0] = this tile's first column block
    // The copies of tile t + 1 are requested between the matrix instructions as before, waited for before the tile's one barrier.
    auto sweep4 = [&](const i32x16v &a, int c0, uint32_t &fail, int r4) {     // the value passes where S + r0 + c0 >= 0: sign bits shifted into the lane's word
#pragma unroll
        for (int r = 4 * r4; r < 4 * r4 + 4; ++r) fail = __builtin_amdgcn_alignbit(fail, (uint32_t)(a[r] + c0), 31);
    };
    // verdict words of tile t (set bit = fails; row 16 i + r in bit 31 - (16 i + r)) -> records in the wave's ring: a ballot and an LDS write per
    // column block with a survivor
    auto append = [&](uint32_t fail0, uint32_t fail1, uint32_t t, bool valid) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const size_t col = (size_t)t * TN + wn + 32 * jj + (lane & 31);
            const uint32_t s1 = (valid && col < p.nb_count) ? ~(jj ? fail1 : fail0) : 0u;      // columns past the end of the chunk belong to the next one
            const uint64_t m = __ballot(s1 != 0);
            if (m) {                                                    // wave-uniform
                const uint32_t slot = rc + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                // (the clamp never acts -- a ring is worked off while it has room for a tile's 128 records -- it bounds the address for the alias
                // analysis: an append must not wait for the copies in flight)
                if (s1) ring[slot < RCAP - 1 ? slot : RCAP - 1] = make_uint2(s1, ((t - ct0) << 8) | ((uint32_t)jj << 6) | (uint32_t)lane);
                rc += (uint32_t)__popcll(m);
            }
        }
    };
    i32x16v acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) { acc[i][0] = r0t[i]; acc[i][1] = r0t[i]; }     // (defined values for the first half's sweep, whose words are dropped)
    uint32_t fail0 = 0;                                                 // first column block's word of the tile before (finished in its half 1)
    int c0p1 = 0;                                                       // ... and the column half of its second block
    // 8 matrix instructions of column block JJ of the tile in `buf` beside the sweep of block 1 - JJ (column half c0s) into `fail`
    auto half = [&](auto JJc, const char *buf, int c0s, uint32_t &fail, uint32_t td, char *bufD) {
        constexpr int JJ = decltype(JJc)::value;
        const char *fb = buf + (wn + 32 * JJ + (lane & 31)) * PITCH + (lane >> 5) * 16;
        i32x4v b[STEPS];
#pragma unroll
        for (int ks = 0; ks < STEPS; ++ks) b[ks] = *reinterpret_cast<const i32x4v *>(fb + ks * 32);
        constexpr int NM = MI * STEPS;                                  // matrix instructions of the half; 8 sweep groups (4 values each) spread over them
        __builtin_amdgcn_sched_barrier(0);
        int g = 0;
#pragma unroll
        for (int ks = 0; ks < STEPS; ++ks)
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                acc[i][JJ] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[i][ks], b[ks], ks == 0 ? r0t[i] : acc[i][JJ], 0, 0, 0);
                const int m = ks * MI + i, g_end = (8 * (m + 1) + NM - 1) / NM;          // groups due after matrix instruction m
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (g + e < g_end) sweep4(acc[(g + e) >> 2][1 - JJ], c0s, fail, (g + e) & 3);
                g = g_end;
                // the copies of tile t + 1: half 0 issues the first sweeps, half 1 the rest, one behind a matrix instruction each
                constexpr uint32_t NS = SWEEPS + (REM ? 1 : 0), S0 = (NS + 1) / 2;
                const uint32_t it = JJ == 0 ? (uint32_t)m : S0 + (uint32_t)m;
                if (it < (JJ == 0 ? S0 : NS)) stage_sweep(td, bufD, it);
                if (m == NM - 1) {                                      // (more sweeps than matrix instructions: short rows)
#pragma unroll
                    for (uint32_t it2 = (JJ == 0 ? 0 : S0) + NM; it2 < (JJ == 0 ? S0 : NS); ++it2) stage_sweep(td, bufD, it2);
                }
            }
        asm volatile("" : "+v"(fail));                                  // (the word is complete HERE: without this the compiler sinks the whole sweep down to the append that reads it)
        // the order the scheduler is to emit: a matrix instruction, then the vector instructions of one sweep group (and whatever address
        // arithmetic the copies need), NM times -- left to itself hipcc emits the matrix instructions back to back and the sweep behind them
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, (64 + NM - 1) / NM + 1, 0);     // VALU
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (uint32_t ct = ct0; ct < ct1; ++ct) {
        const bool odd = ((ct - ct0) & 1u) != 0;
        char *const buf_cur = odd ? B1 : B0, *const buf_nxt = odd ? B0 : B1;
        PF_FSTAMP(0);
        const uint32_t td = ct + 1 < ct1 ? ct + 1 : ct;                 // (the walk's last tile requests itself again: harmless, and no branch around the copies)
        const char *fbx = buf_cur + (wn + (lane & 31)) * PITCH;
        int c0v[NJ];
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) c0v[jj] = *reinterpret_cast<const int *>(fbx + 32 * jj * PITCH + D);
        PF_FSTAMP(1);
        uint32_t fail1 = 0;
        half(std::integral_constant<int, 0>{}, buf_cur, c0p1, fail1, td, buf_nxt);
        PF_FSTAMP(2);
        append(fail0, fail1, ct - 1, ct > ct0);
        PF_FSTAMP(3);
        fail0 = 0;
        half(std::integral_constant<int, 1>{}, buf_cur, c0v[0], fail0, td, buf_nxt);
        c0p1 = c0v[1];
        if (lane == 0) pend.wcnt[(ct - ct0) & 1u][wave] = rc;          // (read by every wave after the barrier)
        PF_FSTAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's pieces of tile ct + 1 have landed
        __syncthreads();                                                // the tile's one barrier: the other buffer is complete, nobody reads this one any more
        PF_FSTAMP(5);
        const uint32_t *wc = pend.wcnt[(ct - ct0) & 1u];
        const uint32_t c01 = wc[0] > wc[1] ? wc[0] : wc[1], c23 = wc[2] > wc[3] ? wc[2] : wc[3];
        if ((c01 > c23 ? c01 : c23) > RCAP - 128) {                     // workgroup-uniform (every wave read the same four counts); a tile adds at most 128 records to a ring
            uint32_t none[PF_FLAT_MT][NJ] = {};
            pend16_flush<D, PF_FLAT_MT, NJ, TN, true, true>(p, pend, stage, q0, tid, none, ct0, wm, wn, false, nullptr, q_valid, false, 0, ring, rc);
            rc = 0;
            // (64 registers that need not live across the flush: fetched again -- a flush in mid-walk is rare where the walk is long)
            load_afrag();
            load_r0t();
#pragma unroll
            for (int i = 0; i < MI; ++i) {                              // (waited for here, not by an s_waitcnt vmcnt(0) behind the next tile's copy requests)
#pragma unroll
                for (int ks = 0; ks < STEPS; ++ks) asm volatile("" : "+v"(afrag[i][ks]));
                asm volatile("" : "+v"(r0t[i]));
            }
        }
    }
    // the last tile's second column block, then whatever the rings hold
    {
        uint32_t fail1 = 0;
#pragma unroll
        for (int g = 0; g < 8; ++g) sweep4(acc[g >> 2][1], c0p1, fail1, g & 3);
        append(fail0, fail1, ct1 - 1, true);
        uint32_t none[PF_FLAT_MT][NJ] = {};
        pend16_flush<D, PF_FLAT_MT, NJ, TN, true, true>(p, pend, stage, q0, tid, none, ct0, wm, wn, false, nullptr, q_valid, true, 0, ring, rc);
    }
}

}  // namespace pf
