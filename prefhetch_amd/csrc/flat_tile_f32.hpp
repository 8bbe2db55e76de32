// flat_tile_f32.hpp -- fp32 distance tiles on the f32 matrix pipe and the filtering epilogue (k_l2_tile)
// (part of the pre-filter translation unit pf_flat.hip: included there, in order; gfx950 only)
#pragma once
#include "flat_common.hpp"

namespace pf {

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
    if (p.only_flagged && !(p.q_inexact[q0 / 128] & 2u)) return;   // (beside the slab tiles: this tile is theirs; workgroup-uniform)
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

}  // namespace pf
