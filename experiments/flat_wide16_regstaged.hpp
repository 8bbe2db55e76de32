// flat_wide16.hpp -- rows longer than 256 values: bf16 tiles with BOTH operands staged through LDS in 64-deep k-slabs (k_l2_wide16) as a
// conservative filter, then the k-ordered fp32 chain over every candidate it let through (k_wide_fixup)
// (part of the pre-filter translation unit pf_flat.hip: included there, in order; gfx950 only)
//
// Up to 256 values a wave of k_l2_tile16 keeps its query fragments in registers for the whole walk; beyond, they do not fit (d / 4 registers per
// 32 query rows).  Here a 128 x 128 tile is a plain blocked product: per 64-deep slab, 128 query rows and 128 base rows of the bf16 images go
// global -> registers -> LDS (row-major, 16-byte chunks, no transposition: the matrix instruction's fragment IS 16 consecutive bytes of a row), two
// slabs in flight, one barrier per slab, v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The images are the nearest bf16 of every value, so the
// accumulator is x.y up to |x~.y~ - x.y| <= (2^-9 + 2^-19)(|x|^2 + |y|^2) plus the accumulation's own rounding; the epilogue (l2_tile_epilogue, the
// fp32 tiles' own) tests  (|x|^2 + |y|^2)(1 - m) - 2 acc <= tau  with m = 2.1 x 2^-8 + d x 2^-20 (d x 2^-20 alone when every value of the base and of the
// query tile is exactly representable -- 8-bit data: only the two accumulations' rounding is left), which never drops a row the fp32 chain would keep,
// and appends (approximate distance, id) keys to the query's candidate list.  k_wide_fixup then recomputes the distance of every entry of the
// lists with the same k-ordered fmaf chain and final expression as every other path (one lane per entry) and rewrites the key in place: what the
// merge sees is bit for bit what the fp32 tiles would have appended, plus a few rows beyond tau that cannot enter the top k.
#pragma once
#include "flat_tile_f32.hpp"

namespace pf {

constexpr int WK = 64;                                  // slab depth in values
constexpr int WPITCH = WK * 2 + 16;                     // bytes per LDS row: 9 x 16 -- the 16-byte fragment reads of 16 consecutive rows cover all 64 banks once
constexpr size_t WIDE_LDS = 2 * 2 * 128 * (size_t)WPITCH;      // two stages x (queries, base rows) x 128 rows = 72 KiB: two workgroups per CU
constexpr float WIDE_MARGIN = 2.1f * 0x1p-8f;

// row length of the wide images: d padded with zeros to whole slabs
inline uint32_t wide_row_length(uint32_t d) { return (d + WK - 1) / WK * WK; }

// bf16 image of rows of any length: out[r][0 .. dpw) = nearest-even bf16 of x[r][0 .. d), zeros behind; one thread per 8 output values
// inexact[r / rows_per_flag] (rows_per_flag 0: one word) gets bit 0 when a value of the row is not exactly representable
__global__ void __launch_bounds__(256) k_rows_bf16(const float *__restrict__ x, size_t n, uint32_t d, uint16_t *__restrict__ out, uint32_t dpw,
                                                   uint32_t *__restrict__ inexact, uint32_t rows_per_flag) {
    const uint32_t per = dpw / 8;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, r = t / per;
    if (r >= n) return;
    const uint32_t k0 = (uint32_t)(t - r * per) * 8;
    const float *row = x + r * (size_t)d;
    u32x4 w;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t h[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t k = k0 + 2 * j + e;
            const float v = k < d ? row[k] : 0.f;
            h[e] = bf16_rne(v);
            bad = bad || !bf16_exact(v);
        }
        w[j] = h[0] | (h[1] << 16);
    }
    *reinterpret_cast<u32x4 *>(out + r * (size_t)dpw + k0) = w;
    if (inexact) {
        // one word may take every thread of the launch (725 ms for a 1M x 512 base when each asked for its own atomic): a wave settles its words
        // one at a time -- a leader looks first and asks for the atomic only when the bit is still clear
        uint32_t idx = bad ? (uint32_t)(rows_per_flag ? r / rows_per_flag : 0) : 0xFFFFFFFFu;
        for (;;) {
            const uint64_t m = __ballot(idx != 0xFFFFFFFFu);
            if (m == 0) break;                                  // wave-uniform
            const int leader = __ffsll((long long)m) - 1;
            const uint32_t lidx = (uint32_t)__shfl((int)idx, leader);
            if ((int)(threadIdx.x & 63) == leader && !(*reinterpret_cast<volatile uint32_t *>(inexact + lidx) & 1u)) atomicOr(&inexact[lidx], 1u);
            if (idx == lidx) idx = 0xFFFFFFFFu;
        }
    }
}

// row norms (the fp32 fma chain in index order: k_row_norms' number) of a FEW long rows -- the queries of a search: one wave per row.  The chain is
// serial, but a thread per row (k_row_norms) also reads its row alone, 4 bytes at a time: 66 us for 1024 rows of 512 values.  Here the wave copies
// the row into LDS (coalesced, segments of 2048 values) and its first lane chains it from there.
__global__ void __launch_bounds__(256) k_row_norms_wave(const float *__restrict__ x, size_t n, uint32_t d, float *__restrict__ out) {
    constexpr uint32_t SEG = 2048;
    __shared__ float seg[4][SEG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t r = (size_t)blockIdx.x * 4 + wave;
    if (r >= n) return;                                         // wave-uniform
    const float *row = x + r * (size_t)d;
    float acc = 0.f;
    for (uint32_t k0 = 0; k0 < d; k0 += SEG) {
        const uint32_t kn = d - k0 < SEG ? d - k0 : SEG;
        for (uint32_t k = lane; k < kn; k += 64) seg[wave][k] = row[k0 + k];
        wave_sync();
        if (lane == 0)
            for (uint32_t k = 0; k < kn; ++k) acc = fmaf(seg[wave][k], seg[wave][k], acc);
        wave_sync();
    }
    if (lane == 0) out[r] = acc;
}

// 128 rows x one slab: thread t moves the 16-byte chunks c = t + 256 it (row c / 8, chunk c % 8); rows past the end re-read the last valid row
__device__ __forceinline__ void wide_fetch(u32x4 (&v)[4], const uint16_t *__restrict__ img, size_t row0, size_t rows_valid, uint32_t dpw, uint32_t k0, int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const uint32_t c = (uint32_t)tid + 256u * it, row = c >> 3, kc = c & 7u;
        const size_t rr = row < rows_valid ? row : rows_valid - 1;
        v[it] = *reinterpret_cast<const u32x4 *>(img + (row0 + rr) * (size_t)dpw + k0 + 8 * kc);
    }
}
__device__ __forceinline__ void wide_commit(char *lds, const u32x4 (&v)[4], int tid) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const uint32_t c = (uint32_t)tid + 256u * it, row = c >> 3, kc = c & 7u;
        *reinterpret_cast<u32x4 *>(lds + row * WPITCH + kc * 16) = v[it];
    }
}

__global__ void __launch_bounds__(256, 2) k_l2_wide16(TileArgs p, const uint16_t *__restrict__ q16w, const uint16_t *__restrict__ x16w, uint32_t dpw, float margin,
                                                      float margin_exact) {
    using GEO = GeoBatch;
    __shared__ __align__(16) char smem[WIDE_LDS];
    auto sA = [&](uint32_t stage) { return smem + stage * (128 * WPITCH); };              // stages 0 / 1 of the query slab, then of the base slab
    auto sB = [&](uint32_t stage) { return smem + (2 + stage) * (128 * WPITCH); };
    // XCD-aware tile order, as k_l2_tile: XCD x takes the column tiles = x (mod 8) and runs all query tiles of a column tile back to back
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const uint32_t qt = j % p.n_qtiles, ct = (j / p.n_qtiles) * 8 + xcd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t q0 = (size_t)qt * 128, c0 = (size_t)ct * 128;
    if (c0 >= p.nb_count) return;
    // the query tile's word: bit 1 -- the filter does not separate for this tile (predicted by the bootstrap selection or found out by a list that
    // overflowed: flat_select.hpp): the fp32 tiles take it (k_l2_tile, launched behind this kernel for the flagged tiles only); bit 0 -- a value of
    // the tile is not exactly representable.  Exact operands on both sides leave only the accumulation's rounding to cover.
    const uint32_t flags = p.q_inexact[qt];
    if (flags & 2u) return;                                     // workgroup-uniform
    if (p.base_exact && !(flags & 1u)) margin = margin_exact;
    const size_t q_valid = p.nq - q0 < 128 ? p.nq - q0 : 128, c_valid = p.nb_count - c0 < 128 ? p.nb_count - c0 : 128;
    const int wm = (wave / 2) * 64, wn = (wave % 2) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;
    // the epilogue's operands, requested now: the norms of this lane's columns and, for the first 128 threads, one query row's (norm, threshold) --
    // both scaled by (1 - margin): the epilogue then tests the conservative expression.  A query whose norm is not finite lets everything through
    // (its list overflows and the merge rescans the chunk exactly); columns past the end of the chunk compare false (NaN).
    const float scale = 1.f - margin;
    size_t col[2]; bool col_ok[2]; float bnv[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        col[jj] = c0 + wn + 32 * jj + (lane & 31);
        col_ok[jj] = col[jj] < p.nb_count;
        bnv[jj] = col_ok[jj] ? p.bn[p.nb_first + col[jj]] * scale : __builtin_nanf("");
    }
    float row_qn = 0.f, row_tau = -INFINITY;                    // rows past nq: nothing passes
    if (tid < 128 && q0 + tid < p.nq) {
        row_qn = p.qn[q0 + tid] * scale; row_tau = p.tau[q0 + tid];
        if (!(fabsf(row_qn) < INFINITY)) { row_qn = 0.f; row_tau = INFINITY; }
    }
    u32x4 ra[4], rb[4];
    wide_fetch(ra, q16w, q0, q_valid, dpw, 0, tid);
    wide_fetch(rb, x16w, p.nb_first + c0, c_valid, dpw, 0, tid);
    wide_commit(sA(0), ra, tid);
    wide_commit(sB(0), rb, tid);
    if (WK < dpw) {
        wide_fetch(ra, q16w, q0, q_valid, dpw, WK, tid);
        wide_fetch(rb, x16w, p.nb_first + c0, c_valid, dpw, WK, tid);
    }
    __syncthreads();
    // slab s feeds the matrix pipe from stage s & 1 while slab s + 1 (in registers since the iteration before) is committed to the other stage and
    // slab s + 2 is requested: one barrier per slab
    for (uint32_t k0 = 0, cur = 0; k0 < dpw; k0 += WK, cur ^= 1) {
        if (k0 + WK < dpw) {
            wide_commit(sA(cur ^ 1), ra, tid);
            wide_commit(sB(cur ^ 1), rb, tid);
            if (k0 + 2 * WK < dpw) {
                wide_fetch(ra, q16w, q0, q_valid, dpw, k0 + 2 * WK, tid);
                wide_fetch(rb, x16w, p.nb_first + c0, c_valid, dpw, k0 + 2 * WK, tid);
            }
        }
        // lane l: row l & 31 of each 32-row block, 8 consecutive k of every 16-deep step starting at 8 (l >> 5): 16 bytes of the row
        const char *fa = sA(cur) + (wm + (lane & 31)) * WPITCH + (lane >> 5) * 16, *fb = sB(cur) + (wn + (lane & 31)) * WPITCH + (lane >> 5) * 16;
        bf16x8 a[2][2], b[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { a[0][i] = *reinterpret_cast<const bf16x8 *>(fa + 32 * i * WPITCH); b[0][i] = *reinterpret_cast<const bf16x8 *>(fb + 32 * i * WPITCH); }
#pragma unroll
        for (int ks = 0; ks < WK / 16; ++ks) {
            const int c = ks & 1, n = c ^ 1;
            if (ks + 1 < WK / 16) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[n][i] = *reinterpret_cast<const bf16x8 *>(fa + 32 * i * WPITCH + (ks + 1) * 32);
                    b[n][i] = *reinterpret_cast<const bf16x8 *>(fb + 32 * i * WPITCH + (ks + 1) * 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the reads ahead of the matrix instructions they overlap with
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c][i], b[c][jj], acc[i][jj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    l2_tile_epilogue<true, GEO, false>(p, acc, reinterpret_cast<float *>(smem), q0, wm, tid, col, col_ok, bnv, row_qn, row_tau);
}

// Every entry of the candidate lists gets its distance from the k-ordered fp32 chain (the accumulator of v_mfma_f32_32x32x2_f32, bit for bit)
// and the expression every path ends in.  The chain is serial in k, so an entry is one lane's work -- but a lane reading its own row 16 bytes at a
// time makes every load instruction 64 line requests and the wave one long string of exposed round trips (first version: 1.2 ms per call at
// d = 1024 whatever the number of entries).  Here a wave takes 64 entries of one query and moves their rows through LDS in slabs of 32 values:
// eight lanes read the 128 bytes of a row together (8 rows = 16 whole lines per instruction, the next slab already requested into registers while
// this one is evaluated), each lane then reads ITS row back from LDS (row pitch 144 bytes: conflict-free 16-byte reads).  Wave-private: no barrier.
constexpr int FX_SLAB = 32, FX_PITCH = FX_SLAB * 4 + 16;        // values per slab; bytes per LDS row
__global__ void __launch_bounds__(256) k_wide_fixup(TileArgs p, uint32_t blocks_per_query) {
    __shared__ __align__(16) char lds[4][64 * FX_PITCH];
    const uint32_t q = blockIdx.x / blocks_per_query;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t e0 = (blockIdx.x - q * blocks_per_query) * 256u + 64u * (uint32_t)wave;
    const uint32_t cnt = p.cand_cnt[q] < p.cap ? p.cand_cnt[q] : p.cap;
    if (e0 >= cnt) return;                                      // wave-uniform
    const uint32_t e = e0 + (uint32_t)lane < cnt ? e0 + (uint32_t)lane : cnt - 1;      // lanes past the end repeat the last entry (and write nothing)
    uint64_t *slot = p.cand + (size_t)q * p.cap + e;
    const uint32_t id = (uint32_t)*slot;
    const float *x = p.xq + (size_t)q * p.d;
    float acc = 0.f;
    const uint32_t d = p.d;
    if ((d & 3u) == 0) {
        char *mine = lds[wave];
        // instruction i of a slab: lane l fetches 16 bytes of row 8 i + (l >> 3) at chunk l & 7
        const float *rowp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) rowp[i] = p.xb + (size_t)(uint32_t)__shfl((int)id, 8 * i + (lane >> 3)) * d + 4 * (lane & 7);
        const uint32_t nslab = (d + FX_SLAB - 1) / FX_SLAB;
        u32x4 v[8];
        auto fetch = [&](uint32_t s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t k = s * FX_SLAB + 4 * ((uint32_t)lane & 7u);
                v[i] = k < d ? *reinterpret_cast<const u32x4 *>(rowp[i] + s * FX_SLAB) : u32x4{0u, 0u, 0u, 0u};        // (the last slab of a row length that is not a multiple of 32)
            }
        };
        fetch(0);
        for (uint32_t s = 0; s < nslab; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4 *>(mine + (8 * i + (lane >> 3)) * FX_PITCH + (lane & 7) * 16) = v[i];
            wave_sync();
            if (s + 1 < nslab) fetch(s + 1);
            const uint32_t k0 = s * FX_SLAB, kn = d - k0 < FX_SLAB ? d - k0 : FX_SLAB;      // values of this slab (a multiple of 4)
            const char *row = mine + lane * FX_PITCH;
#pragma unroll
            for (int j = 0; j < FX_SLAB / 4; ++j) {
                if (4u * j < kn) {                                  // wave-uniform
                    const float4 yv = *reinterpret_cast<const float4 *>(row + 16 * j);
                    const float4 xv = *reinterpret_cast<const float4 *>(x + k0 + 4 * j);        // one address for the whole wave
                    acc = fmaf(xv.x, yv.x, acc); acc = fmaf(xv.y, yv.y, acc);
                    acc = fmaf(xv.z, yv.z, acc); acc = fmaf(xv.w, yv.w, acc);
                }
            }
            wave_sync();                                            // the slab is read before the next one overwrites it
        }
    } else {
        const float *y = p.xb + (size_t)id * d;
        for (uint32_t k = 0; k < d; ++k) acc = fmaf(x[k], y[k], acc);
    }
    const float dist = fmaf(-2.f, acc, p.qn[q] + p.bn[id]);
    if (e0 + (uint32_t)lane < cnt) *slot = make_key(dist < 0.f ? 0.f : dist, id);
}

}  // namespace pf
