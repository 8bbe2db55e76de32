// flat_flush16.hpp -- survivor list of the 16-bit / 8-bit tile walks: parking, row reservations, exact re-evaluation (pend16_flush)
// (part of the pre-filter translation unit pf_flat.hip: included there, in order; gfx950 only)
#pragma once
#include "flat_common.hpp"

namespace pf {

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
#ifndef PF_APPROX_SLABS
#define PF_APPROX_SLABS 0       // experiment: the last flush of an inexact walk staged slab by slab through LDS by LDS-DMA, 64 chains per wave -- 0.644 against 0.61 ms on N(0,1): the copies (16 line requests per survivor) cost what the per-lane loads did; off
#endif
#ifndef PF_APPROX_UNROLL2
#define PF_APPROX_UNROLL2 4     // ... both rows from memory
#endif
#define PF_FLUSH_INLINE __forceinline__
// RING: the verdict words come from this wave's ring of records in LDS instead (the int8 walk appends a (word, tile, column block, lane) record per
// non-zero word as the tile ends -- a ballot and an LDS write, no barrier -- and calls this only when a ring is nearly full or the walk ends:
// the parking above cost a quarter of the int8 walk's time); rc records, ct_base = the walk's first tile.
// LAST: this instantiation is only ever the walk's last flush (compiled with the slab-wise evaluation of inexact survivors)
// PADDED: the image rows are the data's rows padded with zeros (p.d < D): the fp32 rows the inexact survivors are evaluated on then have a run-time
// length.  A separate instantiation: with the run-time forms merely present beside the compile-time ones, the N(0,1) search at d = 128 lost 8 %
// (0.64 -> 0.69 ms: more code and registers in a kernel at its limits).
template <int D, int MT, int NJ, int TN, bool I8 = false, bool RING = false, bool LAST = false, bool PADDED = false>
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
            // (the fp32 rows have p.d values; D is the images' row length, p.d padded with zeros.  Rows whose length is not a multiple of 4 are not
            // 16-byte aligned: they are read value by value.)
            const uint32_t dv = PADDED ? p.d : (uint32_t)D;
            const bool vec4 = !PADDED || (dv & 3u) == 0;
            if (tid < 128) {
                const uint32_t c = pd.rcnt[tid];
                pd.rbase[tid] = c ? atomicAdd(&p.cand_cnt[q0 + tid], c) : 0u;
                pd.rcnt[tid] = 0;
            }
            constexpr bool SLABS_OK = PF_APPROX_SLABS && LAST && D % 64 == 0 && (((2u * BUF) / 4u) & ~15u) >= 16384u;
            if (SLABS_OK && dv == (uint32_t)D) {                          // (a compile-time false unless the experiment is switched on)
                // The walk's last flush (both tile buffers free).  One lane per survivor reading its own two rows from memory made the texture path
                // see 64 different cache lines per load instruction: 82 us of a search on N(0,1) data (round 3's ablation).  Now every WAVE takes 64
                // survivors at a time and walks their rows in slabs of 16 values: the slab's 64-byte pieces of the 64 query rows and the 64 base rows
                // are copied by LDS-DMA (four row pieces per 16 lanes of a copy: whole cache lines) into the wave's own quarter of the buffers while
                // the chains of the slab before run out of the other half of it -- all 64 lanes busy, no registers for staging, no workgroup barrier.
                // LDS image of a slab: [x rows | y rows][64 rows][4 x 16 bytes], the four pieces of row r stored at position piece ^ ((r >> 2) & 3):
                // 16 lanes reading their rows' same piece then hit 16 different 16-byte bank groups.
                constexpr uint32_t QUARTER = ((2u * BUF) / 4u) & ~15u, SLABS = D / 16;
                __syncthreads();                                          // the rows' ranges are reserved
                const uint32_t lane = (uint32_t)tid & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);
                char *const S0 = xstage + wv * QUARTER, *const S1 = S0 + 8192;
                for (uint32_t eb = wv * 64u; eb < n; eb += 256u) {        // (per wave: the trip counts differ, no barrier inside)
                    // the rows this lane copies: piece c of the image = rows 16 (c & 3) + (lane >> 2) of x (c < 4) or y, 16-byte part (lane & 3) of its slab
                    uint32_t rix[8];                                      // (row numbers, not addresses: 8 registers through the slab loop instead of 16)
#pragma unroll
                    for (uint32_t c = 0; c < 8; ++c) {
                        const uint32_t r = 16u * (c & 3u) + (lane >> 2);
                        uint32_t e = eb + r;
                        e = e < n ? e : n - 1;
                        rix[c] = c < 4 ? (uint32_t)q0 + pd.loc[e] : pd.id[e];
                    }
                    const uint32_t part = 4u * ((lane & 3u) ^ ((lane >> 4) & 3u));     // this lane's 16-byte part of a slab: (lane & 3) ^ ((r >> 2) & 3), r = 16 (c & 3) + (lane >> 2)
                    const uint32_t e = eb + lane < n ? eb + lane : n - 1, row = pd.loc[e], id = pd.id[e];
                    const float bnv = p.bn[id];
                    auto copy_slab = [&](uint32_t sl, char *buf) {
#pragma unroll
                        for (uint32_t c = 0; c < 8; ++c) {
                            const float *srcp = (c < 4 ? p.xq : p.xb) + (size_t)rix[c] * D + part + 16u * sl;
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)srcp,
                                                             (__attribute__((address_space(3))) void *)(buf + 1024u * c), 16, 0, 0);
                        }
                    };
                    copy_slab(0, S0);
                    float acc = 0.f;
                    for (uint32_t sl = 0; sl < SLABS; ++sl) {
                        const bool odd = (sl & 1u) != 0;
                        char *const cur = odd ? S1 : S0, *const nxt = odd ? S0 : S1;   // (selects on one condition: the alias analysis sees that the copy in flight never touches the slab being read)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this slab has landed (wave-private: nobody else to wait for)
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        if (sl + 1 < SLABS) copy_slab(sl + 1, nxt);
                        const char *xr = cur + lane * 64u, *yr = cur + 4096u + lane * 64u;
#pragma unroll
                        for (uint32_t t = 0; t < 4; ++t) {
                            const uint32_t pos = (t ^ ((lane >> 2) & 3u)) * 16u;
                            const float4 a = *reinterpret_cast<const float4 *>(xr + pos), b = *reinterpret_cast<const float4 *>(yr + pos);
                            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    }
                    if (eb + lane < n) {
                        const uint32_t pos = atomicAdd(&pd.rbase[row], 1u);
                        if (pos < p.cap) {                                // (at or past cap: the list of this query overflowed, k_select rescans the chunk)
                            const float dist = fmaf(-2.f, acc, sA[2 * row] + bnv);
                            p.cand[(q0 + row) * p.cap + pos] = make_key(dist < 0.f ? 0.f : dist, id);
                        }
                    }
                }
            } else
            for (uint32_t r0 = 0; r0 < (xstage ? 128u : 1u); r0 += XROWS) {           // (without staging: one round over everything)
                if (xstage) {
                    if (r0) __syncthreads();                              // the first half's readers are done
                    if (vec4) {
                        const uint32_t segs = dv / 4;
                        for (uint32_t i = tid; i < XROWS * segs; i += 256) {
                            const uint32_t row = r0 + i / segs, seg = i % segs;
                            *reinterpret_cast<float4 *>(xstage + (row - r0) * XP + seg * 16) =
                                reinterpret_cast<const float4 *>(p.xq + (q0 + (row < q_valid ? row : q_valid - 1)) * (size_t)dv)[seg];
                        }
                    } else {
                        for (uint32_t i = tid; i < XROWS * dv; i += 256) {
                            const uint32_t row = r0 + i / dv, t = i % dv;
                            reinterpret_cast<float *>(xstage + (row - r0) * XP)[t] = p.xq[(q0 + (row < q_valid ? row : q_valid - 1)) * (size_t)dv + t];
                        }
                    }
                }
                __syncthreads();
                for (uint32_t e = tid; e < n; e += 256) {
                    const uint32_t row = pd.loc[e], id = pd.id[e];
                    if (xstage && (row < r0 || row >= r0 + XROWS)) continue;
                    const uint32_t pos = atomicAdd(&pd.rbase[row], 1u);
                    if (pos >= p.cap) continue;                           // the list of this query overflowed: k_select rescans the chunk
                    const float *yr = p.xb + (size_t)id * dv;
                    const float4 *y = reinterpret_cast<const float4 *>(yr);
                    float acc = 0.f;
                    if (!vec4) {                                          // workgroup-uniform: odd row lengths, value by value (same k-ordered chain)
                        const float *x = xstage ? reinterpret_cast<const float *>(xstage + (row - r0) * XP) : p.xq + (q0 + row) * (size_t)dv;
#pragma unroll 16
                        for (uint32_t t = 0; t < dv; ++t) acc = fmaf(x[t], yr[t], acc);
                    } else if (xstage) {
                        const float4 *x = reinterpret_cast<const float4 *>(xstage + (row - r0) * XP);
                        if (dv == (uint32_t)D) {
#pragma unroll PF_APPROX_UNROLL
                            for (int t = 0; t < D / 4; ++t) {
                                const float4 a = x[t], b = y[t];
                                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
                            }
                        } else {
#pragma unroll 8
                            for (uint32_t t = 0; t < dv / 4; ++t) {
                                const float4 a = x[t], b = y[t];
                                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
                            }
                        }
                    } else {
                        const float4 *x = reinterpret_cast<const float4 *>(p.xq + (q0 + row) * (size_t)dv);
                        if (dv == (uint32_t)D) {
#pragma unroll PF_APPROX_UNROLL2
                            for (int t = 0; t < D / 4; ++t) {
                                const float4 a = x[t], b = y[t];
                                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
                            }
                        } else {
#pragma unroll 8
                            for (uint32_t t = 0; t < dv / 4; ++t) {
                                const float4 a = x[t], b = y[t];
                                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
                            }
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

}  // namespace pf
