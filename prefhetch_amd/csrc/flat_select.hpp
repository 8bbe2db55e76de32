// flat_select.hpp -- top-k selection: bootstrap by radix selection, reservoir scan, per-wave merges (k_select, k_merge4)
// (part of the pre-filter translation unit pf_flat.hip: included there, in order; gfx950 only)
#pragma once
#include "flat_common.hpp"

namespace pf {

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

// Bootstrap by sampling (batches, exact operands): four radix passes over all 8192 distances of a row made the first selection of a search 27 us
// for 1024 queries.  Here the row's first THREADS distances are a sample; the sample value of rank r (ranks by counting: every thread compares its
// sample with all of them, broadcast LDS reads) is a threshold that lets about 2.5 k of the n distances through; those few hundred keys are collected
// and the caller cuts them to the k smallest (radix_cut over the survivors only).  Returns false -- nothing collected -- when the threshold let
// fewer than k or more than the reservoir holds through (then the radix selection above runs): the result never depends on the sample.
constexpr uint32_t BOOT_CUT_VPT = 16;                           // keys per lane of the wave that cuts a sampled bootstrap down to k (1024 keys: 32 registers)
template <uint32_t THREADS>
__device__ __forceinline__ bool sample_bootstrap(uint64_t *keys, uint32_t &cnt, uint32_t *hist, uint32_t *ctl, uint32_t k, const float *row, size_t nb_first,
                                                 uint32_t n, int tid) {
    static_assert(THREADS == 256, "the sample lives in the 256-word histogram");
    constexpr int VPT = 8192 / THREADS;
    if (n < 8 * THREADS || 3 * (uint64_t)k * THREADS > (uint64_t)n * (THREADS - 2)) return false;       // workgroup-uniform: too few rows to sample, or k too large a share of them
    uint32_t u[VPT];
#pragma unroll
    for (int e = 0; e < VPT; ++e) {
        const uint32_t col = tid + e * THREADS;
        u[e] = col < n ? __float_as_uint(row[col]) : 0xFFFFFFFFu;    // the filler is above every distance
    }
#if defined(PF_ABL_SEL_STAGE) && PF_ABL_SEL_STAGE == 1   // ablation (timing only, wrong results): the row is loaded, nothing is selected
    { uint32_t x = 0;
#pragma unroll
      for (int e = 0; e < VPT; ++e) x ^= u[e];
      if (x == 0x5EEDFACEu) ctl[5] = x; }
    return true;
#endif
    hist[tid] = u[0];
    __syncthreads();
    // The sample value that lets about 2.5 k of the n distances through: the one with `want` samples below it (want >= 2.5 k THREADS / n).  ONE
    // wave finds it: every lane sorts four samples, then the wave pops its smallest head `want` + 1 times (a DPP minimum and a ballot per pop) --
    // ranking every sample against all the others by broadcast LDS reads kept the CU's LDS pipe busy for 10 us (five workgroups per CU).
    const uint32_t want = (uint32_t)((5ull * k * THREADS + 2ull * n - 1) / (2ull * n)) + 1u;
    if ((uint64_t)want * n > (uint64_t)(64 * BOOT_CUT_VPT * 3 / 4) * THREADS) return false;           // workgroup-uniform: more than the cutting wave holds (large k): the radix selection
    if (tid < 64) {
        const u32x4 sv = *reinterpret_cast<const u32x4 *>(&hist[4 * tid]);
        uint32_t a = sv[0], b = sv[1], c = sv[2], d = sv[3];
        auto cx = [](uint32_t &x, uint32_t &y) { const uint32_t lo = x < y ? x : y, hi = x < y ? y : x; x = lo; y = hi; };
        cx(a, b); cx(c, d); cx(a, c); cx(b, d); cx(b, c);             // a <= b <= c <= d
        uint32_t below = 0, thr_v = 0;
        for (;;) {                                                    // wave-uniform trip count
            uint32_t m = a;
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) { const uint32_t o = __shfl_xor(m, sh); m = o < m ? o : m; }
            const uint64_t eq = __ballot(a == m);
            thr_v = m;
            below += (uint32_t)__popcll(eq);
            if (below > want) break;                                  // m has at least `want` samples below or equal before it: the threshold
            if (a == m) { a = b; b = c; c = d; d = 0xFFFFFFFFu; }     // the lanes that held it move on
        }
        if (tid == 0) ctl[4] = thr_v;
    }
    __syncthreads();
    const uint32_t thr = ctl[4];
    uint32_t take = 0;
#pragma unroll
    for (int e = 0; e < VPT; ++e) take += u[e] <= thr ? 1u : 0u;     // (the filler never passes: thr is a real distance)
    uint32_t base = take ? atomicAdd(&cnt, take) : 0u;
    __syncthreads();
    const uint32_t total = cnt;
    if (total < k || total > 64 * BOOT_CUT_VPT) {                     // workgroup-uniform: the sample misjudged the row (or more than the cutting wave holds)
        __syncthreads();
        if (tid == 0) cnt = 0;
        __syncthreads();
        return false;
    }
#pragma unroll
    for (int e = 0; e < VPT; ++e)
        if (u[e] <= thr) keys[base++] = ((uint64_t)u[e] << 32) | (uint32_t)(nb_first + tid + e * THREADS);
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
    if (tid == 0) { cnt = 0; ctl[3] = prefix; }                                   // (ctl[3]: the k-th distance, for a caller that leaves the keys unsorted)
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
    if (p.mode == 0 && c0 == 0 && p.nb_count > k && p.nb_count <= 8192) {             // first chunk, more rows than results
        // batches on exact operands, more chunks to come: the sampled threshold + a cut of its few hundred survivors (the state may stay unsorted)
        if constexpr (THREADS == 256) {
            if (!p.last && !(p.q_flags && (!p.base_exact || (p.q_flags[q / 128] & 1u)))) {        // workgroup-uniform
                done = sample_bootstrap<THREADS>(keys, cnt, hist, ctl, k, p.slab + q * (size_t)p.slab_ld, p.nb_first, (uint32_t)p.nb_count, tid);
#if defined(PF_ABL_SEL_STAGE) && PF_ABL_SEL_STAGE <= 2   // ablation (timing only, wrong results): no cut, nothing written
                if (done) return;
#endif
                if (done) {
                    // the few hundred keys the threshold let through are cut to the k smallest by ONE wave, in registers (the merge's selection: no
                    // workgroup barrier; the four-pass radix cut of the whole workgroup is sixteen barriers for this handful of keys)
                    if (tid < 64) {
                        const uint32_t n = cnt;
                        uint64_t v[BOOT_CUT_VPT];
#pragma unroll
                        for (uint32_t e = 0; e < BOOT_CUT_VPT; ++e) v[e] = (e * 64 < n && e * 64 + tid < n) ? keys[e * 64 + tid] : KEY_INF;
                        uint32_t unused = 0;
                        wave_keep_k_smallest<BOOT_CUT_VPT>(p, q, n, hist, tid, unused, [&](uint32_t e) { return (uint32_t)(v[e] >> 32); }, [&](uint32_t e) { return (uint32_t)v[e]; });
                    }
                    return;
                }
            }
        }
        if (!done) done = radix_bootstrap<THREADS>(keys, cnt, hist, ctl, k, p.slab + q * (size_t)p.slab_ld, p.nb_first, (uint32_t)p.nb_count, tid);
    }
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

}  // namespace pf
