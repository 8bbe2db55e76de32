// pf_ntt_kernels.hpp -- gfx950 kernels wrapping the bodies of ntt_core.hpp, and their launch argument
// blocks.  Included by pf_ntt_inst.hip (one translation unit per ring degree, so the degrees build
// in parallel) and by pf_ntt.hip (element-wise kernels, context, C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "ntt_core.hpp"

namespace pf {

// Per-limb constants in HBM; a workgroup reads its limb's entry through the scalar cache.
// Twiddle tables are addressed as 8-byte-word offsets from ONE kernel-argument base pointer, so the
// loads are global_/s_load (a pointer fetched from memory would make them flat_load).
struct LimbDev {
    uint64_t q, two_q, ratio0, ratio1;
    double qd, qinv;
    uint32_t fwd_u, inv_u, fwd_f, inv_f;     // word offsets into NttArgs::tables (u64 tables 16-byte aligned)
    // key switching with the LAST modulus of the context as special prime P
    uint64_t ks_half_mod;                    // floor(P/2) mod q
    uint64_t ks_pinv, ks_pinv_quot;          // P^-1 mod q and its Shoup quotient
};

struct NttArgs {
    const LimbDev *limbs;
    const void *tables;
    const uint64_t *src;
    uint64_t *dst;
    const uint64_t *pt;
    size_t n_pairs;          // ct x pt only: B * L (ciphertext, limb) pairs
    uint32_t L;
    uint32_t pt_broadcast;
    uint32_t ks_decomp;      // key switching: number of digits (data limbs); L above is then the key-modulus count K
    uint32_t ct_fanout;      // ct x pt only: consecutive outputs that share one input ciphertext (0 or 1: none)
    // k_rows_ntt: the polynomial is packed on the fly from rows of a base matrix
    const float *rows_xb; const int64_t *rows_ids; size_t rows_nb; uint32_t rows_d, rows_per_poly;
};

template <class A> struct ArithOf;
template <> struct ArithOf<ArithF64> {
    static __device__ __forceinline__ ArithF64 make(const LimbDev &l) { return ArithF64{l.qd, l.qinv}; }
    static __device__ __forceinline__ const TwF64 *fwd(const void *t, const LimbDev &l) { return reinterpret_cast<const TwF64 *>(static_cast<const uint64_t *>(t) + l.fwd_f); }
    static __device__ __forceinline__ const TwF64 *inv(const void *t, const LimbDev &l) { return reinterpret_cast<const TwF64 *>(static_cast<const uint64_t *>(t) + l.inv_f); }
};
template <bool LAZY> struct ArithOf<ArithU64T<LAZY>> {
    static __device__ __forceinline__ ArithU64T<LAZY> make(const LimbDev &l) { return ArithU64T<LAZY>{l.q, l.two_q, l.ratio0, l.ratio1}; }
    static __device__ __forceinline__ const TwU64 *fwd(const void *t, const LimbDev &l) { return reinterpret_cast<const TwU64 *>(static_cast<const uint64_t *>(t) + l.fwd_u); }
    static __device__ __forceinline__ const TwU64 *inv(const void *t, const LimbDev &l) { return reinterpret_cast<const TwU64 *>(static_cast<const uint64_t *>(t) + l.inv_u); }
};

// waves per SIMD the kernels are register-budgeted for: workgroups per CU (LDS-limited, at most 3) x waves per
// workgroup / 4 SIMDs, at least 2
#define PF_WG_PER_CU(LOGN, A) ((160 * 1024) / (Xchg<Geo<LOGN>, A>::LDS_ENTRIES * 8) > 3 ? 3 : (160 * 1024) / (Xchg<Geo<LOGN>, A>::LDS_ENTRIES * 8))
#define PF_WAVES_PER_SIMD(LOGN, A) ((PF_WG_PER_CU(LOGN, A) * Geo<LOGN>::T / 256) < 2 ? 2 : (PF_WG_PER_CU(LOGN, A) * Geo<LOGN>::T / 256))
// ... for an explicit geometry class
#ifndef PF_SMALL_GEO_WG
#define PF_SMALL_GEO_WG 4
#endif
template <class G, class A> constexpr int wg_cap() { return G::LOGR < default_logr(G::LOGN) ? PF_SMALL_GEO_WG : 3; }
template <class G, class A> constexpr int wg_per_cu() { return (160 * 1024) / (Xchg<G, A>::LDS_ENTRIES * 8) > wg_cap<G, A>() ? wg_cap<G, A>() : (160 * 1024) / (Xchg<G, A>::LDS_ENTRIES * 8); }
template <class G, class A> constexpr int waves_per_simd() { return (wg_per_cu<G, A>() * G::T / 256) < 2 ? 2 : (wg_per_cu<G, A>() * G::T / 256); }

#ifndef PF_KS_PERSIST
#define PF_KS_PERSIST 0        // experiment, see k_ks_ntt: measured 24.1 ms per 256 key switches against 23.5 (the loop costs 124 B of scratch per thread)
#endif

struct WgSync { __device__ __forceinline__ void operator()() const { __syncthreads(); } };

// One workgroup = one limb-polynomial (blockIdx.x).  Consecutive block ids cycle through the limbs, and blocks
// b and b+8 share an XCD under round-robin placement, so for L | 8 an XCD's L2 only ever holds one or two limbs'
// twiddle tables.  (A persistent-workgroup loop was measured: no gain -- dispatch is not the bottleneck -- and
// its loop-invariant LDS/global addresses, hoisted by hipcc, cost ~80 VGPRs.)
template <int LOGN, class A, bool INVERSE>
__global__ void __launch_bounds__(Geo<LOGN>::T, PF_WAVES_PER_SIMD(LOGN, A)) k_ntt(NttArgs p) {
    using G = Geo<LOGN>;
    __shared__ typename A::V lds[Xchg<G, A>::LDS_ENTRIES];
    const size_t poly = blockIdx.x;
    const LimbDev &lm = p.limbs[poly % p.L];
    const A ar = ArithOf<A>::make(lm);
    const uint64_t *src = p.src + poly * G::N;
    uint64_t *dst = p.dst + poly * G::N;
    if constexpr (INVERSE) body_ntt_inv<G, A>(ar, ArithOf<A>::inv(p.tables, lm), src, dst, lds, (int)threadIdx.x, WgSync{});
    else body_ntt_fwd<G, A>(ar, ArithOf<A>::fwd(p.tables, lm), src, dst, lds, (int)threadIdx.x, WgSync{});
}

// Key switching, step 1: block id = (b * D + I) * K + J computes X[b][I][J] = NTT_{m_J}(target[b][I] mod m_J)
// (D digits = data limbs, K = D+1 key moduli).  SEAL: modulo_poly_coeffs + ntt_negacyclic_harvey inside
// Evaluator::switch_key_inplace.
template <int LOGN, class A>
__global__ void __launch_bounds__(Geo<LOGN>::T, PF_WAVES_PER_SIMD(LOGN, A)) k_ks_ntt(NttArgs p) {
    using G = Geo<LOGN>;
    __shared__ typename A::V lds[Xchg<G, A>::LDS_ENTRIES];
    if constexpr (LOGN >= 15 && PF_KS_PERSIST) {
        // One workgroup owns a CU at this degree (128 KiB of LDS), so nothing else covers its load and store phases: the
        // workgroup stays and walks the transforms id, id + gridDim, ... (n_pairs of them) -- the loads of the next one are issued
        // while the stores of the last one drain.  gridDim is a multiple of the modulus count, so a workgroup keeps its limb.
        for (size_t id = blockIdx.x; id < p.n_pairs; id += gridDim.x) {
            int tid = (int)threadIdx.x;
            PF_LAUNDER(tid);                             // per-iteration addresses: hoisted out of the loop they would cost ~80 VGPRs
            const uint32_t J = (uint32_t)(id % p.L);
            const size_t digit = id / p.L;
            const LimbDev &lm = p.limbs[J];
            const A ar = ArithOf<A>::make(lm);
            body_ntt_fwd_mod<G, A>(ar, ArithOf<A>::fwd(p.tables, lm), p.src + digit * G::N, p.dst + id * G::N, lm.q, lm.ratio1, lds, tid,
                                   WgSync{});
        }
        return;
    }
    const size_t id = blockIdx.x;
    const uint32_t J = (uint32_t)(id % p.L);
    const size_t digit = id / p.L;                       // b * D + I
    const LimbDev &lm = p.limbs[J];
    const A ar = ArithOf<A>::make(lm);
    body_ntt_fwd_mod<G, A>(ar, ArithOf<A>::fwd(p.tables, lm), p.src + digit * G::N, p.dst + id * G::N, lm.q, lm.ratio1, lds,
                           (int)threadIdx.x, WgSync{});
}

// The same arithmetic, exchanging through a whole-polynomial LDS buffer (one round and two barriers per exchange instead of two
// and four; twiddles requested ahead of the exchange): for kernels that sit at two workgroups per CU anyway (k_rows_ctpt).
template <class A> struct WholeXchg : A { static constexpr bool HALF_EXCHANGE_OK = false; };
template <class A> struct ArithOf<WholeXchg<A>> {
    static __device__ __forceinline__ WholeXchg<A> make(const LimbDev &l) { return WholeXchg<A>{ArithOf<A>::make(l)}; }
    static __device__ __forceinline__ auto fwd(const void *t, const LimbDev &l) { return ArithOf<A>::fwd(t, l); }
    static __device__ __forceinline__ auto inv(const void *t, const LimbDev &l) { return ArithOf<A>::inv(t, l); }
};

// Fused ct x pt over the ciphertext batch [B][2][L][N].  XCD-aware block order: blocks b and b+8 share an XCD
// under round-robin placement, so XCD x takes the (ciphertext, limb) pairs m = x (mod 8) and runs the two
// polynomials of a pair back to back -- the plaintext limb pt[b][l] they both multiply by is fetched from HBM
// once and re-read from that XCD's L2 (rocprof FETCH_SIZE showed it coming from memory twice under the natural
// order).  For L | 8 an XCD still only ever touches one or two limbs' twiddle tables.
#ifdef PF_CTPT_WHOLE     // experiment (N = 8192): two workgroups per CU with the whole-polynomial exchange; measured 0.47 ms against 0.445 sustained
template <int LOGN, class A0, int FLAGS>
__global__ void __launch_bounds__(Geo<LOGN>::T, 2) k_ctpt(NttArgs p) {
    using A = WholeXchg<A0>;
#else
template <int LOGN, class A, int FLAGS, int LOGR = default_logr(LOGN)>
__global__ void __launch_bounds__((Geo<LOGN, LOGR>::T), (waves_per_simd<Geo<LOGN, LOGR>, A>())) k_ctpt(NttArgs p) {
#endif
    using G = Geo<LOGN, LOGR>;
    __shared__ typename A::V lds[Xchg<G, A>::LDS_ENTRIES];
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const size_t m = (size_t)(j >> 1) * 8 + xcd;              // (ciphertext, limb) pair
    if (m >= p.n_pairs) return;
    const uint32_t limb = (uint32_t)(m % p.L);
    const size_t ctidx = m / p.L;
    const size_t poly = (ctidx * 2 + (j & 1)) * p.L + limb;   // limb-polynomial index in [B][2][L]
    const size_t src_poly = p.ct_fanout > 1 ? ((ctidx / p.ct_fanout) * 2 + (j & 1)) * p.L + limb : poly;
    const LimbDev &lm = p.limbs[limb];
    const A ar = ArithOf<A>::make(lm);
    const uint64_t *pt = p.pt + ((p.pt_broadcast ? 0 : ctidx) * p.L + limb) * G::N;
    body_ctpt<G, A, FLAGS>(ar, ArithOf<A>::fwd(p.tables, lm), ArithOf<A>::inv(p.tables, lm), p.src + src_poly * G::N, pt,
                           p.dst + poly * G::N, lds, (int)threadIdx.x, WgSync{});
}

// Plaintext packing fused into the forward transform (pf_pack_rows_ntt): block b of candidate rows is packed as
// pf_pack_rows does (coefficient d*j - i <- x[ids[b][j]][i], negated on wrap-around) straight into the registers of the
// transform -- consecutive lanes read consecutive floats of one row, backwards -- so the coefficient-form plaintext
// never exists in memory.
// Coefficient c holds x[row jj][i] with jj = ceil(c / d), i = jj*d - c while jj < rows; the top d - 1 coefficients hold
// row 0 negated (X^-i = -X^(N-i)); the rest is empty.  Register k of thread tid is coefficient k*T + tid, so when d divides
// T (d = 128 with T = 256) i does not depend on k and jj advances by T/d per register: one division per thread.
template <class G, class A>
struct RowsLoader {
    const float *xb; const int64_t *ids; size_t nb; uint32_t d, rows; uint64_t q;
    // Register k holds row j[k], element i[k] (negated for the wrap-around of row 0; `live` false = empty coefficient).  The loads go
    // out in two batches -- all row ids, then all values -- from addresses that are always valid (an empty coefficient reads row
    // 0 / id 0 and discards it): written as "if (live) { id = ids[j]; if (in range) v = xb[...]; }" per register, hipcc waited
    // for every one of the 2 x R loads on its own (s_waitcnt vmcnt(0) behind each: 64 serialised round trips per thread).
    __device__ __forceinline__ void fill(typename A::V (&r)[G::R], const bool (&live)[G::R], const bool (&wrap)[G::R], const uint32_t (&j)[G::R],
                                         const uint32_t (&i)[G::R]) const {
        int64_t id[G::R];
#pragma unroll
        for (int k = 0; k < G::R; ++k) id[k] = ids[live[k] ? j[k] : 0u];
        float v[G::R];
#pragma unroll
        for (int k = 0; k < G::R; ++k) {
            const bool ok = live[k] && id[k] >= 0 && (size_t)id[k] < nb;
            v[k] = xb[ok ? (size_t)id[k] * d + i[k] : 0];
            if (!ok) v[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < G::R; ++k) {
            float w = rintf(v[k]);
            if (wrap[k]) w = -w;
            const int64_t iv = (int64_t)w;
            r[k] = A::from_u64(iv >= 0 ? (uint64_t)iv : q - (uint64_t)(-iv));
        }
    }
    __device__ __forceinline__ void operator()(typename A::V (&r)[G::R], int tid) const {
        bool live[G::R], wrap[G::R];
        uint32_t j[G::R], i[G::R];
        if (nb == 0 || rows == 0) {                              // nothing to read from (workgroup-uniform)
#pragma unroll
            for (int k = 0; k < G::R; ++k) r[k] = A::from_u64(0);
            return;
        }
        if (G::T % d == 0) {                                     // workgroup-uniform; d then divides N as well
            // jj = ceil(c / d) advances by T/d per register and i = jj*d - c does not depend on the register: one division
            // per thread.  jj == N/d is row 0 negated (its i equals N - c), jj in [rows, N/d) is empty.
            const uint32_t jj0 = ((uint32_t)tid + d - 1) / d, i0 = jj0 * d - (uint32_t)tid, step = G::T / d;
#pragma unroll
            for (int k = 0; k < G::R; ++k) {
                const uint32_t jj = jj0 + (uint32_t)(G::koff(0, k) / G::T) * step;
                wrap[k] = jj == G::N / d;
                live[k] = wrap[k] || jj < rows;
                j[k] = wrap[k] ? 0u : jj;
                i[k] = i0;
            }
        } else {
            // any d (the criterion of k_pack_rows): row jj = ceil(c / d) while jj < rows, else the wrap-around of row 0
            // when N - c < d, else empty
#pragma unroll
            for (int k = 0; k < G::R; ++k) {
                const uint32_t c = (uint32_t)(G::koff(0, k) + tid), jj = (c + d - 1) / d;
                const bool row = jj < rows;
                wrap[k] = !row && (uint32_t)G::N - c < d;
                live[k] = row || wrap[k];
                j[k] = row ? jj : 0u;
                i[k] = row ? jj * d - c : (uint32_t)G::N - c;
            }
        }
        fill(r, live, wrap, j, i);
    }
};

template <int LOGN, class A>
__global__ void __launch_bounds__(Geo<LOGN>::T, PF_WAVES_PER_SIMD(LOGN, A)) k_rows_ntt(NttArgs p) {
    using G = Geo<LOGN>;
    __shared__ typename A::V lds[Xchg<G, A>::LDS_ENTRIES];
    const size_t poly = blockIdx.x;                              // plaintext block * L + limb
    const uint32_t limb = (uint32_t)(poly % p.L);
    const size_t block = poly / p.L;
    const LimbDev &lm = p.limbs[limb];
    const A ar = ArithOf<A>::make(lm);
    const RowsLoader<G, A> load{p.rows_xb, p.rows_ids + block * p.rows_per_poly, p.rows_nb, p.rows_d, p.rows_per_poly, lm.q};
    body_ntt_fwd_from<G, A>(ar, ArithOf<A>::fwd(p.tables, lm), load, p.dst + poly * G::N, lds, (int)threadIdx.x, WgSync{});
}

// Element-wise kernels: 24 B (dyadic/add/sub) or 16 B (negate) of HBM traffic per coefficient, HBM-bound.
// A block covers CHUNK consecutive coefficients of one limb-polynomial, 16 B per lane per access.
enum EwOp : int { EW_MUL = 0, EW_ADD = 1, EW_SUB = 2, EW_NEG = 3 };

struct EwArgs {
    const LimbDev *limbs;
    const uint64_t *a, *b;
    uint64_t *out;
    uint32_t L, logn;
};

template <int OP>
__device__ __forceinline__ uint64_t ew_apply(const ArithU64 &ar, uint64_t x, uint64_t y) {
    if constexpr (OP == EW_MUL) return ar.dyadic(x, y);
    else if constexpr (OP == EW_ADD) { const uint64_t s = x + y; return s >= ar.q ? s - ar.q : s; }
    else if constexpr (OP == EW_SUB) { return x >= y ? x - y : x + ar.q - y; }
    else return x ? ar.q - x : 0;
}

template <int OP>
__global__ void __launch_bounds__(256) k_elementwise(EwArgs p) {
    // chunk = 2048 coefficients (or N when N = 1024): 256 lanes x 16 B x 4 accesses
    const uint32_t chunk_log = p.logn < 11 ? p.logn : 11;
    const size_t first = (size_t)blockIdx.x << chunk_log;
    const size_t poly = first >> p.logn;
    const LimbDev &lm = p.limbs[poly % p.L];
    const ArithU64 ar{lm.q, lm.two_q, lm.ratio0, lm.ratio1};
    const uint32_t per_thread = (1u << chunk_log) / 512;      // 16-byte pairs per lane
    const ulonglong2 *a2 = reinterpret_cast<const ulonglong2 *>(p.a + first);
    const ulonglong2 *b2 = reinterpret_cast<const ulonglong2 *>(p.b + first);
    ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(p.out + first);
    for (uint32_t j = 0; j < per_thread; ++j) {
        const uint32_t i = j * 256 + threadIdx.x;
        const ulonglong2 x = a2[i];
        ulonglong2 y = x;
        if constexpr (OP != EW_NEG) y = b2[i];
        ulonglong2 r;
        r.x = ew_apply<OP>(ar, x.x, y.x);
        r.y = ew_apply<OP>(ar, x.y, y.y);
        o2[i] = r;
    }
}

}  // namespace pf


namespace pf {
// Defined in pf_ntt_inst.hip, one per ring degree.  arith: 0 = ArithF64, 1 = ArithU64, 2 = ArithU64L;
// op: 0 forward NTT, 1 inverse NTT, 2 fused ct x pt with `flags`, 3 key-switch digit NTT, 4 forward NTT of plaintexts packed on the fly from base rows.
#define PF_DECL_LAUNCH(LN) void launch_logn_##LN(int arith, int op, int flags, const NttArgs &a, unsigned grid, hipStream_t s);
PF_DECL_LAUNCH(10) PF_DECL_LAUNCH(11) PF_DECL_LAUNCH(12) PF_DECL_LAUNCH(13) PF_DECL_LAUNCH(14) PF_DECL_LAUNCH(15)
#undef PF_DECL_LAUNCH
// pf_ct_rows_mul: out[b] = ct[b / fanout] x pack(ids[b]) with the plaintext packed, transformed and multiplied inside one
// workgroup per (product, limb); ct in NTT form, out in coefficient form.  Block order: XCD x (blocks x, x + 8, ...) walks
// the (ciphertext, limb) groups g = x (mod 8) and runs a group's `fanout` products back to back, so the ciphertext limb
// they share comes from HBM once and from that XCD's L2 after.
// The 64 registers of the kept plaintext put the kernel at two workgroups per CU whatever it does, so up to N = 8192 it takes
// the 80 KiB of LDS that leaves it and exchanges through a whole-polynomial buffer (WholeXchg).
template <int LOGN, class A> struct RowsCtptArith { using type = A; };
template <class A> struct RowsCtptArith<13, A> { using type = WholeXchg<A>; };
template <class A> struct RowsCtptArith<12, A> { using type = WholeXchg<A>; };

template <int LOGN, class A0>
__global__ void __launch_bounds__(Geo<LOGN>::T, 2) k_rows_ctpt(NttArgs p) {
    using G = Geo<LOGN>;
    using A = typename RowsCtptArith<LOGN, A0>::type;
    __shared__ typename A::V lds[Xchg<G, A>::LDS_ENTRIES];
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3, fan = p.ct_fanout;
    const size_t g = (size_t)(j / fan) * 8 + xcd;             // (input ciphertext, limb)
    const size_t ctidx = g / p.L, prod = ctidx * fan + j % fan;
    if (prod >= p.n_pairs) return;                            // n_pairs: number of products here
    const uint32_t limb = (uint32_t)(g % p.L);
    const LimbDev &lm = p.limbs[limb];
    const A ar = ArithOf<A>::make(lm);
    const RowsLoader<G, A> load{p.rows_xb, p.rows_ids + prod * p.rows_per_poly, p.rows_nb, p.rows_d, p.rows_per_poly, lm.q};
    const uint64_t *ct = p.src + (ctidx * 2 * p.L + limb) * G::N;
    uint64_t *out = p.dst + (prod * 2 * p.L + limb) * G::N;
    body_rows_ctpt<G, A>(ar, ArithOf<A>::fwd(p.tables, lm), ArithOf<A>::inv(p.tables, lm), load, ct, ct + (size_t)p.L * G::N, out,
                         out + (size_t)p.L * G::N, lds, (int)threadIdx.x, WgSync{});
}

}  // namespace pf
