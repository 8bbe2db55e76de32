// ntt_core.hpp -- register-resident negacyclic NTT / ct x pt core for gfx950 (MI355X).
//
// One workgroup transforms one limb-polynomial of N = 2^LOGN coefficients.  Every thread keeps
// R = 2^LOGR coefficients in VGPRs and runs LOGR radix-2 stages on them without touching memory
// ("pass"); between passes the workgroup re-distributes coefficients through LDS so the next
// LOGR index bits become thread-local ("exchange").  N = 8192: 256 threads x 32 coefficients,
// passes over index bits 12..8 | 7..3 | 2..0, two exchanges per transform, 64 KiB of LDS.
//
// Replaces (behaviour, not code): SEAL util::ntt_negacyclic_harvey / inverse_ntt_negacyclic_harvey /
// dyadic_product_coeffmod and Evaluator::multiply_plain, which the reference links
// (/root/reference/CMakeLists.txt:33-38,66) but does not vendor.  Results are canonical residues,
// so any internal reduction strategy is admissible.
//
// Two arithmetic back-ends share this skeleton:
//   ArithF64 -- q < 2^45: coefficients travel as exact integers in doubles; a modular product is an
//               error-free FMA transformation (6 FP64 ops) and forward butterflies need NO range
//               correction at all.  Measured on MI355X (profiles/r01_ubench_instruction_rates.txt):
//               27 cycles per wave-mulmod against 59 for the 64-bit Shoup form below.
//   ArithU64 -- any prime q < 2^61: Harvey lazy butterflies with Shoup quotients on u64.
//
// The file compiles for the device (hipcc) and, unchanged, for the host: tests/cpp/sim_ntt.cpp runs
// the same functions with one OS thread per lane and a std::barrier for s_barrier, which checks the
// index maps and the floating-point error analysis bit-for-bit on a machine without a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PF_HD __device__ __forceinline__
// keeps hipcc's scheduler from hoisting the next phase's loads into this one (register budget)
#define PF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define PF_HD inline
#define PF_SCHED_FENCE() ((void)0)
#endif

namespace pf {

struct TwF64 { double w, wq; };        // twiddle and fl(w/q)
struct TwU64 { uint64_t w, wq; };      // twiddle and floor(w*2^64/q)  (SEAL MultiplyUIntModOperand)

PF_HD uint64_t d2u(double d) { return __builtin_bit_cast(uint64_t, d); }
PF_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

PF_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIPCC__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// ------------------------------------------------------------------------------------------------
// Geometry
// ------------------------------------------------------------------------------------------------
template <int LOGN_>
struct Geo {
    static constexpr int LOGN = LOGN_;
    static constexpr int N = 1 << LOGN;
    static constexpr int LOGR = LOGN >= 12 ? 5 : 4;
    static constexpr int R = 1 << LOGR;
    static constexpr int T = N / R;                               // threads per workgroup
    static constexpr int P = (LOGN + LOGR - 1) / LOGR;            // passes per transform
    // lowest thread-local index bit of pass p
    static constexpr int a(int p) { return p == P - 1 ? 0 : LOGN - (p + 1) * LOGR; }
    static constexpr int bhi(int p) { return LOGN - 1 - p * LOGR; }   // highest bit transformed by pass p
    static constexpr int blo(int p) { return a(p); }                   // lowest bit transformed by pass p
    // coefficient index of register k of thread tid in the layout of pass p
    static PF_HD int base(int p, int tid) {
        const int aa = a(p);
        const int low = tid & ((1 << aa) - 1), high = tid >> aa;
        return (high << (aa + LOGR)) | low;
    }
    // LDS slot of coefficient i for an exchange between two layouts, p being the one with the lower
    // thread-local field: XOR the index bits above that field into the bank-selecting low bits so
    // that the 32 lanes of a ds_read_b64 group (and the 16 of a ds_write_b64 group) hit different
    // 8-byte slots from either layout (MI355X_MICROARCH.md, LDS table).
    static PF_HD int slot(int p, int i) {
        const int aa = a(p);   // callers pass the later (finer) of the two layouts an exchange joins
        if (aa >= 5) return i;
        return i ^ (((i >> (aa + LOGR)) & ((1 << (5 - aa)) - 1)) << aa);
    }
};

// ------------------------------------------------------------------------------------------------
// Arithmetic back-ends
// ------------------------------------------------------------------------------------------------
// ArithF64.  Invariants (q < 2^45, LOGN <= 15):
//   * every value is an integer-valued double with |v| <= 2^50 at any modular product input, so
//     c = rint(fl(v*wq)) is within 1 of v*w/q and r = v*w - c*q is computed EXACTLY:
//     h = fl(v*w), l = v*w - h (exact, FMA), d = fl(h - c*q) is exact because h - c*q = r - l is an
//     integer below 2^47, r = d + l.  |r| < q (0.75 q at the bounds above).
//   * forward: X' = X + T, Y' = X - T with |T| < q, so |X| <= q*(1 + LOGN) after the last stage;
//     (1+15)*2^45 = 2^49: no intermediate correction is ever needed.
//   * inverse: X' = X + Y doubles per stage; values are re-centred (|v| <= q/2) once per pass, so a
//     5-stage pass peaks at 32 q <= 2^50.
struct ArithF64 {
    using V = double;
    using Tw = TwF64;
    double q, qinv;

    static PF_HD V from_u64(uint64_t x) {            // x < 2^52 : set exponent of 2^52, subtract
        return u2d(x | 0x4330000000000000ull) - 4503599627370496.0;
    }
    static PF_HD uint64_t to_u64(V v) {              // 0 <= v < 2^51, integer valued
        return d2u(v + 4503599627370496.0) & 0x000FFFFFFFFFFFFFull;
    }
    PF_HD V mulmod(V y, Tw t) const {
        const double h = y * t.w;
        const double l = __builtin_fma(y, t.w, -h);
        const double c = __builtin_rint(y * t.wq);
        const double d = __builtin_fma(-c, q, h);
        return d + l;
    }
    PF_HD V mulmod2(V a, V b) const {                // both operands variable (dyadic product)
        const double h = a * b;
        const double l = __builtin_fma(a, b, -h);
        const double c = __builtin_rint(h * qinv);
        const double d = __builtin_fma(-c, q, h);
        return d + l;
    }
    PF_HD V recentre(V v) const { return __builtin_fma(-__builtin_rint(v * qinv), q, v); }
    PF_HD void fwd_bfly(V &x, V &y, Tw t) const { const V m = mulmod(y, t); y = x - m; x = x + m; }
    PF_HD void inv_bfly(V &x, V &y, Tw t) const { const V s = x + y, d = x - y; x = s; y = mulmod(d, t); }
    PF_HD void inv_last(V &x, V &y, Tw tn, Tw t) const { const V s = x + y, d = x - y; x = mulmod(s, tn); y = mulmod(d, t); }
    PF_HD void pass_reduce(V &v) const { v = recentre(v); }
    PF_HD V dyadic(V a, V b) const { return mulmod2(a, b); }
    PF_HD V add(V a, V b) const { return a + b; }
    // canonical residue in [0,q) from any in-range lazy value
    PF_HD V canon(V v) const { V r = recentre(v); return r < 0.0 ? r + q : r; }
    // canonical residue from |v| < q
    PF_HD V canon_small(V v) const { return v < 0.0 ? v + q : v; }
    // canonical residue from -q < v < 2q
    PF_HD V canon_sum(V v) const { V r = v < 0.0 ? v + q : v; return r >= q ? r - q : r; }
    static PF_HD V for_dyadic(V v) { return v; }
};

// ArithU64: SEAL's lazy Harvey butterflies.  forward values live in [0,4q), inverse values in [0,2q).
struct ArithU64 {
    using V = uint64_t;
    using Tw = TwU64;
    uint64_t q, two_q, ratio0, ratio1;               // ratio = floor(2^128/q)

    static PF_HD V from_u64(uint64_t x) { return x; }
    static PF_HD uint64_t to_u64(V v) { return v; }
    PF_HD V mul_lazy(V y, Tw t) const { return y * t.w - mulhi64(y, t.wq) * q; }     // [0,2q)
    PF_HD V guard(V v) const { return v >= two_q ? v - two_q : v; }
    PF_HD void fwd_bfly(V &x, V &y, Tw t) const {
        const V u = guard(x), m = mul_lazy(y, t);
        x = u + m; y = u + two_q - m;
    }
    PF_HD void inv_bfly(V &x, V &y, Tw t) const {
        const V s = guard(x + y), d = x + two_q - y;
        x = s; y = mul_lazy(d, t);
    }
    PF_HD void inv_last(V &x, V &y, Tw tn, Tw t) const {
        const V s = guard(x + y), d = x + two_q - y;
        x = mul_lazy(s, tn); y = mul_lazy(d, t);
    }
    PF_HD void pass_reduce(V &) const {}
    // Barrett 128->64 (SEAL dyadic_product_coeffmod); operands must be canonical
    PF_HD V dyadic(V a, V b) const {
        const uint64_t z0 = a * b, z1 = mulhi64(a, b);
        const uint64_t carry = mulhi64(z0, ratio0);
        const uint64_t t2lo = z0 * ratio1, t2hi = mulhi64(z0, ratio1);
        uint64_t tmp1 = t2lo + carry;
        uint64_t tmp3 = t2hi + (tmp1 < t2lo ? 1 : 0);
        const uint64_t t3lo = z1 * ratio0, t3hi = mulhi64(z1, ratio0);
        const uint64_t s = tmp1 + t3lo;
        const uint64_t carry2 = t3hi + (s < tmp1 ? 1 : 0);
        const uint64_t qhat = z1 * ratio1 + tmp3 + carry2;
        const uint64_t r = z0 - qhat * q;
        return r >= q ? r - q : r;
    }
    PF_HD V add(V a, V b) const { return a + b; }
    PF_HD V canon(V v) const { V r = guard(v); return r >= q ? r - q : r; }          // from [0,4q)
    PF_HD V canon_small(V v) const { return v >= q ? v - q : v; }                    // from [0,2q)
    PF_HD V canon_sum(V v) const { V r = v >= two_q ? v - two_q : v; return r >= q ? r - q : r; }   // from [0,3q)
    PF_HD V for_dyadic(V v) const { return canon(v); }
};

// ------------------------------------------------------------------------------------------------
// Passes
// ------------------------------------------------------------------------------------------------
// Twiddle tables are indexed like SEAL's root_powers: entry m+i (m = 2^s groups, group i) holds
// psi^bitrev(m+i); the inverse table holds the modular inverse of the same entry, except
// inv[1] = psi^-bitrev(1) * N^-1 and inv[0] = N^-1 (both only used by the last inverse layer).

template <class G, class A, int PASS>
PF_HD void fwd_pass(typename A::V (&r)[G::R], const A &ar, const typename A::Tw *__restrict__ tw, int tid) {
    constexpr int aa = G::a(PASS);
    constexpr int kb_hi = G::bhi(PASS) - aa, kb_lo = G::blo(PASS) - aa;
    const int high = PASS == 0 ? 0 : (tid >> aa);
#pragma unroll
    for (int kb = kb_hi; kb >= kb_lo; --kb) {
        const int m = 1 << (G::LOGN - 1 - (kb + aa));
        const typename A::Tw *__restrict__ tp = tw + m + (high << (G::LOGR - 1 - kb));
#pragma unroll
        for (int g = 0; g < (G::R >> (kb + 1)); ++g) {
            const typename A::Tw t = tp[g];
#pragma unroll
            for (int j = 0; j < (1 << kb); ++j) {
                const int k0 = (g << (kb + 1)) | j, k1 = k0 | (1 << kb);
                ar.fwd_bfly(r[k0], r[k1], t);
            }
        }
    }
}

template <class G, class A, int PASS>
PF_HD void inv_pass(typename A::V (&r)[G::R], const A &ar, const typename A::Tw *__restrict__ itw, int tid) {
    constexpr int aa = G::a(PASS);
    constexpr int kb_lo = G::blo(PASS) - aa;
    constexpr int kb_hi = G::bhi(PASS) - aa - (PASS == 0 ? 1 : 0);   // pass 0 ends with the N^-1 layer below
    const int high = PASS == 0 ? 0 : (tid >> aa);
#pragma unroll
    for (int kb = kb_lo; kb <= kb_hi; ++kb) {
        const int m = 1 << (G::LOGN - 1 - (kb + aa));
        const typename A::Tw *__restrict__ tp = itw + m + (high << (G::LOGR - 1 - kb));
#pragma unroll
        for (int g = 0; g < (G::R >> (kb + 1)); ++g) {
            const typename A::Tw t = tp[g];
#pragma unroll
            for (int j = 0; j < (1 << kb); ++j) {
                const int k0 = (g << (kb + 1)) | j, k1 = k0 | (1 << kb);
                ar.inv_bfly(r[k0], r[k1], t);
            }
        }
    }
    if constexpr (PASS == 0) {                               // last layer: fold N^-1 in (SEAL does the same)
        const typename A::Tw tn = itw[0], t = itw[1];
#pragma unroll
        for (int j = 0; j < G::R / 2; ++j) ar.inv_last(r[j], r[j + G::R / 2], tn, t);
    }
}

// ------------------------------------------------------------------------------------------------
// Exchange through LDS.  `Sync` is a callable: s_barrier on the device, std::barrier in the simulator.
// WR is the layout the registers are in, RD the layout they are wanted in.
// ------------------------------------------------------------------------------------------------
template <class G, class V, int WR, int RD, class Sync>
PF_HD void exchange(V (&r)[G::R], V *lds, int tid, Sync &&sync) {
    constexpr int PS = WR > RD ? WR : RD;
    const int bw = G::base(WR, tid), br = G::base(RD, tid);
    sync();                                   // previous readers of the buffer are done
#pragma unroll
    for (int k = 0; k < G::R; ++k) lds[G::slot(PS, bw | (k << G::a(WR)))] = r[k];
    sync();
#pragma unroll
    for (int k = 0; k < G::R; ++k) r[k] = lds[G::slot(PS, br | (k << G::a(RD)))];
}

// ------------------------------------------------------------------------------------------------
// Whole-transform drivers (registers hold layout 0 on entry of fwd, layout 0 on exit of inv)
// ------------------------------------------------------------------------------------------------
// forward passes [0, P-1) with their exchanges: leaves the registers in the layout of the last pass,
// that pass still to run (callers may issue independent loads in between)
template <class G, class A, class Sync>
PF_HD void fwd_head(typename A::V (&r)[G::R], const A &ar, const typename A::Tw *__restrict__ tw,
                    typename A::V *lds, int tid, Sync &&sync) {
    if constexpr (G::P >= 2) { fwd_pass<G, A, 0>(r, ar, tw, tid); exchange<G, typename A::V, 0, 1>(r, lds, tid, sync); }
    if constexpr (G::P >= 3) { fwd_pass<G, A, 1>(r, ar, tw, tid); exchange<G, typename A::V, 1, 2>(r, lds, tid, sync); }
    if constexpr (G::P >= 4) { fwd_pass<G, A, 2>(r, ar, tw, tid); exchange<G, typename A::V, 2, 3>(r, lds, tid, sync); }
}

template <class G, class A, class Sync>
PF_HD void fwd_all(typename A::V (&r)[G::R], const A &ar, const typename A::Tw *__restrict__ tw,
                   typename A::V *lds, int tid, Sync &&sync) {
    fwd_head<G, A>(r, ar, tw, lds, tid, sync);
    fwd_pass<G, A, G::P - 1>(r, ar, tw, tid);
}

template <class G, class A, class Sync>
PF_HD void inv_all(typename A::V (&r)[G::R], const A &ar, const typename A::Tw *__restrict__ itw,
                   typename A::V *lds, int tid, Sync &&sync) {
    if constexpr (G::P >= 4) {
        inv_pass<G, A, 3>(r, ar, itw, tid);
#pragma unroll
        for (int k = 0; k < G::R; ++k) ar.pass_reduce(r[k]);
        exchange<G, typename A::V, 3, 2>(r, lds, tid, sync);
    }
    if constexpr (G::P >= 3) {
        inv_pass<G, A, 2>(r, ar, itw, tid);
#pragma unroll
        for (int k = 0; k < G::R; ++k) ar.pass_reduce(r[k]);
        exchange<G, typename A::V, 2, 1>(r, lds, tid, sync);
    }
    if constexpr (G::P >= 2) {
        inv_pass<G, A, 1>(r, ar, itw, tid);
#pragma unroll
        for (int k = 0; k < G::R; ++k) ar.pass_reduce(r[k]);
        exchange<G, typename A::V, 1, 0>(r, lds, tid, sync);
    }
    inv_pass<G, A, 0>(r, ar, itw, tid);
}

// ------------------------------------------------------------------------------------------------
// Kernels bodies.  Global data is always read/written in layout 0 (lane-contiguous, coalesced).
// ------------------------------------------------------------------------------------------------
enum : int { CTPT_ACCUMULATE = 1, CTPT_IN_NTT = 2, CTPT_OUT_NTT = 4 };

template <class G, class A>
PF_HD void load_l0(typename A::V (&r)[G::R], const uint64_t *src, int tid) {
#pragma unroll
    for (int k = 0; k < G::R; ++k) r[k] = A::from_u64((src + (k << G::a(0)))[tid]);
}

// forward NTT of one limb-polynomial, in place or out of place
template <class G, class A, class Sync>
PF_HD void body_ntt_fwd(const A &ar, const typename A::Tw *__restrict__ tw, const uint64_t *src,
                        uint64_t *dst, typename A::V *lds, int tid, Sync &&sync) {
    typename A::V r[G::R];
    load_l0<G, A>(r, src, tid);
    fwd_all<G, A>(r, ar, tw, lds, tid, sync);
    if constexpr (G::P >= 2) exchange<G, typename A::V, G::P - 1, 0>(r, lds, tid, sync);
#pragma unroll
    for (int k = 0; k < G::R; ++k) (dst + (k << G::a(0)))[tid] = A::to_u64(ar.canon(r[k]));
}

template <class G, class A, class Sync>
PF_HD void body_ntt_inv(const A &ar, const typename A::Tw *__restrict__ itw, const uint64_t *src,
                        uint64_t *dst, typename A::V *lds, int tid, Sync &&sync) {
    typename A::V r[G::R];
    load_l0<G, A>(r, src, tid);
    if constexpr (G::P >= 2) exchange<G, typename A::V, 0, G::P - 1>(r, lds, tid, sync);
    inv_all<G, A>(r, ar, itw, lds, tid, sync);
#pragma unroll
    for (int k = 0; k < G::R; ++k) (dst + (k << G::a(0)))[tid] = A::to_u64(ar.canon_small(r[k]));
}

// ct x pt for one limb-polynomial: [NTT] -> dyadic with pt (NTT form) -> [INTT] -> [+= out]
template <class G, class A, int FLAGS, class Sync>
PF_HD void body_ctpt(const A &ar, const typename A::Tw *__restrict__ tw, const typename A::Tw *__restrict__ itw,
                     const uint64_t *ct, const uint64_t *pt, uint64_t *out,
                     typename A::V *lds, int tid, Sync &&sync) {
    using V = typename A::V;
    constexpr int LAST = G::P - 1;
    V r[G::R];
    load_l0<G, A>(r, ct, tid);
    V pv[G::R];
    if constexpr (FLAGS & CTPT_IN_NTT) {
        load_l0<G, A>(pv, pt, tid);
#pragma unroll
        for (int k = 0; k < G::R; ++k) r[k] = ar.dyadic(r[k], pv[k]);
        if constexpr (FLAGS & CTPT_OUT_NTT) {
#pragma unroll
            for (int k = 0; k < G::R; ++k) {
                V v = ar.canon_small(r[k]);
                if constexpr (FLAGS & CTPT_ACCUMULATE) v = ar.canon_sum(ar.add(v, A::from_u64((out + (k << G::a(0)))[tid])));
                (out + (k << G::a(0)))[tid] = A::to_u64(v);
            }
            return;
        }
        if constexpr (G::P >= 2) exchange<G, V, 0, LAST>(r, lds, tid, sync);
    } else {
        fwd_all<G, A>(r, ar, tw, lds, tid, sync);
        // fetch the plaintext limb (coalesced, layout 0) and move it into the layout the forward
        // transform ended in.  Issued here rather than earlier: together with a pass's hoisted
        // twiddles 64 more live VGPRs spill at the 256-register budget of 2 workgroups per CU.
        PF_SCHED_FENCE();
        load_l0<G, A>(pv, pt, tid);
        if constexpr (G::P >= 2) exchange<G, V, 0, LAST>(pv, lds, tid, sync);
        PF_SCHED_FENCE();
#pragma unroll
        for (int k = 0; k < G::R; ++k) r[k] = ar.dyadic(ar.for_dyadic(r[k]), pv[k]);
        if constexpr (FLAGS & CTPT_OUT_NTT) {
            if constexpr (G::P >= 2) exchange<G, V, LAST, 0>(r, lds, tid, sync);
#pragma unroll
            for (int k = 0; k < G::R; ++k) {
                V v = ar.canon_small(r[k]);
                if constexpr (FLAGS & CTPT_ACCUMULATE) v = ar.canon_sum(ar.add(v, A::from_u64((out + (k << G::a(0)))[tid])));
                (out + (k << G::a(0)))[tid] = A::to_u64(v);
            }
            return;
        }
    }
    inv_all<G, A>(r, ar, itw, lds, tid, sync);
#pragma unroll
    for (int k = 0; k < G::R; ++k) {
        V v = ar.canon_small(r[k]);
        if constexpr (FLAGS & CTPT_ACCUMULATE) v = ar.canon_sum(ar.add(v, A::from_u64((out + (k << G::a(0)))[tid])));
        (out + (k << G::a(0)))[tid] = A::to_u64(v);
    }
}

}  // namespace pf
