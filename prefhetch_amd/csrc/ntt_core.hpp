// ntt_core.hpp -- register-resident negacyclic NTT / ct x pt core for gfx950 (MI355X).
//
// One workgroup transforms one limb-polynomial of N = 2^LOGN coefficients.  Every thread keeps
// R = 2^LOGR coefficients in VGPRs and runs radix-2 stages on them without touching memory ("pass");
// between passes the workgroup re-distributes coefficients through LDS so the next index bits become
// thread-local ("exchange").  N = 8192: 256 threads x 32 coefficients, passes over index bits
// 12..8 | 7..3 | 2..0, two exchanges per transform, 64 KiB of LDS.
//
// Layouts.  Pass p < P-1 owns the contiguous index bits [a, a+LOGR); the thread id fills the rest.
// The LAST pass owns the low `nl` bits it still has to transform plus the TOP LOGR-nl bits (carried):
// a lane then holds runs of 2^nl consecutive coefficients and consecutive lanes hold consecutive
// runs, so (1) its twiddles, (2) the NTT-form plaintext of the fused kernel and (3) NTT-form input /
// output of the stand-alone transforms are all read and written lane-contiguously from HBM, with no
// extra pass through LDS.  Coefficient-form data is read/written in layout 0 (lane-contiguous too).
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
#include <type_traits>
#include "lds_swizzle_tab.hpp"

#if defined(__HIPCC__)
#define PF_HD __device__ __forceinline__
// keeps hipcc's scheduler from re-serialising the batched arithmetic below
#define PF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// distinct asm statements (compiler memory barriers) at both ends of two sibling branches keep LLVM from
// sinking/hoisting their common LDS accesses into one block that picks the REGISTER by a pointer phi -- which
// would demote the register-file array to scratch memory
#define PF_BRANCH_TAG(s) asm volatile("; " s ::: "memory")
// makes a value opaque to CSE: what is derived from it afterwards is recomputed instead of being kept alive
// (and spilled) from its first use
#define PF_LAUNDER(v) asm volatile("" : "+v"(v))
#else
#define PF_LAUNDER(v) ((void)0)
#define PF_BRANCH_TAG(s) ((void)0)
#define PF_HD inline
#define PF_SCHED_FENCE() ((void)0)
#endif

// phase stamps of the diagnostic build (tools/ntt_phase_stamps.hip); nothing in the product
#ifndef PF_STAMP
#define PF_STAMP(id) ((void)0)
#endif
#ifndef PF_STAMP_X
#define PF_STAMP_X(step) ((void)0)       // inside an exchange round: 0 entered, 1 first barrier passed, 2 written, 3 second barrier passed, 4 read
#endif

namespace pf {

struct TwU64 { uint64_t w, wq; };      // twiddle and floor(w*2^64/q)  (SEAL MultiplyUIntModOperand)
struct TwF64 { double w; };            // table entry of the FP64 back-end: the twiddle as a double
struct TwF64R { double w, wq; };       // ... resolved on load: wq = fl(w * fl(1/q))
struct alignas(16) U64x2 { uint64_t x, y; };

// Load through the CONSTANT address space: the twiddle tables are never written while a kernel runs, and a
// uniform constant-space load becomes an s_load even after a barrier (a global-space load behind the barrier's
// fence is no longer provably unclobbered, so hipcc would issue it once per lane into VGPRs).
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
PF_HD T const_load(const T *p) {               // T = double or uint64_t
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}
#else
template <class T>
PF_HD T const_load(const T *p) { return *p; }
#endif
PF_HD TwF64 const_load_tw(const TwF64 *p) { return TwF64{const_load(&p->w)}; }
PF_HD TwU64 const_load_tw(const TwU64 *p) { return TwU64{const_load(&p->w), const_load(&p->wq)}; }

PF_HD uint64_t d2u(double d) { return __builtin_bit_cast(uint64_t, d); }
PF_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

PF_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIPCC__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// floor(a*b / 2^64) - e with e in {0, 1, 2}: the three partial products that reach the high word, without the
// low x low product and without the carries of the cross terms (three 32-bit multiplies instead of four plus carries)
PF_HD uint64_t mulhi64_under(uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    return (uint64_t)a1 * b1 + (uint32_t)(((uint64_t)a1 * b0) >> 32) + (uint32_t)(((uint64_t)a0 * b1) >> 32);
}

// y*w - c*q (mod 2^64), the tail of a Shoup product, as multiply-add chains.  hipcc builds the two 64-bit products
// separately (2 x {v_mad_u64_u32, 2 v_mul_lo_u32, v_add3_u32}) and subtracts with a carry chain (2 more, plus wait states
// on the carry): 10 instructions and a nop.  Here the subtraction is an addition of c*(2^64 - q) and every partial
// product is accumulated by the multiplier itself:
//     t = y0*w1 + y1*w0 + c0*n1 + c1*n0      4 v_mad_u64_u32; only the low word of t matters
//     u = y0*w0 + c0*n0                       2 v_mad_u64_u32
//     result = u + (t << 32)                  1 v_add_u32 on the high word
// 7 instructions, no carry chain.  NB products go into ONE asm statement, their chains interleaved (hipcc pads every asm
// statement with a wait state and cannot see inside to schedule).  No instruction here reads a scalar register written by
// a vector instruction (the carry-out pair `cy` is write-only), so the block needs no wait states of its own.
// UTW: the twiddles are workgroup-uniform and sit in scalar registers (one scalar operand per instruction either way);
// nq = 2^64 - q is uniform.
#if defined(__HIP_DEVICE_COMPILE__)
#define PF_MS_T0(i) "v_mad_u64_u32 %[t" #i "], %[cy], %[y" #i "0], %[w" #i "1], 0\n\t"
#define PF_MS_T1(i) "v_mad_u64_u32 %[t" #i "], %[cy], %[y" #i "1], %[w" #i "0], %[t" #i "]\n\t"
#define PF_MS_T2(i) "v_mad_u64_u32 %[t" #i "], %[cy], %[c" #i "0], %[n1], %[t" #i "]\n\t"
#define PF_MS_T3(i) "v_mad_u64_u32 %[t" #i "], %[cy], %[c" #i "1], %[n0], %[t" #i "]\n\t"
#define PF_MS_U0(i) "v_mad_u64_u32 %[u" #i "], %[cy], %[y" #i "0], %[w" #i "0], 0\n\t"
#define PF_MS_U1(i) "v_mad_u64_u32 %[u" #i "], %[cy], %[c" #i "0], %[n0], %[u" #i "]\n\t"
#define PF_MS_ALL(OP) OP(0) OP(1) OP(2) OP(3)
// order: both halves of y are dead once T1 has issued, so u is written OVER y (the tied operand u_i is the pair whose halves
// the 32-bit operands y_i0 / y_i1 name: same registers, and U0 reads y_i0 in the instruction that overwrites it)
#define PF_MS_BODY PF_MS_ALL(PF_MS_T0) PF_MS_ALL(PF_MS_T1) PF_MS_ALL(PF_MS_U0) PF_MS_ALL(PF_MS_T2) PF_MS_ALL(PF_MS_U1) PF_MS_ALL(PF_MS_T3)
#define PF_MS_OUT(i) [t##i] "=&v"(t[i]), [u##i] "+v"(y[i])
#define PF_MS_IN(i, WC) [y##i##0] "v"((uint32_t)y[i]), [y##i##1] "v"((uint32_t)(y[i] >> 32)), [w##i##0] WC((uint32_t)w[i]), \
                        [w##i##1] WC((uint32_t)(w[i] >> 32)), [c##i##0] "v"((uint32_t)c[i]), [c##i##1] "v"((uint32_t)(c[i] >> 32))
template <bool UTW>
PF_HD void mulsub4_lo64(uint64_t (&y)[4], const uint64_t (&w)[4], const uint64_t (&c)[4], uint64_t nq) {
    uint64_t t[4], cy;
    const uint32_t n0 = (uint32_t)nq, n1 = (uint32_t)(nq >> 32);
    if constexpr (UTW)
        asm(PF_MS_BODY : PF_MS_OUT(0), PF_MS_OUT(1), PF_MS_OUT(2), PF_MS_OUT(3), [cy] "=&s"(cy)
            : PF_MS_IN(0, "s"), PF_MS_IN(1, "s"), PF_MS_IN(2, "s"), PF_MS_IN(3, "s"), [n0] "s"(n0), [n1] "s"(n1));
    else
        asm(PF_MS_BODY : PF_MS_OUT(0), PF_MS_OUT(1), PF_MS_OUT(2), PF_MS_OUT(3), [cy] "=&s"(cy)
            : PF_MS_IN(0, "v"), PF_MS_IN(1, "v"), PF_MS_IN(2, "v"), PF_MS_IN(3, "v"), [n0] "s"(n0), [n1] "s"(n1));
    // y now holds u.  The sum stays a 32-bit add on the high word (left to itself hipcc builds the pair {0, t} with two
    // moves and adds 64 bits) ...
    uint32_t h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = (uint32_t)(y[i] >> 32) + (uint32_t)t[i];
    asm("" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]));
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = ((uint64_t)h[i] << 32) | (uint32_t)y[i];
    // ... and the pair stays a pair (or hipcc splits every later 64-bit add of it into two adds of zero-padded halves)
    asm("" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
}

// d[i] = K - m[i] (64 bits, K uniform) for four values at once.  hipcc subtracts through VCC (v_sub_co_u32 / v_subb_co_u32
// back to back per value) and has to pad every pair with a wait state: a vector instruction may read a carry two states
// after it was written.  Here the four low halves go first, each with its own carry pair, then the four high halves: three
// instructions between every write and its read, no padding.  (The high word of K sits in a vector register: the carry
// pair is the one scalar operand a VOP3 instruction may read.)
#define PF_SB_LO(i) "v_sub_co_u32 %[l" #i "], %[c" #i "], %[klo], %[m" #i "l]\n\t"
#define PF_SB_HI(i) "v_subb_co_u32 %[h" #i "], %[c" #i "], %[khi], %[m" #i "h], %[c" #i "]\n\t"
#define PF_SB_OUT(i) [l##i] "=&v"(lo[i]), [h##i] "=&v"(hi[i]), [c##i] "=&s"(cy[i])
#define PF_SB_IN(i) [m##i##l] "v"((uint32_t)m[i]), [m##i##h] "v"((uint32_t)(m[i] >> 32))
PF_HD void sub_from_const4(uint64_t K, const uint64_t (&m)[4], uint64_t (&d)[4]) {
    uint32_t lo[4], hi[4];
    uint64_t cy[4];
    asm(PF_MS_ALL(PF_SB_LO) PF_MS_ALL(PF_SB_HI)
        : PF_SB_OUT(0), PF_SB_OUT(1), PF_SB_OUT(2), PF_SB_OUT(3)
        : PF_SB_IN(0), PF_SB_IN(1), PF_SB_IN(2), PF_SB_IN(3), [klo] "s"((uint32_t)K), [khi] "v"((uint32_t)(K >> 32)));
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = ((uint64_t)hi[i] << 32) | lo[i];
    asm("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));          // keeps the pairs whole (see mulsub4_lo64)
}
#else
template <bool UTW>
PF_HD void mulsub4_lo64(uint64_t (&y)[4], const uint64_t (&w)[4], const uint64_t (&c)[4], uint64_t nq) {
    for (int i = 0; i < 4; ++i) y[i] = y[i] * w[i] + c[i] * nq;
}
PF_HD void sub_from_const4(uint64_t K, const uint64_t (&m)[4], uint64_t (&d)[4]) {
    for (int i = 0; i < 4; ++i) d[i] = K - m[i];
}
#endif

// ------------------------------------------------------------------------------------------------
// Geometry
// ------------------------------------------------------------------------------------------------
#ifndef PF_LOGR_LARGE
#define PF_LOGR_LARGE 5        // log2(coefficients per thread) for N >= 4096
#endif
#ifndef PF_LOGR_15
#define PF_LOGR_15 6           // N = 32768: see the note inside Geo
#endif
constexpr int default_logr(int logn) { return logn >= 15 ? PF_LOGR_15 : (logn >= 12 ? PF_LOGR_LARGE : 4); }
// LOGR_: log2(coefficients per thread).  The default is the geometry every kernel is built with; k_ctpt is also instantiated with 16 coefficients
// per thread at N = 4096 / 8192 (pf_ntt_inst.hip) for launches that fill the device less than twice: twice the waves per polynomial.
template <int LOGN_, int LOGR_ = default_logr(LOGN_)>
struct Geo {
    static constexpr int LOGN = LOGN_;
    static constexpr int N = 1 << LOGN;
    // N = 32768: 64 coefficients per thread, 512 threads -- 2 waves per SIMD leave the 64-bit butterflies 256 VGPRs.
    // PF_LOGR_15 = 5 selects 1024 threads x 32 instead (three passes of five stages, four waves per SIMD inside the one
    // workgroup a CU holds, 128 registers): built, verified on the host simulator and on the GPU, and measured in round 2 --
    // forward 1.90 vs 1.95 ms, inverse 2.30 vs 1.99 ms, ct x pt 3.88 vs 3.67 ms, key switch 25.9 vs 26.6 ms per 256: no win.
    // The lone workgroup's phases (global load, passes, exchanges, store) do not overlap whatever its wave count, and the
    // last-pass layout then reads 256-byte runs per lane.  Kept selectable; the default stays 64.
    static constexpr int LOGR = LOGR_;
    static constexpr int R = 1 << LOGR;
    static constexpr int T = N / R;                               // threads per workgroup
    static constexpr int P = (LOGN + LOGR - 1) / LOGR;            // passes per transform
    static constexpr int LAST = P - 1;
    // pass p transforms the nl(p) index bits [a(p), a(p)+nl(p)); the last pass also carries the top nh bits
    static constexpr int nl(int p) { return p == P - 1 ? LOGN - (P - 1) * LOGR : LOGR; }
    static constexpr int nh(int p) { return LOGR - nl(p); }
    static constexpr int a(int p) { return p == P - 1 ? 0 : LOGN - (p + 1) * LOGR; }
    // coefficient index = base(p, tid) | koff(p, k)
    static constexpr int koff(int p, int k) {
        return ((k >> nl(p)) << (LOGN - nh(p))) | ((k & ((1 << nl(p)) - 1)) << a(p));
    }
    // Thread-id placement.  Passes before the last: the low a(p) id bits sit below the thread-local field, the
    // rest above it.  Last pass: the id fills the bits between the low run and the carried top bits in order,
    // except that its MOST SIGNIFICANT bit sits at YBIT, the index bit that is the register MSB of the pass
    // before -- so that across every exchange "top register bit" and "top thread-id bit" swap roles (what the
    // two-round half-buffer exchange below relies on) while lanes 0..2^y-1 still cover one contiguous run.
    static constexpr int TB = LOGN - LOGR;                        // thread-id bits
    static constexpr int YBIT = P >= 2 ? a(P - 2 < 0 ? 0 : P - 2) + LOGR - 1 : 0;
    // index bit of the top REGISTER bit (k >= R/2) of layout p: the highest carried bit, or the highest transformed one
    static constexpr int topreg(int p) { return (p == P - 1 && nh(p) > 0) ? LOGN - 1 : a(p) + nl(p) - 1; }
    // thread-id bit that carries index bit b in layout p (the inverse of base(); b must be a thread bit there)
    static constexpr int tidbit_of(int p, int b) {
        if (NH0 && p == 1) return b == LOGN - 1 ? TB - 1 : b == a(1) - 1 ? TB - 2 : b < a(1) - 1 ? b : b - (a(1) + LOGR) + (a(1) - 1);
        if (NH0 && p == 2) return b == LOGN - 1 ? TB - 1 : b == YBIT ? TB - 2 : b < YBIT ? b - nl(2) : b - nl(2) - 1;
        if (p == P - 1 && P >= 2) {
            const int NL = nl(P - 1), y = YBIT - NL, f = b - NL;
            return f == y ? TB - 1 : (f < y ? f : f - 1);
        }
        return b < a(p) ? b : b - nl(p);
    }
    // NH0: three passes of LOGR stages each (N = 32768 at 32 coefficients per thread): the last pass carries no top bits, so
    // the role-swapping bit pairs of the two exchanges are (LOGN-1, a(0)-1) and (a(0)-1, a(1)-1).  Every thread must take the
    // same half as a writer and as a reader of an exchange (it overwrites the registers it has just written out), i.e. on
    // both sides of an exchange the selector has to be the SAME wave-level thread-id bit: bit TB-1 carries index bit a(0)-1
    // in layout 0 and LOGN-1 in layout 1 (first exchange), bit TB-2 carries a(1)-1 in layout 1 and a(0)-1 in layout 2.
    static constexpr bool NH0 = P == 3 && nh(P - 1) == 0 && LOGN - LOGR >= 8;
    static PF_HD int base(int p, int tid) {
        if constexpr (NH0) {
            if (p == 1) {
                constexpr int A1 = a(1);                             // thread bits: [0, A1-1) low, then the bits above pass 1's field
                const int low = tid & ((1 << (A1 - 1)) - 1), mid = (tid >> (A1 - 1)) & ((1 << (TB - 2 - (A1 - 1))) - 1);
                return low | (mid << (A1 + LOGR)) | (((tid >> (TB - 2)) & 1) << (A1 - 1)) | ((tid >> (TB - 1)) << (LOGN - 1));
            }
            if (p == 2) {
                constexpr int NL = nl(2), y = YBIT - NL;             // lanes count runs in order: thread bits [0, y) sit right above the run
                const int lo = tid & ((1 << (TB - 2)) - 1);
                return ((lo & ((1 << y) - 1)) << NL) | (((tid >> (TB - 2)) & 1) << YBIT) | ((lo >> y) << (YBIT + 1)) | ((tid >> (TB - 1)) << (LOGN - 1));
            }
        }
        if (p == P - 1 && P >= 2) {
            constexpr int NL = nl(P - 1), y = YBIT - NL;
            const int msb = tid >> (TB - 1), lo = tid & ((1 << (TB - 1)) - 1);
            const int field = ((lo >> y) << (y + 1)) | (msb << y) | (lo & ((1 << y) - 1));
            return field << NL;
        }
        const int aa = a(p);
        const int low = tid & ((1 << aa) - 1), high = tid >> aa;
        return (high << (aa + nl(p))) | low;
    }
    // LDS slot of coefficient i in the exchange between passes `pair` and `pair`+1 (GF(2)-linear, see table)
    template <int PAIR>
    static PF_HD int slot(int i) {
        constexpr SwzEntry e = swz_lookup(LOGN, LOGR, PAIR);
        int s = i;
        if constexpr (e.t[0].w != 0) s ^= ((i >> e.t[0].s) & ((1 << e.t[0].w) - 1)) << e.t[0].d;
        if constexpr (e.t[1].w != 0) s ^= ((i >> e.t[1].s) & ((1 << e.t[1].w) - 1)) << e.t[1].d;
        if constexpr (e.t[2].w != 0) s ^= ((i >> e.t[2].s) & ((1 << e.t[2].w) - 1)) << e.t[2].d;
        return s;
    }
};

// ------------------------------------------------------------------------------------------------
// Arithmetic back-ends
// ------------------------------------------------------------------------------------------------
// ArithF64.  Invariants (q < 2^45, LOGN <= 15):
//   * every value is an integer-valued double with |v| <= 2^50 at any modular product input.  With
//     wq = fl(w*fl(1/q)) the estimate fl(v*wq) is within 3*2^-53*|v*w/q| <= 3/8 of v*w/q, so
//     c = rint(.) is within 1 of it and r = v*w - c*q is computed EXACTLY: h = fl(v*w),
//     l = v*w - h (exact, FMA), d = fl(h - c*q) is exact because h - c*q = r - l is an integer below
//     2^47, r = d + l.  |r| < 0.875 q.
//   * forward: X' = X + T, Y' = X - T with |T| < q, so |X| <= q*(1 + LOGN) after the last stage;
//     (1+15)*2^45 = 2^49: no intermediate correction is ever needed.
//   * inverse: X' = X + Y doubles per stage; values are re-centred (|v| <= q/2) once per pass, so a
//     5-stage pass peaks at 32 q <= 2^50.
// The batched members work phase by phase over NB independent operands with scheduling fences in
// between: hipcc otherwise emits each product as one serial dependent chain.
struct ArithF64 {
    using V = double;
    using Tw = TwF64;
    using TwR = TwF64R;
    static constexpr bool PREFETCH_TW = true;      // 8-byte twiddles: a whole pass's worth fits in registers across an exchange
    static constexpr bool HALF_EXCHANGE_OK = true; // fits the 168-VGPR budget of three workgroups per CU
    double q, qinv;

    static PF_HD V from_u64(uint64_t x) {            // x < 2^52 : set exponent of 2^52, subtract
        return u2d(x | 0x4330000000000000ull) - 4503599627370496.0;
    }
    static PF_HD uint64_t to_u64(V v) {              // 0 <= v < 2^51, integer valued
        return d2u(v + 4503599627370496.0) & 0x000FFFFFFFFFFFFFull;
    }
    static constexpr bool WQ0_TABLE = true;        // table entries N .. N+R-1 hold fl(w/q) of entries 0 .. R-1
    PF_HD TwR resolve(Tw t) const { return TwR{t.w, t.w * qinv}; }
    static PF_HD TwR with_quotient(Tw t, double wq) { return TwR{t.w, wq}; }
    template <int NB, bool UTW = false>
    PF_HD void mul_tw_n(V (&y)[NB], const TwR (&t)[NB]) const {     // y[i] <- y[i]*t[i] (mod q), |.| < q
        double h[NB], l[NB], c[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) h[i] = y[i] * t[i].w;
#pragma unroll
        for (int i = 0; i < NB; ++i) c[i] = y[i] * t[i].wq;
        PF_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < NB; ++i) l[i] = __builtin_fma(y[i], t[i].w, -h[i]);
#pragma unroll
        for (int i = 0; i < NB; ++i) c[i] = __builtin_rint(c[i]);
        PF_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < NB; ++i) h[i] = __builtin_fma(-c[i], q, h[i]);
        PF_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < NB; ++i) y[i] = h[i] + l[i];
    }
    template <int NB>
    PF_HD void dyadic_n(V (&a)[NB], const V (&b)[NB]) const {       // a[i] <- a[i]*b[i] (mod q), both variable
        double h[NB], l[NB], c[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) h[i] = a[i] * b[i];
        PF_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < NB; ++i) l[i] = __builtin_fma(a[i], b[i], -h[i]);
#pragma unroll
        for (int i = 0; i < NB; ++i) c[i] = h[i] * qinv;
        PF_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < NB; ++i) c[i] = __builtin_rint(c[i]);
        PF_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < NB; ++i) h[i] = __builtin_fma(-c[i], q, h[i]);
        PF_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < NB; ++i) a[i] = h[i] + l[i];
    }
    template <int NB>
    PF_HD void recentre_n(V (&v)[NB]) const {                       // v <- v - q*rint(v/q)
        double c[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) c[i] = v[i] * qinv;
        PF_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < NB; ++i) c[i] = __builtin_rint(c[i]);
        PF_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < NB; ++i) v[i] = __builtin_fma(-c[i], q, v[i]);
    }
    PF_HD void fwd_combine(V &x, V &y, V m) const { y = x - m; x = x + m; }
    template <int J> PF_HD void inv_split(V &x, V y, V &d) const { d = x - y; x = x + y; }
    template <int NB> PF_HD void fwd_combine_n(V (&x)[NB], V (&y)[NB], const V (&m)[NB]) const {
#pragma unroll
        for (int i = 0; i < NB; ++i) fwd_combine(x[i], y[i], m[i]);
    }
    template <int J, int NB> PF_HD void inv_split_n(V (&x)[NB], const V (&y)[NB], V (&d)[NB]) const {
#pragma unroll
        for (int i = 0; i < NB; ++i) inv_split<J>(x[i], y[i], d[i]);
    }
    template <int NB> PF_HD void pass_reduce_n(V (&v)[NB]) const { recentre_n<NB>(v); }
    template <int NB> PF_HD void for_dyadic_n(V (&)[NB]) const {}
    PF_HD V add(V a, V b) const { return a + b; }
    template <int NB> PF_HD void canon_n(V (&v)[NB]) const {        // [0,q) from any in-range lazy value
        recentre_n<NB>(v);
#pragma unroll
        for (int i = 0; i < NB; ++i) v[i] = v[i] < 0.0 ? v[i] + q : v[i];
    }
    PF_HD V canon_small(V v) const { return v < 0.0 ? v + q : v; }                       // from |v| < q
    PF_HD V canon_sum(V v) const { V r = v < 0.0 ? v + q : v; return r >= q ? r - q : r; }   // from -q < v < 2q
};

// ArithU64: SEAL's lazy Harvey butterflies.  forward values live in [0,4q), inverse values in [0,2q).
// ArithU64T<false> ("ArithU64"): Harvey's lazy butterflies as SEAL runs them -- values in [0, 4q), one conditional
// subtraction of 2q per butterfly, Shoup products in [0, 2q); any q < 2^61.
// ArithU64T<true> ("ArithU64L", q < 2^56, LOGN <= 15, at most 6 stages per pass): the 64-bit word has 8 spare bits,
// so the range corrections leave the butterflies.
//   * product: m = y*w - c'*q with c' = mulhi64_under(y, w') >= floor(y*w'/2^64) - 2.  Shoup's bound
//     y*w - floor(y*w'/2^64)*q < q*(1 + y/2^64) < 2q holds for EVERY y < 2^64, hence 0 <= m < 4q.
//   * forward: x' = x + m, y' = x + 4q - m: the bound grows by 4q per stage, (1 + 4*15) q = 61 q < 2^62 at the end;
//     canon_n brings a value back with one exact Shoup product by 1 (ratio1 = floor(2^64/q)).  The dyadic Barrett
//     reduction takes the lazy operand as it is (a*b < 61 q^2 < 2^128).
//   * inverse: stage j of a pass (j = 0 first) sees values below 4q*2^j: s = x + y < 4q*2^(j+1),
//     d = x + 4q*2^j - y < 4q*2^(j+1), the product brings d back below 4q; 4q*2^6 = 2^8 q < 2^64.  After a pass the
//     registers that ended as sums are reduced to [0, 2q) (pass_reduce_n, again a Shoup product by 1), so every
//     pass starts below 4q.  The last layer multiplies both halves; canon_small takes [0, 4q).
#if !defined(__HIPCC__) && defined(PF_RANGE_CHECK)
inline unsigned long long pf_range_violations = 0;          // host simulator only: a 64-bit sum that wrapped, a bound that failed
#define PF_RANGE_ASSERT(c) do { if (!(c)) ++pf_range_violations; } while (0)
#else
#define PF_RANGE_ASSERT(c) do { } while (0)
#endif

#ifndef PF_U64_HALF_EXCHANGE
#define PF_U64_HALF_EXCHANGE 1      // measured (r02): 2-3 workgroups per CU beat the 256-register budget by 7-15 % at N <= 16384
#endif
template <bool LAZY>
struct ArithU64T {
    using V = uint64_t;
    using Tw = TwU64;
    using TwR = TwU64;
    static constexpr bool PREFETCH_TW = false;     // 16-byte twiddles: fetched after the exchange (register budget)
    static constexpr bool HALF_EXCHANGE_OK = PF_U64_HALF_EXCHANGE; // 64-bit integer butterflies: 0 = 256-VGPR budget (whole exchange), 1 = half-buffer exchange
    uint64_t q, two_q, ratio0, ratio1;               // ratio = floor(2^128/q)

    static PF_HD V from_u64(uint64_t x) { return x; }
    static PF_HD uint64_t to_u64(V v) { return v; }
    static constexpr bool WQ0_TABLE = false;
    PF_HD TwR resolve(Tw t) const { return t; }
    static PF_HD TwR with_quotient(Tw t, double) { return t; }
    PF_HD V guard(V v) const { return v >= two_q ? v - two_q : v; }
    PF_HD V shoup_one(V v) const { return v - mulhi64(v, ratio1) * q; }                // any v < 2^64 -> [0, 2q)
    // UTW: the twiddles of this call are workgroup-uniform (first pass)
    template <int NB, bool UTW = false>
    PF_HD void mul_tw_n(V (&y)[NB], const TwR (&t)[NB]) const {
        uint64_t hi[NB];
        const uint64_t nq = 0 - q;
#pragma unroll
        for (int i = 0; i < NB; ++i) hi[i] = LAZY ? mulhi64_under(y[i], t[i].wq) : mulhi64(y[i], t[i].wq);
        PF_SCHED_FENCE();
        static_assert(NB % 4 == 0, "products go in fours");
#pragma unroll
        for (int i = 0; i < NB; i += 4) {                                               // y*w - hi*q: [0,2q), LAZY: [0,4q)
            uint64_t yy[4] = {y[i], y[i + 1], y[i + 2], y[i + 3]};
            const uint64_t ww[4] = {t[i].w, t[i + 1].w, t[i + 2].w, t[i + 3].w}, cc[4] = {hi[i], hi[i + 1], hi[i + 2], hi[i + 3]};
            mulsub4_lo64<UTW>(yy, ww, cc, nq);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[i + e] = yy[e];
        }
    }
    // batched butterfly halves (NB a multiple of four): the lazy family subtracts through sub_from_const4
    template <int NB> PF_HD void fwd_combine_n(V (&x)[NB], V (&y)[NB], const V (&m)[NB]) const {
        if constexpr (LAZY && NB % 4 == 0) {
#pragma unroll
            for (int i = 0; i < NB; i += 4) {
                const V mm[4] = {m[i], m[i + 1], m[i + 2], m[i + 3]};
                V d[4];
                sub_from_const4(two_q << 1, mm, d);                                     // 4q - m
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    PF_RANGE_ASSERT(mm[e] < (two_q << 1) && x[i + e] <= ~0ull - (two_q << 1));
                    y[i + e] = x[i + e] + d[e]; x[i + e] = x[i + e] + mm[e];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) fwd_combine(x[i], y[i], m[i]);
        }
    }
    template <int J, int NB> PF_HD void inv_split_n(V (&x)[NB], const V (&y)[NB], V (&d)[NB]) const {
        if constexpr (LAZY && NB % 4 == 0) {
#pragma unroll
            for (int i = 0; i < NB; i += 4) {
                const V yy[4] = {y[i], y[i + 1], y[i + 2], y[i + 3]};
                V t[4];
                sub_from_const4(two_q << (J + 1), yy, t);                               // B - y
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    PF_RANGE_ASSERT(x[i + e] < (two_q << (J + 1)) && yy[e] < (two_q << (J + 1)) && (two_q << (J + 1)) <= (1ull << 63));
                    d[i + e] = x[i + e] + t[e]; x[i + e] = x[i + e] + yy[e];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) inv_split<J>(x[i], y[i], d[i]);
        }
    }
    PF_HD void fwd_combine(V &x, V &y, V m) const {
        if constexpr (LAZY) {
            PF_RANGE_ASSERT(m < (two_q << 1) && x <= ~0ull - (two_q << 1));
            y = x + (two_q << 1) - m; x = x + m;
        }
        else { const V u = guard(x); x = u + m; y = u + two_q - m; }
    }
    template <int J> PF_HD void inv_split(V &x, V y, V &d) const {
        if constexpr (LAZY) {
            PF_RANGE_ASSERT(x < (two_q << (J + 1)) && y < (two_q << (J + 1)) && (two_q << (J + 1)) <= (1ull << 63));
            d = x + (two_q << (J + 1)) - y; x = x + y;
        }
        else { d = x + two_q - y; x = guard(x + y); }
    }
    template <int NB> PF_HD void pass_reduce_n(V (&v)[NB]) const {
        if constexpr (LAZY) {
#pragma unroll
            for (int i = 0; i < NB; ++i) v[i] = shoup_one(v[i]);
        }
    }
    // Barrett 128->64 (SEAL barrett_reduce_128): the quotient estimate is floor(z/q) or one less for any z < 2^128,
    // so one conditional subtraction canonicalises; the operands themselves need not be canonical
    PF_HD V dyadic(V a, V b) const { return barrett128(a * b, mulhi64(a, b)); }
    PF_HD V barrett128(uint64_t z0, uint64_t z1) const {
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
    template <int NB>
    PF_HD void dyadic_n(V (&a)[NB], const V (&b)[NB]) const {
#pragma unroll
        for (int i = 0; i < NB; ++i) a[i] = dyadic(a[i], b[i]);
    }
    PF_HD V add(V a, V b) const { return a + b; }
    template <int NB> PF_HD void canon_n(V (&v)[NB]) const {                          // from the forward transform's range
#pragma unroll
        for (int i = 0; i < NB; ++i) { V r = LAZY ? shoup_one(v[i]) : guard(v[i]); v[i] = r >= q ? r - q : r; }
    }
    template <int NB> PF_HD void for_dyadic_n(V (&)[NB]) const {}
    PF_HD V canon_small(V v) const {                                                  // from the inverse transform's range
        PF_RANGE_ASSERT(v < (LAZY ? two_q << 1 : two_q));
        if constexpr (LAZY) v = guard(v);
        return v >= q ? v - q : v;
    }
    PF_HD V canon_sum(V v) const { V r = v >= two_q ? v - two_q : v; return r >= q ? r - q : r; }   // from [0,3q)
};
using ArithU64 = ArithU64T<false>;
using ArithU64L = ArithU64T<true>;
inline bool u64_lazy_ok(uint64_t q, int logn) { return logn <= 15 && q < (1ull << 56); }

// ------------------------------------------------------------------------------------------------
// Passes
// ------------------------------------------------------------------------------------------------
// Twiddle tables are indexed like SEAL's root_powers: entry m+i (m = 2^s groups, group i) holds
// psi^bitrev(m+i); the inverse table holds the modular inverse of the same entry, except
// inv[1] = psi^-bitrev(1) * N^-1 and inv[0] = N^-1 (both only used by the last inverse layer).
// The butterfly of local bit kb pairs registers k0 (bit kb clear) and k0 | 2^kb; its twiddle is entry
// m + (i >> (b+1)) for the global bit b = a + kb and i the coefficient index of register k0.

#ifndef PF_TW_AHEAD
#define PF_TW_AHEAD 1          // 64-bit back-end: per-lane twiddles requested one batch of butterflies ahead of their use
#endif
constexpr int NBATCH = 4;     // independent butterflies issued together (covers the FP64 dependent-issue latency)

// The twiddles of one pass, fetched up front so the caller can issue them BEFORE the exchange that precedes
// the pass (plain loads stay in flight across s_barrier): entry off(kb) + g is the twiddle of stage kb,
// register group g = k0 >> (kb+1).
template <class G, class A, int PASS>
struct PassTw {
    static constexpr int NL = G::nl(PASS);
    static constexpr int off(int kb) { int o = 0; for (int j = 0; j < kb; ++j) o += G::R >> (j + 1); return o; }
    static constexpr int COUNT = off(NL);
    // 16-byte (u64) twiddles are not held for a whole pass -- 63 of them would be 252 VGPRs -- but fetched where
    // they are used; 8-byte FP64 twiddles are fetched up front (and, register budget permitting, before the exchange).
    static constexpr bool LAZY = !A::PREFETCH_TW;
    // Pass 0's twiddles (table entries 1 .. R-1) are workgroup-uniform and arrive through the scalar cache; for the
    // FP64 back-end their quotients fl(w/q) are tabulated too (entries N .. N+R-1), so resolving them costs no VALU.
    static constexpr bool TABULATED_WQ = PASS == 0 && A::WQ0_TABLE;
    typename A::Tw t[LAZY ? 1 : COUNT];
    double wq[TABULATED_WQ ? COUNT : 1];
    const typename A::Tw *tw_;
    int tb_;

    static constexpr int table_index(int kb, int g) {
        return (1 << (G::LOGN - (G::a(PASS) + kb + 1))) + (G::koff(PASS, g << (kb + 1)) >> (G::a(PASS) + kb + 1));
    }
    PF_HD typename A::Tw fetch(int kb, int g) const {
        if constexpr (PASS == 0) return const_load_tw(tw_ + table_index(kb, g));
        else return (tw_ + table_index(kb, g))[tb_ >> (G::a(PASS) + kb + 1)];   // uniform pointer + per-lane index
    }
    // DESC: request the stages in the order a FORWARD pass consumes them (local bit NL-1 first: one twiddle, then two,
    // four ...), so that the pass can start as soon as the first load is back; an inverse pass starts at local bit 0.
    // [FROM, TO): the stages requested by this call, counted in consumption order -- the first two stages of a forward
    // pass need three twiddles in all and can be asked for before the exchange even when registers are tight.
    template <bool DESC = false, int FROM = 0, int TO = 99>
    PF_HD void load(const typename A::Tw *__restrict__ tw, int tid) {
        tw_ = tw;
        tb_ = PASS == 0 ? 0 : G::base(PASS, tid);                // pass 0: every twiddle is workgroup-uniform
        if constexpr (!LAZY) {
#pragma unroll
            for (int kk = (FROM < NL ? FROM : NL); kk < (TO < NL ? TO : NL); ++kk) {
                const int kb = DESC ? NL - 1 - kk : kk;
#pragma unroll
                for (int g = 0; g < (G::R >> (kb + 1)); ++g) {
                    t[off(kb) + g] = fetch(kb, g);
                    if constexpr (TABULATED_WQ) wq[off(kb) + g] = const_load(reinterpret_cast<const double *>(tw) + G::N + table_index(kb, g));
                }
            }
        }
    }
    PF_HD typename A::TwR get(const A &ar, int kb, int g) const {
        if constexpr (LAZY) return ar.resolve(fetch(kb, g));
        else if constexpr (TABULATED_WQ) return A::with_quotient(t[off(kb) + g], wq[off(kb) + g]);
        else return ar.resolve(t[off(kb) + g]);
    }
};

// One radix-2 stage (local bit KB) over the whole register file, NBATCH butterflies at a time.  Stages are
// template instances, not iterations of a run-time loop: hipcc declines to fully unroll a 6-stage loop over 64
// registers of 64-bit integer butterflies, and any surviving loop would index the register array dynamically.
template <class G, class A, int PASS, int KB>
PF_HD void fwd_stage(typename A::V (&r)[G::R], const A &ar, const PassTw<G, A, PASS> &T) {
    using V = typename A::V;
    using TwR = typename A::TwR;
    // Twiddles fetched where they are used (16-byte entries, per-lane addresses: passes > 0 of the 64-bit back-end) are requested
    // one batch AHEAD: asked for and waited for inside the same batch, each of the ~31 distinct fetches of a pass cost a full
    // L2 round trip with two waves per SIMD to hide it (pass 1 of N = 32768 took 42 k cycles against 24 k for pass 0).
    constexpr bool AHEAD = PassTw<G, A, PASS>::LAZY && PASS != 0 && PF_TW_AHEAD;
    typename A::Tw nxt[NBATCH], nx2[NBATCH];
    if constexpr (AHEAD) {
#pragma unroll
        for (int i = 0; i < NBATCH; ++i) nxt[i] = T.fetch(KB, i >> KB);
        if constexpr (PF_TW_AHEAD >= 2 && NBATCH < G::R / 2) {
#pragma unroll
            for (int i = 0; i < NBATCH; ++i) nx2[i] = T.fetch(KB, (NBATCH + i) >> KB);
        }
    }
#pragma unroll
    for (int bb = 0; bb < G::R / 2; bb += NBATCH) {
        V ys[NBATCH];
        TwR ts[NBATCH];
        typename A::Tw cur[NBATCH];
        if constexpr (AHEAD) {
#pragma unroll
            for (int i = 0; i < NBATCH; ++i) { cur[i] = nxt[i]; if constexpr (PF_TW_AHEAD >= 2) nxt[i] = nx2[i]; }
        }
        PF_SCHED_FENCE();                 // keeps the twiddle resolves (w * 1/q) of later batches from piling up in registers
        if constexpr (AHEAD) {
            constexpr int D = PF_TW_AHEAD >= 2 ? 2 : 1;
            if (bb + D * NBATCH < G::R / 2) {
#pragma unroll
                for (int i = 0; i < NBATCH; ++i) (D == 2 ? nx2[i] : nxt[i]) = T.fetch(KB, (bb + D * NBATCH + i) >> KB);
            }
        }
#pragma unroll
        for (int i = 0; i < NBATCH; ++i) {
            const int b = bb + i, k0 = ((b >> KB) << (KB + 1)) | (b & ((1 << KB) - 1));
            ys[i] = r[k0 | (1 << KB)];
            if constexpr (AHEAD) ts[i] = ar.resolve(cur[i]);
            else ts[i] = T.get(ar, KB, b >> KB);
        }
        ar.template mul_tw_n<NBATCH, PASS == 0>(ys, ts);
        V xs[NBATCH], yo[NBATCH];
#pragma unroll
        for (int i = 0; i < NBATCH; ++i) {
            const int b = bb + i, k0 = ((b >> KB) << (KB + 1)) | (b & ((1 << KB) - 1));
            xs[i] = r[k0];
        }
        ar.template fwd_combine_n<NBATCH>(xs, yo, ys);
#pragma unroll
        for (int i = 0; i < NBATCH; ++i) {
            const int b = bb + i, k0 = ((b >> KB) << (KB + 1)) | (b & ((1 << KB) - 1));
            r[k0] = xs[i]; r[k0 | (1 << KB)] = yo[i];
        }
    }
    if constexpr (KB > 0) fwd_stage<G, A, PASS, KB - 1>(r, ar, T);
}

template <class G, class A, int PASS>
PF_HD void fwd_pass(typename A::V (&r)[G::R], const A &ar, const PassTw<G, A, PASS> &T) {
    fwd_stage<G, A, PASS, G::nl(PASS) - 1>(r, ar, T);
}

template <class G, class A, int PASS, int KB, int KB_HI>
PF_HD void inv_stage(typename A::V (&r)[G::R], const A &ar, const PassTw<G, A, PASS> &T) {
    using V = typename A::V;
    using TwR = typename A::TwR;
    if constexpr (KB <= KB_HI) {
        constexpr bool AHEAD = PassTw<G, A, PASS>::LAZY && PASS != 0 && PF_TW_AHEAD;     // see fwd_stage
        typename A::Tw nxt[NBATCH];
        if constexpr (AHEAD) {
#pragma unroll
            for (int i = 0; i < NBATCH; ++i) nxt[i] = T.fetch(KB, i >> KB);
        }
#pragma unroll
        for (int bb = 0; bb < G::R / 2; bb += NBATCH) {
            V ds[NBATCH];
            TwR ts[NBATCH];
            typename A::Tw cur[NBATCH];
            if constexpr (AHEAD) {
#pragma unroll
                for (int i = 0; i < NBATCH; ++i) cur[i] = nxt[i];
            }
            PF_SCHED_FENCE();
            if constexpr (AHEAD) {
                if (bb + NBATCH < G::R / 2) {
#pragma unroll
                    for (int i = 0; i < NBATCH; ++i) nxt[i] = T.fetch(KB, (bb + NBATCH + i) >> KB);
                }
            }
            V xs[NBATCH], yi[NBATCH];
#pragma unroll
            for (int i = 0; i < NBATCH; ++i) {
                const int b = bb + i, k0 = ((b >> KB) << (KB + 1)) | (b & ((1 << KB) - 1));
                xs[i] = r[k0]; yi[i] = r[k0 | (1 << KB)];
                if constexpr (AHEAD) ts[i] = ar.resolve(cur[i]);
                else ts[i] = T.get(ar, KB, b >> KB);
            }
            ar.template inv_split_n<KB, NBATCH>(xs, yi, ds);
#pragma unroll
            for (int i = 0; i < NBATCH; ++i) {
                const int b = bb + i, k0 = ((b >> KB) << (KB + 1)) | (b & ((1 << KB) - 1));
                r[k0] = xs[i];
            }
            ar.template mul_tw_n<NBATCH, PASS == 0>(ds, ts);
#pragma unroll
            for (int i = 0; i < NBATCH; ++i) {
                const int b = bb + i, k0 = ((b >> KB) << (KB + 1)) | (b & ((1 << KB) - 1));
                r[k0 | (1 << KB)] = ds[i];
            }
        }
        inv_stage<G, A, PASS, KB + 1, KB_HI>(r, ar, T);
    }
}

template <class G, class A, int PASS>
PF_HD void inv_pass(typename A::V (&r)[G::R], const A &ar, const PassTw<G, A, PASS> &T, const typename A::Tw *__restrict__ itw) {
    using V = typename A::V;
    using TwR = typename A::TwR;
    constexpr int kb_hi = G::nl(PASS) - 1 - (PASS == 0 ? 1 : 0);   // pass 0 ends with the N^-1 layer below
    inv_stage<G, A, PASS, 0, kb_hi>(r, ar, T);
    if constexpr (PASS == 0) {                               // last layer: fold N^-1 in (SEAL does the same)
        const TwR tn = A::WQ0_TABLE ? A::with_quotient(const_load_tw(itw), const_load(reinterpret_cast<const double *>(itw) + G::N)) : ar.resolve(const_load_tw(itw));
        const TwR t = A::WQ0_TABLE ? A::with_quotient(const_load_tw(itw + 1), const_load(reinterpret_cast<const double *>(itw) + G::N + 1)) : ar.resolve(const_load_tw(itw + 1));
#pragma unroll
        for (int bb = 0; bb < G::R / 2; bb += NBATCH / 2) {
            V vs[NBATCH];
            TwR ts[NBATCH];
#pragma unroll
            for (int i = 0; i < NBATCH / 2; ++i) {
                const int j = bb + i;
                vs[2 * i] = r[j];
                ar.template inv_split<G::nl(0) - 1>(vs[2 * i], r[j + G::R / 2], vs[2 * i + 1]);
                ts[2 * i] = tn; ts[2 * i + 1] = t;
            }
            ar.template mul_tw_n<NBATCH, true>(vs, ts);
#pragma unroll
            for (int i = 0; i < NBATCH / 2; ++i) { r[bb + i] = vs[2 * i]; r[bb + i + G::R / 2] = vs[2 * i + 1]; }
        }
    }
}

// compile-time loop: f(integral_constant<int, I>) for I = BEGIN, BEGIN+STEP, ... < END (no run-time loop survives,
// so register-array indices stay static whatever the unroller thinks of the body size)
template <int I, int END, int STEP, class F>
PF_HD void static_for(F &&f) {
    if constexpr (I < END) {
        f(std::integral_constant<int, I>{});
        static_for<I + STEP, END, STEP>(f);
    }
}

// element-wise helpers over the whole register file, in batches
// Re-centre after inverse pass PASS.  Registers whose last transformed local bit is set came straight out of a
// modular product (|v| < q) and are left alone; the other half accumulated sums.
template <class G, class A, int PASS>
PF_HD void pass_reduce_all(typename A::V (&r)[G::R], const A &ar) {
    constexpr int TOP = 1 << (G::nl(PASS) - 1);
    constexpr int NB2 = 2 * NBATCH;
    static_for<0, G::R / 2, NB2>([&](auto bbc) {
        constexpr int bb = decltype(bbc)::value;
        typename A::V v[NB2];
#pragma unroll
        for (int i = 0; i < NB2; ++i) { const int c = bb + i; v[i] = r[((c / TOP) * 2 * TOP) | (c % TOP)]; }
        ar.template pass_reduce_n<NB2>(v);
#pragma unroll
        for (int i = 0; i < NB2; ++i) { const int c = bb + i; r[((c / TOP) * 2 * TOP) | (c % TOP)] = v[i]; }
    });
}

template <class G, class A>
PF_HD void canon_all(typename A::V (&r)[G::R], const A &ar) {
    static_for<0, G::R, 2 * NBATCH>([&](auto bbc) {
        constexpr int bb = decltype(bbc)::value;
        typename A::V v[2 * NBATCH];
#pragma unroll
        for (int i = 0; i < 2 * NBATCH; ++i) v[i] = r[bb + i];
        ar.template canon_n<2 * NBATCH>(v);
#pragma unroll
        for (int i = 0; i < 2 * NBATCH; ++i) r[bb + i] = v[i];
    });
}

template <class G, class A, bool LAZY_IN>
PF_HD void dyadic_all(typename A::V (&r)[G::R], const typename A::V (&pv)[G::R], const A &ar) {
    static_for<0, G::R, NBATCH>([&](auto bbc) {
        constexpr int bb = decltype(bbc)::value;
        typename A::V a[NBATCH], b[NBATCH];
#pragma unroll
        for (int i = 0; i < NBATCH; ++i) { a[i] = r[bb + i]; b[i] = pv[bb + i]; }
        if constexpr (LAZY_IN) ar.template for_dyadic_n<NBATCH>(a);
        ar.template dyadic_n<NBATCH>(a, b);
#pragma unroll
        for (int i = 0; i < NBATCH; ++i) r[bb + i] = a[i];
    });
}

// ------------------------------------------------------------------------------------------------
// Exchange through LDS between the layouts of two ADJACENT passes.  `Sync` is a callable: s_barrier on
// the device, std::barrier in the simulator.  slot() is GF(2)-linear, so slot(base | koff) =
// slot(base) ^ slot(koff) with the second factor a compile-time constant per register.
// ------------------------------------------------------------------------------------------------
#ifndef PF_HALF_EXCHANGE
#define PF_HALF_EXCHANGE 1     // exchange through an N/2-entry LDS buffer in two rounds: 32 KiB per workgroup at N = 8192
#endif

template <class G, class A>
struct Xchg {
    // Across every exchange of a 3-pass transform the top register bit of one side is the top thread-id bit of
    // the other (index bits LOGN-1 and YBIT).  Seen over those two bits an exchange is a 2x2 block transpose:
    // round 0 moves the off-diagonal blocks, round 1 the diagonal ones; in each round a thread writes one half
    // of its registers and reads the same half back, so N/2 LDS entries suffice and three workgroups instead of
    // two fit a CU's 160 KiB.  The thread's half is wave-uniform (T >= 128), so the register choice is a scalar
    // branch, not a per-lane select.
    // Forced when N*8 bytes do not fit a CU's LDS at all (N = 32768).
    static constexpr bool FITS_WHOLE = G::N * 8 <= 128 * 1024;
    static constexpr bool HALF = (!FITS_WHOLE || (PF_HALF_EXCHANGE && A::HALF_EXCHANGE_OK)) && G::P == 3 && G::T >= 128;
    static_assert(HALF || FITS_WHOLE, "this ring degree needs the half-buffer exchange");
    static constexpr int LDS_ENTRIES = HALF ? G::N / 2 : G::N;
};

// The two index bits that swap roles in exchange PAIR (layouts PAIR and PAIR + 1): X is the top register bit of layout PAIR,
// Y the top register bit of layout PAIR + 1; each is a thread bit on the other side.  Seen over (X, Y) the exchange is a 2x2
// block transpose.  A thread's half-selector is ITS index bit that is the other side's top register bit; in a round
// exactly the elements with X != Y (round 0) or X == Y (round 1) are in flight, so ONE of the two bits can be dropped from
// the LDS position: the top index bit where it takes part (every geometry whose last pass carries top bits), else X.
// Geo::base places the thread-id bits so that both selectors of an exchange are the same wave-level thread-id bit.
template <class G, int PAIR>
struct XPair {
    static constexpr int X = G::topreg(PAIR), Y = G::topreg(PAIR + 1);
    static constexpr int DROP = (X == G::LOGN - 1 || Y == G::LOGN - 1) ? G::LOGN - 1 : X;
    template <int SIDE> static constexpr int selbit() { return SIDE == PAIR ? Y : X; }
    template <int SIDE> static constexpr int sel_tidbit() { return G::tidbit_of(SIDE, selbit<SIDE>()); }
    static constexpr int compress(int v) { return (v & ((1 << DROP) - 1)) | ((v >> (DROP + 1)) << DROP); }
};

#if defined(__HIPCC__)
PF_HD int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#else
PF_HD int wave_uniform(int v) { return v; }
#endif

// LDS addressing of one side (layout SIDE) of exchange PAIR.  slot() is GF(2)-linear and only rewrites the low
// five bits, so for register k:  position = (slot(base) ^ c_k) + khi_k, with c_k < 32 a compile-time constant
// taking few distinct values and khi_k the (compile-time) high part of koff -- i.e. a handful of base registers
// plus the 16-bit immediate offset of ds_read/ds_write, instead of one XOR per access.
template <class G, int PAIR, int SIDE, bool HALFBUF>
struct XchgAddr {
    int v[32];                                 // slot(base) ^ c for every c; unused entries are dead code
    static constexpr int pos(int s) { return HALFBUF ? XPair<G, PAIR>::compress(s) : s; }     // drops one index bit (above bit 4)
    PF_HD explicit XchgAddr(int tid) {
        const int s0 = G::template slot<PAIR>(G::base(SIDE, tid));
        const int sb = HALFBUF ? ((s0 & ((1 << XPair<G, PAIR>::DROP) - 1)) | ((s0 >> (XPair<G, PAIR>::DROP + 1)) << XPair<G, PAIR>::DROP)) : s0;
#pragma unroll
        for (int c = 0; c < 32; ++c) v[c] = sb ^ c;
    }
    static constexpr int ck(int k) { return G::template slot<PAIR>(G::koff(SIDE, k) & 31) ^ (G::template slot<PAIR>(G::koff(SIDE, k) & ~31) & 31); }
    static constexpr int khi(int k) { return pos(G::koff(SIDE, k) & ~31); }
    template <class V> PF_HD V *at(V *lds, int k) const { return lds + khi(k) + v[ck(k)]; }
};

// one round of the half-buffer exchange: every thread writes register half `sel` and reads the same half back
template <class G, class A, int WR, int RD, class Sync>
PF_HD void half_round(typename A::V (&r)[G::R], typename A::V *lds, const XchgAddr<G, (WR < RD ? WR : RD), WR, true> &aw,
                      const XchgAddr<G, (WR < RD ? WR : RD), RD, true> &ard, int sel, Sync &&sync) {
    constexpr int H = G::R / 2;
    PF_STAMP_X(0);
    sync();
    PF_STAMP_X(1);
    if (sel) {
        PF_BRANCH_TAG("upper half: write");
#pragma unroll
        for (int j = 0; j < H; ++j) *aw.at(lds, H + j) = r[H + j];
        PF_BRANCH_TAG("upper half: written");
    } else {
        PF_BRANCH_TAG("lower half: write");
#pragma unroll
        for (int j = 0; j < H; ++j) *aw.at(lds, j) = r[j];
        PF_BRANCH_TAG("lower half: written");
    }
    PF_STAMP_X(2);
    sync();
    PF_STAMP_X(3);
    if (sel) {
        PF_BRANCH_TAG("upper half: read");
#pragma unroll
        for (int j = 0; j < H; ++j) r[H + j] = *ard.at(lds, H + j);
        PF_BRANCH_TAG("upper half: read done");
    } else {
        PF_BRANCH_TAG("lower half: read");
#pragma unroll
        for (int j = 0; j < H; ++j) r[j] = *ard.at(lds, j);
        PF_BRANCH_TAG("lower half: read done");
    }
    PF_STAMP_X(4);
}

template <class G, class A, int WR, int RD, class Sync>
PF_HD void exchange(typename A::V (&r)[G::R], typename A::V *lds, int tid, Sync &&sync) {
    static_assert(WR - RD == 1 || RD - WR == 1, "exchanges join adjacent passes");
    constexpr int PAIR = WR < RD ? WR : RD;
    if constexpr (!Xchg<G, A>::HALF) {
        const XchgAddr<G, PAIR, WR, false> aw(tid);
        const XchgAddr<G, PAIR, RD, false> ard(tid);
        sync();                                   // previous readers of the buffer are done
#pragma unroll
        for (int k = 0; k < G::R; ++k) *aw.at(lds, k) = r[k];
        sync();
#pragma unroll
        for (int k = 0; k < G::R; ++k) r[k] = *ard.at(lds, k);
    } else {
        using XP = XPair<G, PAIR>;
        const XchgAddr<G, PAIR, WR, true> aw(tid);               // LDS position = slot with one of the two role-swapping bits dropped
        const XchgAddr<G, PAIR, RD, true> ard(tid);
        // This thread's half: as a writer its index bit Y (or X) in layout WR, as a reader its index bit X (or Y) in layout RD.
        // Both must be the same thread-id bit -- a thread reads into the registers it has just written out -- and a
        // wave-level one, so that the register half is chosen by a scalar branch.
        static_assert(XP::template sel_tidbit<WR>() == XP::template sel_tidbit<RD>() && XP::template sel_tidbit<WR>() >= 6,
                      "the layouts of this geometry do not line the half selectors up");
        const int t = wave_uniform((tid >> XP::template sel_tidbit<WR>()) & 1);
        half_round<G, A, WR, RD>(r, lds, aw, ard, t ^ 1, sync);   // off-diagonal blocks
        half_round<G, A, WR, RD>(r, lds, aw, ard, t, sync);       // diagonal blocks
    }
}

// ------------------------------------------------------------------------------------------------
// Whole-transform drivers: forward goes layout 0 -> layout LAST, inverse goes LAST -> 0
// ------------------------------------------------------------------------------------------------
// Each pass's twiddles are requested before the exchange in front of it (when the register budget of the
// arithmetic allows), so their latency overlaps the LDS round trip and the barriers.
template <class G, class A, int WR, int RD, bool HEAD_EARLY = false, class Sync>
PF_HD void xchg_and_load(typename A::V (&r)[G::R], PassTw<G, A, RD> &t, const typename A::Tw *__restrict__ tw,
                         typename A::V *lds, int tid, Sync &&sync) {
    constexpr bool early = A::PREFETCH_TW && !Xchg<G, A>::HALF;   // at three workgroups per CU the registers go to occupancy instead
    constexpr bool fwd = WR < RD;                                  // forward transforms walk the passes upwards
    // HEAD_EARLY (stand-alone forward transforms; measured: -4 % there, +2 % in the fused ct x pt kernel, which is tighter on
    // registers): the first two stages' three twiddles are requested ahead of the exchange
    constexpr int HEAD = (HEAD_EARLY && fwd && A::PREFETCH_TW) ? 2 : 0;
    if constexpr (early) t.template load<fwd>(tw, tid);
    else if constexpr (HEAD > 0) t.template load<fwd, 0, HEAD>(tw, tid);
    exchange<G, A, WR, RD>(r, lds, tid, sync);
    if constexpr (!early) t.template load<fwd, HEAD>(tw, tid);
}

struct NoHook { PF_HD void operator()() const {} };
// `before_last` runs after the last exchange, ahead of the last pass's arithmetic (the fused kernel requests its plaintext there)
template <class G, class A, bool HEAD_EARLY = false, class Sync, class Hook = NoHook>
PF_HD void fwd_all(typename A::V (&r)[G::R], const A &ar, const typename A::Tw *__restrict__ tw,
                   typename A::V *lds, int tid, Sync &&sync, Hook &&before_last = Hook{}) {
    { PassTw<G, A, 0> t0; t0.template load<true>(tw, tid); fwd_pass<G, A, 0>(r, ar, t0); }
    PF_STAMP(2);
    if constexpr (G::P >= 2) { PassTw<G, A, 1> t; xchg_and_load<G, A, 0, 1, HEAD_EARLY>(r, t, tw, lds, tid, sync); PF_STAMP(3); fwd_pass<G, A, 1>(r, ar, t); }
    PF_STAMP(4);
    if constexpr (G::P >= 3) { PassTw<G, A, 2> t; xchg_and_load<G, A, 1, 2, HEAD_EARLY>(r, t, tw, lds, tid, sync); PF_STAMP(5); if constexpr (G::P == 3) before_last(); fwd_pass<G, A, 2>(r, ar, t); }
    PF_STAMP(6);
    if constexpr (G::P >= 4) { PassTw<G, A, 3> t; xchg_and_load<G, A, 2, 3, HEAD_EARLY>(r, t, tw, lds, tid, sync); fwd_pass<G, A, 3>(r, ar, t); }
}

// `tl` = twiddles of the first inverse pass (LAST), which the caller fetched ahead of time
template <class G, class A, class Sync>
PF_HD void inv_all(typename A::V (&r)[G::R], const A &ar, const PassTw<G, A, G::LAST> &tl, const typename A::Tw *__restrict__ itw,
                   typename A::V *lds, int tid, Sync &&sync) {
    inv_pass<G, A, G::LAST>(r, ar, tl, itw);
    PF_STAMP(9);
    if constexpr (G::P >= 4) {
        PassTw<G, A, 2> t;
        pass_reduce_all<G, A, 3>(r, ar);
        xchg_and_load<G, A, 3, 2>(r, t, itw, lds, tid, sync);
        inv_pass<G, A, 2>(r, ar, t, itw);
    }
    if constexpr (G::P >= 3) {
        PassTw<G, A, 1> t;
        pass_reduce_all<G, A, 2>(r, ar);
        xchg_and_load<G, A, 2, 1>(r, t, itw, lds, tid, sync);
        PF_STAMP(10);
        inv_pass<G, A, 1>(r, ar, t, itw);
        PF_STAMP(11);
    }
    if constexpr (G::P >= 2) {
        PassTw<G, A, 0> t;
        pass_reduce_all<G, A, 1>(r, ar);
        xchg_and_load<G, A, 1, 0>(r, t, itw, lds, tid, sync);
        PF_STAMP(12);
        inv_pass<G, A, 0>(r, ar, t, itw);
        PF_STAMP(13);
    }
}

// ------------------------------------------------------------------------------------------------
// Global memory access.  Layout 0: register k <-> coefficient k*T + tid (8 B per lane, lane-contiguous).
// Layout LAST: runs of 2^nl coefficients per lane, 16 B per access when the run allows.
// ------------------------------------------------------------------------------------------------
template <class G, class A>
PF_HD void load_l0(typename A::V (&r)[G::R], const uint64_t *src, int tid) {
#pragma unroll
    for (int k = 0; k < G::R; ++k) r[k] = A::from_u64((src + G::koff(0, k))[tid]);
}

// same, reducing every coefficient modulo (q, ratio1 = high word of floor(2^128/q)) first: the RNS digit of key
// switching is a residue of ANOTHER modulus (SEAL modulo_poly_coeffs / barrett_reduce_64)
template <class G, class A>
PF_HD void load_l0_mod(typename A::V (&r)[G::R], const uint64_t *src, int tid, uint64_t q, uint64_t ratio1) {
#pragma unroll
    for (int k = 0; k < G::R; ++k) {
        const uint64_t x = (src + G::koff(0, k))[tid];
        uint64_t v = x - mulhi64(x, ratio1) * q;
        v = v >= q ? v - q : v;
        r[k] = A::from_u64(v);
    }
}

template <class G, class A>
PF_HD void load_last(typename A::V (&r)[G::R], const uint64_t *src, int tid) {
    constexpr int L = G::LAST, NL = G::nl(L);
    const int b = G::base(L, tid);
    if constexpr (NL >= 1) {
#pragma unroll
        for (int k = 0; k < G::R; k += 2) {
            const U64x2 v = *reinterpret_cast<const U64x2 *>(src + G::koff(L, k) + b);
            r[k] = A::from_u64(v.x); r[k + 1] = A::from_u64(v.y);
        }
    } else {
#pragma unroll
        for (int k = 0; k < G::R; ++k) r[k] = A::from_u64(src[G::koff(L, k) + b]);
    }
}

// registers [K0, K1) only (K0, K1 even): the fused kernel requests part of its plaintext ahead of the last forward pass
template <class G, class A, int K0, int K1>
PF_HD void load_last_part(typename A::V (&r)[G::R], const uint64_t *src, int tid) {
    constexpr int L = G::LAST;
    static_assert(G::nl(L) >= 1 && K0 % 2 == 0 && K1 % 2 == 0, "16-byte accesses");
    const int b = G::base(L, tid);
#pragma unroll
    for (int k = K0; k < K1; k += 2) {
        const U64x2 v = *reinterpret_cast<const U64x2 *>(src + G::koff(L, k) + b);
        r[k] = A::from_u64(v.x); r[k + 1] = A::from_u64(v.y);
    }
}

template <class G>
PF_HD void store_last(const uint64_t (&o)[G::R], uint64_t *dst, int tid) {
    constexpr int L = G::LAST, NL = G::nl(L);
    const int b = G::base(L, tid);
    if constexpr (NL >= 1) {
#pragma unroll
        for (int k = 0; k < G::R; k += 2) *reinterpret_cast<U64x2 *>(dst + G::koff(L, k) + b) = U64x2{o[k], o[k + 1]};
    } else {
#pragma unroll
        for (int k = 0; k < G::R; ++k) dst[G::koff(L, k) + b] = o[k];
    }
}

// Layout LAST -> memory with every store instruction covering whole 64-byte segments.  In layout LAST a lane owns a run of
// 2^nl consecutive coefficients, so a plain 16-byte store per lane lands at a 64-byte (or wider) lane stride; measured on
// MI355X (tools/ubench_stride.hip) such stores run at 0.93 TB/s against 3.27 TB/s lane-contiguous (loads do not care:
// 6.2 against 6.5 TB/s).  Consecutive lanes of a quad own consecutive runs, so the four lanes exchange 16-byte items
// through a wave-private staging area in LDS (the exchange buffer, idle by now): lane l writes its run at l * 80 bytes (the
// 80-byte pitch keeps both sides free of bank conflicts), then for store j lane (4g + i) reads item i of lane (4g + j) and
// writes it behind that lane's run: the quad covers 64 contiguous bytes.  LDS operations of one wave execute in order, so
// the staging area needs no barrier of its own; `sync` is called once up front, because other waves may still be reading
// the last exchange out of the buffer.  Host build (one OS thread per lane): plain stores, same bytes.
template <class G, class V, class Sync>
PF_HD void store_last_staged(const uint64_t (&o)[G::R], uint64_t *dst, int tid, V *lds, Sync &&sync) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int L = G::LAST, NL = G::nl(L);
    constexpr int y = G::P >= 2 ? G::YBIT - NL : 0;            // low y lane bits count runs in order (Geo::base)
    constexpr int PITCH = 80, WAVES = G::T / 64;
    if constexpr (NL >= 3 && y >= 2 && G::T >= 64 && (size_t)WAVES * 64 * PITCH <= sizeof(V) * (G::N / 2)) {
        const int lane = tid & 63, b = G::base(L, tid);
        char *stage = reinterpret_cast<char *>(lds) + (size_t)(tid >> 6) * (64 * PITCH);
        U64x2 *mine = reinterpret_cast<U64x2 *>(stage + lane * PITCH);
        const int quad0 = lane & ~3, i = lane & 3;
        sync();
#pragma unroll
        for (int k = 0; k < G::R; k += 8) {                      // one half-run-or-run of 8 coefficients = 4 items
#pragma unroll
            for (int c = 0; c < 4; ++c) mine[c] = U64x2{o[k + 2 * c], o[k + 2 * c + 1]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const U64x2 v = reinterpret_cast<const U64x2 *>(stage + (quad0 + j) * PITCH)[i];
                // lane (quad0 + j) keeps its 8 coefficients at koff(L, k) + base(quad0 + j) = ... + b + (j - i) * 2^NL
                *reinterpret_cast<U64x2 *>(dst + G::koff(L, k) + b + (j - i) * (1 << NL) + 2 * i) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        store_last<G>(o, dst, tid);
    }
#else
    (void)lds; (void)sync;
    store_last<G>(o, dst, tid);
#endif
}

// ------------------------------------------------------------------------------------------------
// Kernel bodies
// ------------------------------------------------------------------------------------------------
enum : int { CTPT_ACCUMULATE = 1, CTPT_IN_NTT = 2, CTPT_OUT_NTT = 4 };
#ifndef PF_CTPT_TL_EARLY
#define PF_CTPT_TL_EARLY 0
#endif
#ifndef PF_CTPT_PT_EARLY
#define PF_CTPT_PT_EARLY 8      // plaintext registers requested before the last forward pass (0: all of it after the pass; measured at N = 8192: 8 -> -2.2 %, 16 spills)
#endif

// forward NTT of one limb-polynomial: natural-order coefficients in, bit-reversed evaluations out
template <class G, class A, class Sync>
PF_HD void body_ntt_fwd(const A &ar, const typename A::Tw *__restrict__ tw, const uint64_t *src,
                        uint64_t *dst, typename A::V *lds, int tid, Sync &&sync) {
    typename A::V r[G::R];
    PF_STAMP(0);
    load_l0<G, A>(r, src, tid);
    PF_STAMP(1);
    fwd_all<G, A, true>(r, ar, tw, lds, tid, sync);
    canon_all<G, A>(r, ar);
    PF_STAMP(7);
    uint64_t o[G::R];
#pragma unroll
    for (int k = 0; k < G::R; ++k) o[k] = A::to_u64(r[k]);
    store_last_staged<G>(o, dst, tid, lds, sync);
    PF_STAMP(8);
}

// forward NTT of an RNS digit under another modulus: dst = NTT_q(src mod q)   (key switching, step 1)
template <class G, class A, class Sync>
PF_HD void body_ntt_fwd_mod(const A &ar, const typename A::Tw *__restrict__ tw, const uint64_t *src, uint64_t *dst,
                            uint64_t q, uint64_t ratio1, typename A::V *lds, int tid, Sync &&sync) {
    typename A::V r[G::R];
    load_l0_mod<G, A>(r, src, tid, q, ratio1);
    fwd_all<G, A, true>(r, ar, tw, lds, tid, sync);
    canon_all<G, A>(r, ar);
    uint64_t o[G::R];
#pragma unroll
    for (int k = 0; k < G::R; ++k) o[k] = A::to_u64(r[k]);
    store_last_staged<G>(o, dst, tid, lds, sync);
}

template <class G, class A, class Sync>
PF_HD void body_ntt_inv(const A &ar, const typename A::Tw *__restrict__ itw, const uint64_t *src,
                        uint64_t *dst, typename A::V *lds, int tid, Sync &&sync) {
    typename A::V r[G::R];
    PassTw<G, A, G::LAST> tl;
    tl.load(itw, tid);
    load_last<G, A>(r, src, tid);
    inv_all<G, A>(r, ar, tl, itw, lds, tid, sync);
#pragma unroll
    for (int k = 0; k < G::R; ++k) (dst + G::koff(0, k))[tid] = A::to_u64(ar.canon_small(r[k]));
}

// ct x pt for one limb-polynomial: [NTT] -> dyadic with pt (NTT form) -> [INTT] -> [+= out]
template <class G, class A, int FLAGS, class Sync>
PF_HD void body_ctpt(const A &ar, const typename A::Tw *__restrict__ tw, const typename A::Tw *__restrict__ itw,
                     const uint64_t *ct, const uint64_t *pt, uint64_t *out,
                     typename A::V *lds, int tid, Sync &&sync) {
    using V = typename A::V;
    V r[G::R], pv[G::R];
    PassTw<G, A, G::LAST> tl;                 // twiddles of the first inverse pass, requested with the plaintext
    if constexpr (FLAGS & CTPT_IN_NTT) {
        load_last<G, A>(r, ct, tid);
        load_last<G, A>(pv, pt, tid);
        dyadic_all<G, A, false>(r, pv, ar);
        if constexpr (!(FLAGS & CTPT_OUT_NTT)) tl.load(itw, tid);
    } else {
        PF_STAMP(0);
        load_l0<G, A>(r, ct, tid);
        PF_STAMP(1);
#if PF_CTPT_PT_EARLY
        // the plaintext limb is read in the layout the forward transform ends in (lane-contiguous from HBM), and requested BEFORE the
        // last pass's three stages: its trip to memory runs under them instead of in front of the dyadic product
        if constexpr (G::P == 3 && A::PREFETCH_TW && G::nl(G::LAST) >= 1 && G::LOGN <= 13) {      // (N >= 16384: the FP64 kernels spill already)
            constexpr int EARLY = PF_CTPT_PT_EARLY < G::R ? PF_CTPT_PT_EARLY : G::R;      // registers of the plaintext requested early
            fwd_all<G, A>(r, ar, tw, lds, tid, sync, [&] { load_last_part<G, A, 0, EARLY>(pv, pt, tid); });
            load_last_part<G, A, EARLY, G::R>(pv, pt, tid);
        } else { fwd_all<G, A>(r, ar, tw, lds, tid, sync); load_last<G, A>(pv, pt, tid); }
#else
        fwd_all<G, A>(r, ar, tw, lds, tid, sync);
        // the plaintext limb is read in the layout the forward transform ended in: lane-contiguous from HBM
        load_last<G, A>(pv, pt, tid);
#endif
        PF_STAMP(7);
#if PF_CTPT_TL_EARLY
        if constexpr (!(FLAGS & CTPT_OUT_NTT)) tl.load(itw, tid);      // the first inverse pass's twiddles travel under the dyadic product
        dyadic_all<G, A, true>(r, pv, ar);
#else
        dyadic_all<G, A, true>(r, pv, ar);
        if constexpr (!(FLAGS & CTPT_OUT_NTT)) tl.load(itw, tid);
#endif
        PF_STAMP(8);
    }
    if constexpr (FLAGS & CTPT_OUT_NTT) {
        uint64_t o[G::R];
        if constexpr (FLAGS & CTPT_ACCUMULATE) load_last<G, A>(pv, out, tid);
#pragma unroll
        for (int k = 0; k < G::R; ++k) {
            V v = ar.canon_small(r[k]);
            if constexpr (FLAGS & CTPT_ACCUMULATE) v = ar.canon_sum(ar.add(v, pv[k]));
            o[k] = A::to_u64(v);
        }
        store_last<G>(o, out, tid);
        return;
    }
    if constexpr (!(FLAGS & CTPT_IN_NTT)) PF_LAUNDER(tid);   // the inverse half recomputes its LDS addresses: cheaper than keeping the forward half's alive
    inv_all<G, A>(r, ar, tl, itw, lds, tid, sync);
#pragma unroll
    for (int k = 0; k < G::R; ++k) {
        V v = ar.canon_small(r[k]);
        if constexpr (FLAGS & CTPT_ACCUMULATE) v = ar.canon_sum(ar.add(v, A::from_u64((out + G::koff(0, k))[tid])));
        (out + G::koff(0, k))[tid] = A::to_u64(v);
    }
    PF_STAMP(14);
}

// Forward transform of a polynomial that is PRODUCED instead of read: `load(r, tid)` fills register k with coefficient
// koff(0, k) + tid (canonical).  Used for plaintexts packed on the fly from base rows.
template <class G, class A, class Loader, class Sync>
PF_HD void body_ntt_fwd_from(const A &ar, const typename A::Tw *__restrict__ tw, const Loader &load, uint64_t *dst, typename A::V *lds,
                             int tid, Sync &&sync) {
    typename A::V r[G::R];
    load(r, tid);
    fwd_all<G, A, true>(r, ar, tw, lds, tid, sync);
    canon_all<G, A>(r, ar);
    uint64_t o[G::R];
#pragma unroll
    for (int k = 0; k < G::R; ++k) o[k] = A::to_u64(r[k]);
    store_last_staged<G>(o, dst, tid, lds, sync);
}

// Encrypted precise search in one pass (pf_ct_rows_mul): the plaintext is produced by `load` (as in body_ntt_fwd_from),
// transformed ONCE and kept in registers while both components of the ciphertext (NTT form) are multiplied by it and
// transformed back -- the NTT-form plaintext never exists in memory.  64 more live registers than body_ctpt: budgeted
// for 2 workgroups per CU instead of 3.
template <class G, class A, class Loader, class Sync>
PF_HD void body_rows_ctpt(const A &ar, const typename A::Tw *__restrict__ tw, const typename A::Tw *__restrict__ itw, const Loader &load,
                          const uint64_t *ct0, const uint64_t *ct1, uint64_t *out0, uint64_t *out1, typename A::V *lds, int tid, Sync &&sync) {
    using V = typename A::V;
    V pv[G::R];
    load(pv, tid);
    fwd_all<G, A, true>(pv, ar, tw, lds, tid, sync);
    canon_all<G, A>(pv, ar);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        V r[G::R];
        PassTw<G, A, G::LAST> tl;
        load_last<G, A>(r, c ? ct1 : ct0, tid);
        tl.load(itw, tid);
        dyadic_all<G, A, false>(r, pv, ar);
        PF_LAUNDER(tid);
        inv_all<G, A>(r, ar, tl, itw, lds, tid, sync);
        uint64_t *out = c ? out1 : out0;
#pragma unroll
        for (int k = 0; k < G::R; ++k) (out + G::koff(0, k))[tid] = A::to_u64(ar.canon_small(r[k]));
    }
}

}  // namespace pf
