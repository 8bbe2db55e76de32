// ks_split.hpp -- key switching at N = 32768 as a two-kernel split of the digit transforms (round 3).
//
// SEAL's Evaluator::switch_key_inplace (linked un-vendored by /root/reference/CMakeLists.txt:33-38,66; restated in
// oracle/pf_oracle.c: pfo_key_switch) needs, per switched polynomial, D x K forward transforms of 32768 points (config 5:
// 15 x 16 = 240), each multiplied into two 128-bit accumulators.  The single-kernel transform (ntt_core.hpp) gives a whole CU
// to one workgroup at this degree (128 KiB of LDS, every register): its load, pass, exchange and store phases run one after
// the other (VALU ~ 50 % busy) and the finished transforms travel through memory twice more for the multiply-accumulate.
// Here the 15 stages are cut after the seventh:
//
//   pass A (body_ksA)  index = r * 256 + c.  A workgroup of 512 threads takes a tile of all 128 rows r x 64 adjacent columns c
//                      of ONE (digit, key modulus) transform and runs stages 0..6 (index bits 14..8) on it: four stages in
//                      registers (thread = one (r mod 8, c), registers = r div 8), one exchange through LDS, three more
//                      (wave = two values of r div 8, lane = c, registers = r mod 8).  Every twiddle of pass A is wave-uniform:
//                      scalar loads, scalar operands.  The tile goes back to memory lazily reduced (values < 2^62).
//   pass B (body_ksB)  a wave takes two of the 128 contiguous 256-point blocks (one per half-wave: 32 lanes x 8 coefficients)
//                      of ONE (ciphertext, key modulus) and loops over the D digits: load the block of pass A's output,
//                      finish stages 7..14 (3 + 3 + 2 stages in registers, two wave-private transposes through LDS, no
//                      workgroup barrier anywhere), multiply by the two key components on the same 8 consecutive
//                      coefficients and add into 128-bit accumulators that stay in registers; after the loop ONE Barrett
//                      reduction and one store per component.  Finished digit transforms never exist in memory.
//
// Arithmetic: the lazy 64-bit family only (ArithU64L: every modulus below 2^56).  A digit is a residue of ANOTHER modulus
// below 2^56; the lazy forward butterflies take it as it is (x' = x + m, y' = x + 4q - m with 0 <= m < 4q for ANY y < 2^64:
// after 15 stages a value is below 2^56 + 60 q < 2^62), so neither the reduction modulo m_J on load nor the canonical
// form after the last stage is needed: the 128-bit sums take 15 x 2^62 x 2^56 < 2^122.
//
// Like ntt_core.hpp the file compiles for the device and, unchanged, for the host (tests/cpp/sim_ntt.cpp: one OS thread per
// lane), which checks the index maps against the oracle without a GPU.
#pragma once
#include "ntt_core.hpp"

namespace pf {

struct KsGeo {
    static constexpr int LOGN = 15, N = 1 << LOGN;
    static constexpr int A_T = 512, A_R = 16, A_COLS = 64, A_TILES = 256 / A_COLS;      // pass A: 128 rows x 64 columns per workgroup
    static constexpr int A_LDS = 128 * A_COLS;                                            // u64 entries
    static constexpr int B_T = 256, B_R = 8, B_CHUNK = 2048, B_CHUNKS = N / B_CHUNK;     // pass B: 8 blocks of 256 per workgroup
    static constexpr int B_LDS_WAVE = 2 * 320;                                            // u64 entries per wave (two padded blocks)
    static constexpr int B_LDS = (B_T / 64) * B_LDS_WAVE;
};

// One forward stage over NR registers: butterfly b pairs registers k0 (bit KB clear) and k0 | 2^KB, its twiddle is twf(b >> KB).
template <int NR, int KB, bool UTW, class A, class TwFn>
PF_HD void ks_fwd_stage(typename A::V (&r)[NR], const A &ar, TwFn &&twf) {
    using V = typename A::V;
    static_for<0, NR / 2, 4>([&](auto bbc) {
        constexpr int bb = decltype(bbc)::value;
        V ys[4], xs[4], yo[4];
        typename A::TwR ts[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = bb + i, k0 = ((b >> KB) << (KB + 1)) | (b & ((1 << KB) - 1));
            ys[i] = r[k0 | (1 << KB)];
            ts[i] = twf(b >> KB);
        }
        ar.template mul_tw_n<4, UTW>(ys, ts);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = bb + i, k0 = ((b >> KB) << (KB + 1)) | (b & ((1 << KB) - 1));
            xs[i] = r[k0];
        }
        ar.template fwd_combine_n<4>(xs, yo, ys);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = bb + i, k0 = ((b >> KB) << (KB + 1)) | (b & ((1 << KB) - 1));
            r[k0] = xs[i]; r[k0 | (1 << KB)] = yo[i];
        }
    });
}

// One inverse (Gentleman-Sande) stage over NR registers: (x, y) -> (x + y, (x - y) * w), w = twf(b >> KB) from the INVERSE table
// (same entry index as the forward stage it undoes).  J = stages since the values were last below 4q (ArithU64L's contract:
// stage J sees values below 4q * 2^J and keeps the differences' products below 4q).
template <int NR, int KB, int J, bool UTW, class A, class TwFn>
PF_HD void ks_inv_stage(typename A::V (&r)[NR], const A &ar, TwFn &&twf) {
    using V = typename A::V;
    static_for<0, NR / 2, 4>([&](auto bbc) {
        constexpr int bb = decltype(bbc)::value;
        V xs[4], yi[4], ds[4];
        typename A::TwR ts[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = bb + i, k0 = ((b >> KB) << (KB + 1)) | (b & ((1 << KB) - 1));
            xs[i] = r[k0]; yi[i] = r[k0 | (1 << KB)];
            ts[i] = twf(b >> KB);
        }
        ar.template inv_split_n<J, 4>(xs, yi, ds);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = bb + i, k0 = ((b >> KB) << (KB + 1)) | (b & ((1 << KB) - 1));
            r[k0] = xs[i];
        }
        ar.template mul_tw_n<4, UTW>(ds, ts);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = bb + i, k0 = ((b >> KB) << (KB + 1)) | (b & ((1 << KB) - 1));
            r[k0 | (1 << KB)] = ds[i];
        }
    });
}

template <int NR, class A>
PF_HD void ks_reduce_all(typename A::V (&r)[NR], const A &ar) {          // any value below 2^64 -> [0, 2q)
    static_for<0, NR, 8>([&](auto bbc) {
        constexpr int bb = decltype(bbc)::value;
        typename A::V v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = r[bb + i];
        ar.template pass_reduce_n<8>(v);
#pragma unroll
        for (int i = 0; i < 8; ++i) r[bb + i] = v[i];
    });
}

// ------------------------------------------------------------------------------------------------------------------
// Pass A: stages 0..6 of one (digit, modulus) transform on the tile of columns [64 cb, 64 cb + 64).
//   src  the digit polynomial (coefficient form, residues of ITS modulus: any value below 2^56)
//   dst  this transform's slot of the intermediate buffer, natural index order
// `sync` is the workgroup barrier.
// ------------------------------------------------------------------------------------------------------------------
template <class A, class Sync>
PF_HD void body_ksA(const A &ar, const TwU64 *__restrict__ tw, const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, int cb,
                    uint64_t *lds, int tid, Sync &&sync) {
    static_assert(std::is_same<typename A::V, uint64_t>::value, "the split key switch runs the 64-bit lazy family");
    const int lane = tid & 63, w = wave_uniform(tid >> 6);
    uint64_t r[16];
    // registers = r div 8 (index bits 14..11), wave = r mod 8, lane = column
    const uint64_t *s0 = src + (size_t)w * 256 + cb * 64 + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = s0[(size_t)k * 2048];
    // stages 0..3: the twiddle of stage s, group g is table entry 2^s + g -- workgroup-uniform
    ks_fwd_stage<16, 3, true>(r, ar, [&](int g) { return const_load_tw(tw + 1 + g); });
    ks_fwd_stage<16, 2, true>(r, ar, [&](int g) { return const_load_tw(tw + 2 + g); });
    ks_fwd_stage<16, 1, true>(r, ar, [&](int g) { return const_load_tw(tw + 4 + g); });
    ks_fwd_stage<16, 0, true>(r, ar, [&](int g) { return const_load_tw(tw + 8 + g); });
    // exchange: LDS position of (row, column) is row * 64 + column; lanes stay columns on both sides
    uint64_t *l0 = lds + w * 64 + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) l0[k * 512] = r[k];
    sync();
    // registers = (g, r mod 8) with r div 8 = 2 w + g
    const uint64_t *l1 = lds + w * 1024 + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = l1[k * 64];
    // stages 4..6 (index bits 10..8 = register bits 2..0): entry 2^s + ((r div 8) << (s - 4)) + (r mod 8 >> (7 - s)); with
    // b the butterfly number over the 16 registers that is 2^s + (2 w << (s - 4)) + (b >> KB): wave-uniform
    const TwU64 *t4 = tw + 16 + 2 * w, *t5 = tw + 32 + 4 * w, *t6 = tw + 64 + 8 * w;
    ks_fwd_stage<16, 2, true>(r, ar, [&](int g) { return const_load_tw(t4 + g); });
    ks_fwd_stage<16, 1, true>(r, ar, [&](int g) { return const_load_tw(t5 + g); });
    ks_fwd_stage<16, 0, true>(r, ar, [&](int g) { return const_load_tw(t6 + g); });
    uint64_t *d0 = dst + (size_t)w * 4096 + cb * 64 + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) d0[(size_t)k * 256] = r[k];
}

// ------------------------------------------------------------------------------------------------------------------
// Pass B.  Per half-wave (32 lanes, lane l) one 256-point block; c = index inside the block, bits i7..i0.
//   R1: registers (i7 i6 i0), lanes (i5 i4 i3 i2 i1)          stages 7, 8      (memory: 16 bytes per lane, 512 per half-wave)
//   R2: registers (i5 i4 i3), lanes (i7 i6 i0 | i2 i1)        stages 9, 10, 11
//   R3: registers (i2 i1 i0), lanes (i7 i6 i5 i4 i3)          stages 12, 13, 14 (a lane owns 8 consecutive coefficients)
// LDS positions (u64 entries inside the half-wave's 320-entry area): exchange 1 uses c + 8 (c >> 6), exchange 2 uses
// c + 2 (c >> 3) -- both free of bank conflicts on their narrow side (ds_read_b64 over the 32 lanes of R2; ds_write_b64 of R2
// and the 80-byte lane pitch of R3's 16-byte reads).  LDS operations of one wave execute in order; `wsync` keeps the
// compiler (device) or the other lanes' threads (host) in step.
// ------------------------------------------------------------------------------------------------------------------
PF_HD constexpr int ksb_p1(int c) { return c + 8 * (c >> 6); }
PF_HD constexpr int ksb_p2(int c) { return c + 2 * (c >> 3); }

// 128-bit lazy multiply-accumulate, four at a time: (hi_i : lo_i) += x_i * k_i with x_i < 2^62, k_i < 2^56 and sums below 2^128
// (the caller's bound: at most 63 terms).  hipcc expands the C form (x * k, mulhi64, compare, add) into 24 instructions per
// term, most of them moves into the aligned register pairs that 64-bit multiplies want.  Here, per term, with x = x1:x0, k = k1:k0:
//     M  = x1 k0 + x0 k1                 2 v_mad_u64_u32; below 2^63, no carry
//     lo = x0 k0 + lo                    1 v_mad_u64_u32, carry-out cA (weight 2^64)
//     hi = x1 k1 + hi                    1 v_mad_u64_u32 (sums stay below 2^128: hi never wraps)
//     lo.high += M.low                   1 v_add_co_u32,  carry-out c1 (weight 2^64)
//     t  = M.high + c1 + cA              2 v_addc_co_u32 (M.high < 2^31: no wrap)
//     hi = t * 1 + hi                    1 v_mad_u64_u32
// 8 instructions.  The halves of a pair are named by 32-bit operands of a second asm statement (the compiler keeps them in place, as
// in mulsub4_lo64); the four terms are interleaved so that no carry is read within two states of its write.
#if defined(__HIP_DEVICE_COMPILE__)
#define PF_MC_M0(i) "v_mad_u64_u32 %[M" #i "], %[cy], %[x" #i "1], %[k" #i "0], 0\n\t"
#define PF_MC_M1(i) "v_mad_u64_u32 %[M" #i "], %[cy], %[x" #i "0], %[k" #i "1], %[M" #i "]\n\t"
#define PF_MC_LO(i) "v_mad_u64_u32 %[lo" #i "], %[cA" #i "], %[x" #i "0], %[k" #i "0], %[lo" #i "]\n\t"
#define PF_MC_HI(i) "v_mad_u64_u32 %[hi" #i "], %[cy], %[x" #i "1], %[k" #i "1], %[hi" #i "]\n\t"
#define PF_MC_OUT1(i) [M##i] "=&v"(M[i]), [lo##i] "+v"(lo[i]), [hi##i] "+v"(hi[i]), [cA##i] "=&s"(cA[i])
#define PF_MC_IN1(i) [x##i##0] "v"((uint32_t)x[i]), [x##i##1] "v"((uint32_t)(x[i] >> 32)), [k##i##0] "v"((uint32_t)k[i]), [k##i##1] "v"((uint32_t)(k[i] >> 32))
#define PF_MC_A1(i) "v_add_co_u32 %[a" #i "], %[c" #i "], %[a" #i "], %[ml" #i "]\n\t"
#define PF_MC_T0(i) "v_addc_co_u32 %[t" #i "], %[c" #i "], %[mh" #i "], 0, %[c" #i "]\n\t"
#define PF_MC_T1(i) "v_addc_co_u32 %[t" #i "], %[cA" #i "], %[t" #i "], 0, %[cA" #i "]\n\t"
#define PF_MC_H2(i) "v_mad_u64_u32 %[hi" #i "], %[cy], %[t" #i "], 1, %[hi" #i "]\n\t"
#define PF_MC_OUT2(i) [a##i] "+v"(a1[i]), [t##i] "=&v"(t[i]), [hi##i] "+v"(hi[i]), [c##i] "=&s"(c1[i]), [cA##i] "+s"(cA[i])
#define PF_MC_IN2(i) [ml##i] "v"((uint32_t)M[i]), [mh##i] "v"((uint32_t)(M[i] >> 32))
PF_HD void ks_mac128x4(uint64_t (&lo)[4], uint64_t (&hi)[4], const uint64_t (&x)[4], const uint64_t (&k)[4]) {
    uint64_t M[4], cA[4], c1[4], cy;
    asm(PF_MS_ALL(PF_MC_M0) PF_MS_ALL(PF_MC_M1) PF_MS_ALL(PF_MC_LO) PF_MS_ALL(PF_MC_HI)
        : PF_MC_OUT1(0), PF_MC_OUT1(1), PF_MC_OUT1(2), PF_MC_OUT1(3), [cy] "=&s"(cy)
        : PF_MC_IN1(0), PF_MC_IN1(1), PF_MC_IN1(2), PF_MC_IN1(3));
    uint32_t a1[4], t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a1[i] = (uint32_t)(lo[i] >> 32);
    asm(PF_MS_ALL(PF_MC_A1) PF_MS_ALL(PF_MC_T0) PF_MS_ALL(PF_MC_T1) PF_MS_ALL(PF_MC_H2)
        : PF_MC_OUT2(0), PF_MC_OUT2(1), PF_MC_OUT2(2), PF_MC_OUT2(3), [cy] "=&s"(cy)
        : PF_MC_IN2(0), PF_MC_IN2(1), PF_MC_IN2(2), PF_MC_IN2(3));
#pragma unroll
    for (int i = 0; i < 4; ++i) lo[i] = ((uint64_t)a1[i] << 32) | (uint32_t)lo[i];
    asm("" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]));              // keeps the pairs whole (see mulsub4_lo64)
}
#else
PF_HD void ks_mac128x4(uint64_t (&lo)[4], uint64_t (&hi)[4], const uint64_t (&x)[4], const uint64_t (&k)[4]) {
    for (int i = 0; i < 4; ++i) {
        const uint64_t pl = x[i] * k[i], ph = mulhi64(x[i], k[i]);
        const uint64_t s = lo[i] + pl;
        hi[i] += ph + (s < pl ? 1 : 0);
        lo[i] = s;
    }
}
#endif

// The 17 per-lane twiddles of stages 7..14 of one block: they depend on (modulus, block, lane) only -- not on the digit nor on the
// ciphertext -- so pass B fetches them once and keeps them in registers (68 VGPRs) for its whole digit loop.
struct KsbTw {
    TwU64 s7, s8[2];                 // R1
    TwU64 s9, s10[2], s11[4];        // R2
    TwU64 s12, s13[2], s14[4];       // R3
    PF_HD void load(const TwU64 *__restrict__ tw, int blk, int l) {
        s7 = tw[128 + blk]; s8[0] = tw[256 + 2 * blk]; s8[1] = tw[257 + 2 * blk];
        // stage 7 + u, u = 2, 3, 4: entry 2^(7+u) + ((4 blk + (i7 i6)) << (u - 2)) + group, (i7 i6) = l >> 3 in R2
        const int e2 = 4 * blk + (l >> 3);
        s9 = tw[512 + e2]; s10[0] = tw[1024 + 2 * e2]; s10[1] = tw[1025 + 2 * e2];
#pragma unroll
        for (int g = 0; g < 4; ++g) s11[g] = tw[2048 + 4 * e2 + g];
        // stage 7 + u, u = 5, 6, 7: entry 2^(7+u) + ((32 blk + l) << (u - 5)) + group
        const int e3 = 32 * blk + l;
        s12 = tw[4096 + e3]; s13[0] = tw[8192 + 2 * e3]; s13[1] = tw[8193 + 2 * e3];
#pragma unroll
        for (int g = 0; g < 4; ++g) s14[g] = tw[16384 + 4 * e3 + g];
    }
};

// Stages 7..14 of one block, in place in r (R1 order in, R3 order out).  l = lane & 31, area = this half-wave's LDS area.
// `mid` runs between the second exchange and the last three stages (the caller requests the key there: late enough to keep 32
// registers free during the first five stages, early enough for the L2 round trip to hide under the last three).
template <class A, class WSync, class Mid>
PF_HD void ksb_finish_fwd(uint64_t (&r)[8], const A &ar, const KsbTw &T, int l, uint64_t *area, WSync &&wsync, Mid &&mid) {
    // R1: stage 7 (register bit 2 = i7), stage 8 (register bit 1 = i6; group = i7)
    ks_fwd_stage<8, 2, false>(r, ar, [&](int) { return T.s7; });
    ks_fwd_stage<8, 1, false>(r, ar, [&](int g) { return T.s8[g]; });
    // exchange 1: write registers (k2 k1 | k0) at c = 128 k2 + 64 k1 + 2 l + k0, 16 bytes per store
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        *reinterpret_cast<U64x2 *>(area + ksb_p1(128 * (kk >> 1) + 64 * (kk & 1)) + 2 * l) = U64x2{r[2 * kk], r[2 * kk + 1]};
    wsync();
    // R2: lane = (i7 i6 i0 | i2 i1), register k = (i5 i4 i3): c = 128 l4 + 64 l3 + 8 k + 2 (l & 3) + l2
    const int hi2 = l >> 3;                                          // (i7 i6)
    {
        const uint64_t *rd = area + ksb_p1(64 * hi2) + 2 * (l & 3) + ((l >> 2) & 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = rd[8 * k];
    }
    ks_fwd_stage<8, 2, false>(r, ar, [&](int) { return T.s9; });
    ks_fwd_stage<8, 1, false>(r, ar, [&](int g) { return T.s10[g]; });
    ks_fwd_stage<8, 0, false>(r, ar, [&](int g) { return T.s11[g]; });
    wsync();                                                         // every lane has read exchange 1 out
    // exchange 2: write register k at the same c, positions c + 2 (c >> 3)
    {
        uint64_t *wr = area + ksb_p2(64 * hi2) + 2 * (l & 3) + ((l >> 2) & 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) wr[10 * k] = r[k];
    }
    wsync();
    // R3: lane l owns c = 8 l + k
    {
        const U64x2 *rd = reinterpret_cast<const U64x2 *>(area + 10 * l);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const U64x2 v = rd[j]; r[2 * j] = v.x; r[2 * j + 1] = v.y; }
    }
    mid();
    ks_fwd_stage<8, 2, false>(r, ar, [&](int) { return T.s12; });
    ks_fwd_stage<8, 1, false>(r, ar, [&](int g) { return T.s13[g]; });
    ks_fwd_stage<8, 0, false>(r, ar, [&](int g) { return T.s14[g]; });
    wsync();                                                         // exchange 2 read out before the next digit's exchange 1
}

// x        pass A's output for (this ciphertext, digit 0, this modulus); digit I sits at x + I * x_stride
// ksk      key for (digit 0, component 0, this modulus): component c of digit I at ksk + (2 I + c) * k_stride
// out0/1   accumulated products of the two components for (this ciphertext, this modulus): NTT form, canonical (INV = false), or
//          after the first eight inverse stages, reduced to [0, 2q), natural index order (INV = true: pass C continues)
// chunk    which 2048 coefficients (0..15) this workgroup covers
template <class A, bool INV, class WSync>
PF_HD void body_ksB(const A &ar, const TwU64 *__restrict__ tw, const TwU64 *__restrict__ itw, const uint64_t *__restrict__ x, size_t x_stride,
                    const uint64_t *__restrict__ ksk, size_t k_stride, uint64_t *__restrict__ out0, uint64_t *__restrict__ out1, int D, int chunk,
                    uint64_t *lds, int tid, WSync &&wsync) {
    static_assert(std::is_same<typename A::V, uint64_t>::value, "the split key switch runs the 64-bit lazy family");
    const int lane = tid & 63, wv = tid >> 6, h = lane >> 5, l = lane & 31;
    const int blk = chunk * 8 + wv * 2 + h;
    uint64_t *area = lds + wv * KsGeo::B_LDS_WAVE + h * 320;
    const size_t base = (size_t)blk * 256;
    uint64_t lo[2][8], hi[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) lo[c][e] = hi[c][e] = 0;
    KsbTw T;
    T.load(tw, blk, l);
    // the block of digit I + 1 is requested while digit I is worked on (16 registers); the key of digit I at the top of its
    // iteration -- it is only needed after the eight stages
    // addresses = a workgroup-uniform base (scalar registers) + a 32-bit per-lane byte offset: no 64-bit address pairs held per lane
    const uint32_t xoff = (uint32_t)(base + 2 * l) * 8u, koff = (uint32_t)(base + 8 * l) * 8u;
    auto ld16 = [](const uint64_t *ubase, uint32_t byte_off) {
        return *reinterpret_cast<const U64x2 *>(reinterpret_cast<const char *>(ubase) + byte_off);
    };
    U64x2 xn[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xn[kk] = ld16(x, xoff + 8u * (128 * (kk >> 1) + 64 * (kk & 1)));
    for (int I = 0; I < D; ++I) {
        uint64_t r[8];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { r[2 * kk] = xn[kk].x; r[2 * kk + 1] = xn[kk].y; }
        {
            const uint64_t *xnext = x + (size_t)(I + 1 < D ? I + 1 : I) * x_stride;      // last digit: a harmless re-read
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xn[kk] = ld16(xnext, xoff + 8u * (128 * (kk >> 1) + 64 * (kk & 1)));
        }
        PF_SCHED_FENCE();
        U64x2 kv[2][4];
        ksb_finish_fwd(r, ar, T, l, area, wsync, [&] {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint64_t *kp = ksk + (size_t)(2 * I + c) * k_stride;
#pragma unroll
                for (int j = 0; j < 4; ++j) kv[c][j] = ld16(kp, koff + 16u * j);
            }
            PF_SCHED_FENCE();
        });
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 2; ++g) {                            // coefficients 4 g .. 4 g + 3 of this lane
                uint64_t l4[4], h4[4];
                const uint64_t x4[4] = {r[4 * g], r[4 * g + 1], r[4 * g + 2], r[4 * g + 3]};
                const uint64_t k4[4] = {kv[c][2 * g].x, kv[c][2 * g].y, kv[c][2 * g + 1].x, kv[c][2 * g + 1].y};
#pragma unroll
                for (int e = 0; e < 4; ++e) { l4[e] = lo[c][4 * g + e]; h4[e] = hi[c][4 * g + e]; }
                ks_mac128x4(l4, h4, x4, k4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[c][4 * g + e] = l4[e]; hi[c][4 * g + e] = h4[e]; }
            }
    }
    if constexpr (!INV) {
        // one Barrett reduction per sum; stores staged through the half-wave's area so that every store instruction covers 512
        // contiguous bytes (a lane owns 64 consecutive bytes: stored directly that is a 64-byte lane stride, 0.93 against 3.27 TB/s)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            U64x2 *wr = reinterpret_cast<U64x2 *>(area + 10 * l);
#pragma unroll
            for (int j = 0; j < 4; ++j) wr[j] = U64x2{ar.barrett128(lo[c][2 * j], hi[c][2 * j]), ar.barrett128(lo[c][2 * j + 1], hi[c][2 * j + 1])};
            wsync();
            uint64_t *o = (c ? out1 : out0) + base;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = 2 * (32 * j + l);                          // coefficient pair number 32 j + l
                *reinterpret_cast<U64x2 *>(o + cc) = *reinterpret_cast<const U64x2 *>(area + ksb_p2(cc));
            }
            wsync();
        }
    } else {
        // INV: the first eight stages of the INVERSE transform of both sums run here too -- they act inside the same 256-point
        // blocks, in the reverse order of the layouts (R3: stages 14, 13, 12; R2: 11, 10, 9; R1: 8, 7), with the twiddles of the
        // inverse table at the same 17 entries.  The block leaves in R1 order (16 bytes per lane, 512 per half-wave), reduced to
        // [0, 2q): pass C finishes stages 6..0 over the columns and divides by the special prime.
        T.load(itw, blk, l);
        const int hi2 = l >> 3;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint64_t r[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = ar.barrett128(lo[c][e], hi[c][e]);
            ks_inv_stage<8, 0, 0, false>(r, ar, [&](int g) { return T.s14[g]; });
            ks_inv_stage<8, 1, 1, false>(r, ar, [&](int g) { return T.s13[g]; });
            ks_inv_stage<8, 2, 2, false>(r, ar, [&](int) { return T.s12; });
            {   // exchange 2 backwards: R3 (lane l owns c = 8 l + k) -> R2
                U64x2 *wr = reinterpret_cast<U64x2 *>(area + 10 * l);
#pragma unroll
                for (int j = 0; j < 4; ++j) wr[j] = U64x2{r[2 * j], r[2 * j + 1]};
                wsync();
                const uint64_t *rd = area + ksb_p2(64 * hi2) + 2 * (l & 3) + ((l >> 2) & 1);
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = rd[10 * k];
            }
            ks_inv_stage<8, 0, 3, false>(r, ar, [&](int g) { return T.s11[g]; });
            ks_inv_stage<8, 1, 4, false>(r, ar, [&](int g) { return T.s10[g]; });
            ks_inv_stage<8, 2, 5, false>(r, ar, [&](int) { return T.s9; });
            ks_reduce_all<8>(r, ar);
            wsync();                                                 // exchange 2 read out
            {   // exchange 1 backwards: R2 -> R1
                uint64_t *wr = area + ksb_p1(64 * hi2) + 2 * (l & 3) + ((l >> 2) & 1);
#pragma unroll
                for (int k = 0; k < 8; ++k) wr[8 * k] = r[k];
                wsync();
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const U64x2 v = *reinterpret_cast<const U64x2 *>(area + ksb_p1(128 * (kk >> 1) + 64 * (kk & 1)) + 2 * l);
                    r[2 * kk] = v.x; r[2 * kk + 1] = v.y;
                }
            }
            ks_inv_stage<8, 1, 0, false>(r, ar, [&](int g) { return T.s8[g]; });
            ks_inv_stage<8, 2, 1, false>(r, ar, [&](int) { return T.s7; });
            ks_reduce_all<8>(r, ar);
            uint64_t *o = (c ? out1 : out0) + base + 2 * l;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) *reinterpret_cast<U64x2 *>(o + 128 * (kk >> 1) + 64 * (kk & 1)) = U64x2{r[2 * kk], r[2 * kk + 1]};
            wsync();                                                 // exchange 1 read out before the other component's exchange 2
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same split for the stand-alone transforms and the fused ciphertext x plaintext at N = 32768 (round 4): the single-kernel
// transform hands a CU to one workgroup at this degree and its phases serialise (26 % / 24 % / 18 % of 8 TB/s for forward /
// inverse / ct x pt at config 5).  Here a limb-polynomial goes through small workgroups, IN PLACE in the output buffer:
//   forward   pass A (body_ksA: stages 0..6 on 128 x 64 tiles)  ->  body_nsB<NS_FWD>  (stages 7..14 per 256-point block, canonical form)
//   inverse   body_nsB<NS_INV> (stages 14..7 per block)  ->  body_nsC (stages 6..0 on tiles, N^-1 folded in, canonical form)
//   ct x pt   pass A  ->  body_nsB<NS_MUL> (stages 7..14, product with the NTT-form plaintext, stages 14..7)  ->  body_nsC
// Every pass reads and writes the same index set per workgroup (a tile, or a block), so no workspace is needed; the host runs the
// passes over rounds of polynomials small enough for the intermediate to stay in the 256 MB Infinity Cache (pf_ntt.hip).
// ------------------------------------------------------------------------------------------------------------------
#ifndef PF_NS_MODES
#define PF_NS_MODES
enum { NS_FWD = 0, NS_INV = 1, NS_MUL = 2 };
#endif

// in     the limb-polynomial to read (natural index order): pass A's output for NS_FWD / NS_MUL (= data), NTT form for NS_INV
// data   where the block is written back (the same 256 indices it was read from: in may equal data)
// pt     NS_MUL: this limb's plaintext in NTT form
// chunk  which 2048 coefficients (0..15) this workgroup covers
template <class A, int MODE, class WSync>
PF_HD void body_nsB(const A &ar, const TwU64 *__restrict__ tw, const TwU64 *__restrict__ itw, const uint64_t *in, uint64_t *data, const uint64_t *__restrict__ pt,
                    int chunk, uint64_t *lds, int tid, WSync &&wsync) {
    static_assert(std::is_same<typename A::V, uint64_t>::value, "the split transforms run the 64-bit lazy family");
    const int lane = tid & 63, wv = tid >> 6, h = lane >> 5, l = lane & 31;
    const int blk = chunk * 8 + wv * 2 + h;
    uint64_t *area = lds + wv * KsGeo::B_LDS_WAVE + h * 320;
    const size_t base = (size_t)blk * 256;
    uint64_t r[8];
    KsbTw T;
    if constexpr (MODE != NS_INV) {
        T.load(tw, blk, l);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {                              // R1 order: 16 bytes per lane, 512 contiguous bytes per half-wave
            const U64x2 v = *reinterpret_cast<const U64x2 *>(in + base + 128 * (kk >> 1) + 64 * (kk & 1) + 2 * l);
            r[2 * kk] = v.x; r[2 * kk + 1] = v.y;
        }
        U64x2 kv[4];
        ksb_finish_fwd(r, ar, T, l, area, wsync, [&] {
            if constexpr (MODE == NS_MUL) {
#pragma unroll
                for (int j = 0; j < 4; ++j) kv[j] = *reinterpret_cast<const U64x2 *>(pt + base + 8 * l + 2 * j);
                PF_SCHED_FENCE();
            }
        });
        if constexpr (MODE == NS_FWD) {
            // canonical form; stores staged through the half-wave's area so that every store instruction covers 512 contiguous bytes
            ks_reduce_all<8>(r, ar);
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = ar.canon_small(r[e]);
            U64x2 *wr = reinterpret_cast<U64x2 *>(area + 10 * l);
#pragma unroll
            for (int j = 0; j < 4; ++j) wr[j] = U64x2{r[2 * j], r[2 * j + 1]};
            wsync();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cc = 2 * (32 * j + l);                      // coefficient pair number 32 j + l
                *reinterpret_cast<U64x2 *>(data + base + cc) = *reinterpret_cast<const U64x2 *>(area + ksb_p2(cc));
            }
            wsync();
            return;
        } else {
            // dyadic product with the plaintext on this lane's 8 consecutive NTT-form coefficients: x < 2^62, pt < q < 2^56
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                uint64_t l4[4] = {0, 0, 0, 0}, h4[4] = {0, 0, 0, 0};
                const uint64_t x4[4] = {r[4 * g], r[4 * g + 1], r[4 * g + 2], r[4 * g + 3]};
                const uint64_t k4[4] = {kv[2 * g].x, kv[2 * g].y, kv[2 * g + 1].x, kv[2 * g + 1].y};
                ks_mac128x4(l4, h4, x4, k4);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[4 * g + e] = ar.barrett128(l4[e], h4[e]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                 // R3 order: a lane owns 8 consecutive NTT-form coefficients
            const U64x2 v = *reinterpret_cast<const U64x2 *>(in + base + 8 * l + 2 * j);
            r[2 * j] = v.x; r[2 * j + 1] = v.y;
        }
    }
    // the first eight inverse stages, as in body_ksB<INV>: R3 (14, 13, 12), R2 (11, 10, 9), R1 (8, 7); the block leaves in R1 order, in [0, 2q)
    T.load(itw, blk, l);
    const int hi2 = l >> 3;
    ks_inv_stage<8, 0, 0, false>(r, ar, [&](int g) { return T.s14[g]; });
    ks_inv_stage<8, 1, 1, false>(r, ar, [&](int g) { return T.s13[g]; });
    ks_inv_stage<8, 2, 2, false>(r, ar, [&](int) { return T.s12; });
    {
        U64x2 *wr = reinterpret_cast<U64x2 *>(area + 10 * l);
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[j] = U64x2{r[2 * j], r[2 * j + 1]};
        wsync();
        const uint64_t *rd = area + ksb_p2(64 * hi2) + 2 * (l & 3) + ((l >> 2) & 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = rd[10 * k];
    }
    ks_inv_stage<8, 0, 3, false>(r, ar, [&](int g) { return T.s11[g]; });
    ks_inv_stage<8, 1, 4, false>(r, ar, [&](int g) { return T.s10[g]; });
    ks_inv_stage<8, 2, 5, false>(r, ar, [&](int) { return T.s9; });
    ks_reduce_all<8>(r, ar);
    wsync();
    {
        uint64_t *wr = area + ksb_p1(64 * hi2) + 2 * (l & 3) + ((l >> 2) & 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) wr[8 * k] = r[k];
        wsync();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const U64x2 v = *reinterpret_cast<const U64x2 *>(area + ksb_p1(128 * (kk >> 1) + 64 * (kk & 1)) + 2 * l);
            r[2 * kk] = v.x; r[2 * kk + 1] = v.y;
        }
    }
    ks_inv_stage<8, 1, 0, false>(r, ar, [&](int g) { return T.s8[g]; });
    ks_inv_stage<8, 2, 1, false>(r, ar, [&](int) { return T.s7; });
    ks_reduce_all<8>(r, ar);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) *reinterpret_cast<U64x2 *>(data + base + 128 * (kk >> 1) + 64 * (kk & 1) + 2 * l) = U64x2{r[2 * kk], r[2 * kk + 1]};
    wsync();
}

// ------------------------------------------------------------------------------------------------------------------
// Pass C: inverse stages 6..0 (index bits 8..14) of the sums that pass B left half-way, then SEAL's division by the special
// prime P with rounding, added into the ciphertext -- steps 3 and 4 of switch_key_inplace in one pass over the sums.
// A workgroup of 512 threads owns the tile of columns [64 cb, 64 cb + 64) of ONE (ciphertext, component) and walks limbs:
// first the special prime's (its t = (s_P + floor(P/2)) mod P stays in 16 registers per thread), then the data limbs
// j0 .. j1-1, each: load the tile (pass A's OUTPUT order), three stages in registers, reduce, the exchange of pass A
// backwards, four stages (the last folds N^-1 in, as SEAL does), canonical form, then per coefficient
//     ct[J] += P^-1 * (s_J - (t mod q_J) + (floor(P/2) mod q_J))   (mod q_J).
// Every twiddle is wave-uniform again (scalar operands).
// ------------------------------------------------------------------------------------------------------------------
struct KsLimbC {
    uint64_t q, ratio0, ratio1;                 // modulus, floor(2^128 / q)
    uint64_t half_mod, pinv, pinv_quot;         // floor(P/2) mod q, P^-1 mod q and its Shoup quotient (data limbs)
    const TwU64 *itw;                           // inverse table of this modulus
};

template <class A, class Sync>
PF_HD void ksc_inverse_tile(uint64_t (&r)[16], const A &ar, const TwU64 *__restrict__ itw, const uint64_t *__restrict__ src, int cb,
                            uint64_t *lds, int tid, Sync &&sync) {
    const int lane = tid & 63, w = wave_uniform(tid >> 6);
    // registers = (g, r mod 8) with r div 8 = 2 w + g  (what pass A stored from)
    const uint64_t *s0 = src + (size_t)w * 4096 + cb * 64 + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = s0[(size_t)k * 256];
    const TwU64 *t4 = itw + 16 + 2 * w, *t5 = itw + 32 + 4 * w, *t6 = itw + 64 + 8 * w;
    ks_inv_stage<16, 0, 0, true>(r, ar, [&](int g) { return const_load_tw(t6 + g); });
    ks_inv_stage<16, 1, 1, true>(r, ar, [&](int g) { return const_load_tw(t5 + g); });
    ks_inv_stage<16, 2, 2, true>(r, ar, [&](int g) { return const_load_tw(t4 + g); });
    ks_reduce_all<16>(r, ar);
    sync();                                                          // the previous limb's readers are done with the buffer
    uint64_t *l1 = lds + w * 1024 + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) l1[k * 64] = r[k];
    sync();
    const uint64_t *l0 = lds + w * 64 + lane;                       // registers = r div 8, wave = r mod 8
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = l0[k * 512];
    ks_inv_stage<16, 0, 0, true>(r, ar, [&](int g) { return const_load_tw(itw + 8 + g); });
    ks_inv_stage<16, 1, 1, true>(r, ar, [&](int g) { return const_load_tw(itw + 4 + g); });
    ks_inv_stage<16, 2, 2, true>(r, ar, [&](int g) { return const_load_tw(itw + 2 + g); });
    {   // last layer (index bit 14): N^-1 folded into both halves -- entry 0 is N^-1, entry 1 psi^-bitrev(1) * N^-1
        const TwU64 tn = const_load_tw(itw), t1 = const_load_tw(itw + 1);
#pragma unroll
        for (int bb = 0; bb < 8; bb += 2) {
            uint64_t vs[4];
            TwU64 ts[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                vs[2 * i] = r[bb + i];
                ar.template inv_split<3>(vs[2 * i], r[bb + i + 8], vs[2 * i + 1]);
                ts[2 * i] = tn; ts[2 * i + 1] = t1;
            }
            ar.template mul_tw_n<4, true>(vs, ts);
#pragma unroll
            for (int i = 0; i < 2; ++i) { r[bb + i] = vs[2 * i]; r[bb + i + 8] = vs[2 * i + 1]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = ar.canon_small(r[k]);
}

// acc     half-inverted sums [2][K][N] of this ciphertext (pass B, INV); comp selects the component
// ct      this ciphertext [2][D][N], coefficient form, updated in place
// limb    callable: J -> KsLimbC
template <class A, class LimbFn, class Sync>
PF_HD void body_ksC(LimbFn &&limb, const uint64_t *__restrict__ acc, uint64_t *__restrict__ ct, int comp, int D, int K, int j0, int j1, int cb,
                    uint64_t *lds, int tid, Sync &&sync) {
    constexpr size_t N = KsGeo::N;
    const int lane = tid & 63, w = wave_uniform(tid >> 6);
    uint64_t t[16];
    const KsLimbC lp = limb(K - 1);
    const uint64_t P = lp.q, half = P >> 1;
    {
        const A ar{lp.q, 2 * lp.q, lp.ratio0, lp.ratio1};
        ksc_inverse_tile(t, ar, lp.itw, acc + ((size_t)comp * K + (K - 1)) * N, cb, lds, tid, sync);
#pragma unroll
        for (int k = 0; k < 16; ++k) { const uint64_t v = t[k] + half; t[k] = v >= P ? v - P : v; }
    }
    for (int J = j0; J < j1; ++J) {
        const KsLimbC lm = limb(J);
        const A ar{lm.q, 2 * lm.q, lm.ratio0, lm.ratio1};
        uint64_t s[16];
        ksc_inverse_tile(s, ar, lm.itw, acc + ((size_t)comp * K + J) * N, cb, lds, tid, sync);
        uint64_t *c0 = ct + ((size_t)comp * D + J) * N + (size_t)w * 256 + cb * 64 + lane;       // registers = r div 8, wave = r mod 8
        const uint64_t q = lm.q;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            uint64_t tj = t[k] - mulhi64(t[k], lm.ratio1) * q;                        // t mod q_J (barrett_reduce_64)
            tj = tj >= q ? tj - q : tj;
            uint64_t v = s[k] + (q - tj) + lm.half_mod;                                // < 3q
            v = v >= 2 * q ? v - 2 * q : v;
            v = v >= q ? v - q : v;
            uint64_t x = v * lm.pinv - mulhi64(v, lm.pinv_quot) * q;                   // Shoup product, [0, 2q)
            x = x >= q ? x - q : x;
            x += c0[(size_t)k * 2048];
            c0[(size_t)k * 2048] = x >= q ? x - q : x;
        }
    }
}

// Inverse stages 6..0 of one tile of a half-inverted limb-polynomial, in place: canonical coefficients out (natural order).
template <class A, class Sync>
PF_HD void body_nsC(const A &ar, const TwU64 *__restrict__ itw, uint64_t *data, int cb, uint64_t *lds, int tid, Sync &&sync) {
    const int lane = tid & 63, w = wave_uniform(tid >> 6);
    uint64_t r[16];
    ksc_inverse_tile(r, ar, itw, data, cb, lds, tid, sync);
    uint64_t *c0 = data + (size_t)w * 256 + cb * 64 + lane;            // registers = r div 8, wave = r mod 8
#pragma unroll
    for (int k = 0; k < 16; ++k) c0[(size_t)k * 2048] = r[k];
}

}  // namespace pf
