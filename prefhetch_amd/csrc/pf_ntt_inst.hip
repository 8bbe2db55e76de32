// pf_ntt_inst.hip -- instantiates the NTT / ct x pt kernels for ONE ring degree (-DPF_INST_LOGN=13).
#include "pf_ntt_kernels.hpp"

#ifndef PF_INST_LOGN
#error "compile with -DPF_INST_LOGN=<log2 N>"
#endif
#define PF_CAT_(a, b) a##b
#define PF_CAT(a, b) PF_CAT_(a, b)

// A fused ct x pt launch (exact-FP64 family, coefficient form in and out) of fewer workgroups than this takes 16 instead of 32 coefficients per
// thread: twice the waves per polynomial in a launch that does not fill the device (at most two / less than one round of resident workgroups).
// tools/time_ctpt_small.py, same lease, us per launch, 32 | 16 per thread -- N = 4096 x 2 limbs: batch 64 15.2 | 12.0, 128 23.7 | 18.4,
// 256 (BASELINE config 2) 28.4 | 28.0, 383 35.7 | 35.9; N = 8192 x 4 limbs: batch 32 19.5 | 17.8, 64 30.0 | 29.1, 128 49.3 | 53.9 (four passes and a
// whole-polynomial exchange there: two workgroups per CU).
#ifndef PF_SMALL_LAUNCH_BLOCKS
#define PF_SMALL_LAUNCH_BLOCKS(LOGN) ((LOGN) == 12 ? 1536u : 640u)
#endif
#ifndef PF_SMALL_LOGR
#define PF_SMALL_LOGR(LOGN) ((LOGN) == 12 || (LOGN) == 13 ? 4 : 0)
#endif
namespace pf {
namespace {
[[maybe_unused]] inline int small_mode() { static const int m = getenv("PF_CTPT_SMALL") ? atoi(getenv("PF_CTPT_SMALL")) : 1; return m; }     // A/B measurements

template <int LOGN, class A>
void launch_family(int op, int flags, const NttArgs &a, unsigned nblocks, hipStream_t s) {
    const dim3 grid(nblocks), block(Geo<LOGN>::T);
    if (op == 0) { hipLaunchKernelGGL((k_ntt<LOGN, A, false>), grid, block, 0, s, a); return; }
    if (op == 1) { hipLaunchKernelGGL((k_ntt<LOGN, A, true>), grid, block, 0, s, a); return; }
    if (op == 3) { hipLaunchKernelGGL((k_ks_ntt<LOGN, A>), grid, block, 0, s, a); return; }
    if (op == 4) { hipLaunchKernelGGL((k_rows_ntt<LOGN, A>), grid, block, 0, s, a); return; }
    if (op == 5) {
        if constexpr (Geo<LOGN>::R <= 32) hipLaunchKernelGGL((k_rows_ctpt<LOGN, A>), grid, block, 0, s, a);   // host refuses the larger degrees
        return;
    }
    if constexpr (PF_SMALL_LOGR(LOGN) != 0 && std::is_same<A, ArithF64>::value && default_logr(LOGN) != PF_SMALL_LOGR(LOGN)) {
        if ((flags & 7) == 0 && nblocks < PF_SMALL_LAUNCH_BLOCKS(LOGN) && small_mode() == 1) {
            hipLaunchKernelGGL((k_ctpt<LOGN, A, 0, PF_SMALL_LOGR(LOGN)>), grid, dim3(Geo<LOGN, PF_SMALL_LOGR(LOGN)>::T), 0, s, a);
            return;
        }
    }
    switch (flags & 7) {
#define PF_CASE(F) case F: hipLaunchKernelGGL((k_ctpt<LOGN, A, F>), grid, block, 0, s, a); break;
        PF_CASE(0) PF_CASE(1) PF_CASE(2) PF_CASE(3) PF_CASE(4) PF_CASE(5) PF_CASE(6) PF_CASE(7)
#undef PF_CASE
    }
}

}  // namespace

void PF_CAT(launch_logn_, PF_INST_LOGN)(int arith, int op, int flags, const NttArgs &a, unsigned grid, hipStream_t s) {
    if (arith == 0) launch_family<PF_INST_LOGN, ArithF64>(op, flags, a, grid, s);
    else if (arith == 2) launch_family<PF_INST_LOGN, ArithU64L>(op, flags, a, grid, s);
    else launch_family<PF_INST_LOGN, ArithU64>(op, flags, a, grid, s);
}

}  // namespace pf
