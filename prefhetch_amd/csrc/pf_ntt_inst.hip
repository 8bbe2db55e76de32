// pf_ntt_inst.hip -- instantiates the NTT / ct x pt kernels for ONE ring degree (-DPF_INST_LOGN=13).
#include "pf_ntt_kernels.hpp"

#ifndef PF_INST_LOGN
#error "compile with -DPF_INST_LOGN=<log2 N>"
#endif
#define PF_CAT_(a, b) a##b
#define PF_CAT(a, b) PF_CAT_(a, b)

namespace pf {
namespace {

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
