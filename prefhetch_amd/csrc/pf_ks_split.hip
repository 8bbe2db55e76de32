// pf_ks_split.hip -- kernels of the two-pass key switch at N = 32768 (bodies and design: ks_split.hpp).  gfx950 only.
#include <hip/hip_runtime.h>
#include "ks_split.hpp"
#include "pf_ks_split.hpp"

namespace pf {

using AL = ArithU64L;

struct WaveSync {
    __device__ __forceinline__ void operator()() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

// Pass A.  Block id = ((b * D + I) * nJ + Jl) * 4 + column tile.  The 4 tiles and the nJ moduli of a digit are neighbours in
// the grid: the digit polynomial (256 KiB) is fetched from memory once and re-read from the caches.
__global__ void __launch_bounds__(KsGeo::A_T, 4) k_ksA(KsSplitArgs p) {
    __shared__ uint64_t lds[KsGeo::A_LDS];
    const uint32_t cb = blockIdx.x & 3;
    const size_t t = blockIdx.x >> 2;
    const uint32_t Jl = (uint32_t)(t % p.nJ);
    const size_t digit = t / p.nJ;                                 // b * D + I
    const LimbDev &lm = p.limbs[p.J0 + Jl];
    const AL ar = ArithOf<AL>::make(lm);
    body_ksA<AL>(ar, ArithOf<AL>::fwd(p.tables, lm), p.target + digit * KsGeo::N, p.x + t * KsGeo::N, (int)cb, lds, (int)threadIdx.x, WgSync{});
}

// Pass B.  A (modulus, chunk) unit shares its key slice (D x 2 x 2048 coefficients = 480 KiB at config 5) among the nb
// ciphertexts of the round: XCD x (blocks x, x + 8, ...) takes the units u = x (mod 8) and runs a unit's ciphertexts back to
// back, so the slice comes from memory once and from that XCD's L2 after.
template <bool INV>
__global__ void __launch_bounds__(KsGeo::B_T, 2) k_ksB(KsSplitArgs p) {
    __shared__ __attribute__((aligned(16))) uint64_t lds[KsGeo::B_LDS];
    const uint32_t xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const uint32_t unit = (seq / p.nb) * 8 + xcd;                  // Jl * 16 + chunk
    if (unit >= p.nJ * KsGeo::B_CHUNKS) return;
    const size_t b = seq % p.nb;
    const uint32_t chunk = unit % KsGeo::B_CHUNKS, Jl = unit / KsGeo::B_CHUNKS, J = p.J0 + Jl;
    const LimbDev &lm = p.limbs[J];
    const AL ar = ArithOf<AL>::make(lm);
    const size_t N = KsGeo::N;
    body_ksB<AL, INV>(ar, ArithOf<AL>::fwd(p.tables, lm), ArithOf<AL>::inv(p.tables, lm), p.x + ((b * p.D) * p.nJ + Jl) * N, (size_t)p.nJ * N, p.ksk + (size_t)J * N, (size_t)p.K * N,
                 p.acc + ((b * 2 + 0) * p.K + J) * N, p.acc + ((b * 2 + 1) * p.K + J) * N, (int)p.D, (int)chunk, lds, (int)threadIdx.x, WaveSync{});
}

// Pass C.  Block id = ((b * 2 + component) * groups + limb group) * 4 + column tile: a workgroup inverts the special prime's tile, then
// the data limbs of its group (`per` of them), and folds them into the ciphertext.
#ifndef PF_KSC_WAVES
#define PF_KSC_WAVES 4
#endif
__global__ void __launch_bounds__(KsGeo::A_T, PF_KSC_WAVES) k_ksC(KsSplitArgs p, uint32_t groups, uint32_t per) {
    __shared__ uint64_t lds[KsGeo::A_LDS];
    const uint32_t cb = blockIdx.x & 3;
    const uint32_t t = blockIdx.x >> 2;
    const uint32_t grp = t % groups;
    const size_t bc = t / groups;                                  // b * 2 + component
    const size_t b = bc >> 1;
    const int comp = (int)(bc & 1);
    const uint32_t j0 = grp * per, j1 = j0 + per < p.D ? j0 + per : p.D;
    auto limb = [&](int J) {
        const LimbDev &lm = p.limbs[J];
        return KsLimbC{lm.q, lm.ratio0, lm.ratio1, lm.ks_half_mod, lm.ks_pinv, lm.ks_pinv_quot, ArithOf<AL>::inv(p.tables, lm)};
    };
    const size_t N = KsGeo::N;
    body_ksC<AL>(limb, p.acc + b * 2 * p.K * N, p.ct + b * 2 * p.D * N, comp, (int)p.D, (int)p.K, (int)j0, (int)j1, (int)cb, lds, (int)threadIdx.x, WgSync{});
}

// ---- the split stand-alone transforms / ct x pt (ks_split.hpp: body_nsB, body_nsC) ------------------------------------------------
// Block id = polynomial * 4 + column tile (passes A and C) or polynomial * 16 + chunk (pass B): the pieces of a polynomial are
// neighbours in the grid.
__global__ void __launch_bounds__(KsGeo::A_T, 4) k_nsA(NsArgs p) {
    __shared__ uint64_t lds[KsGeo::A_LDS];
    const uint32_t cb = blockIdx.x & 3;
    const size_t poly = blockIdx.x >> 2;
    const LimbDev &lm = p.limbs[poly % p.L];
    const AL ar = ArithOf<AL>::make(lm);
    body_ksA<AL>(ar, ArithOf<AL>::fwd(p.tables, lm), p.src + poly * KsGeo::N, p.data + poly * KsGeo::N, (int)cb, lds, (int)threadIdx.x, WgSync{});
}

template <int MODE>
__global__ void __launch_bounds__(KsGeo::B_T, 2) k_nsB(NsArgs p) {
    __shared__ __attribute__((aligned(16))) uint64_t lds[KsGeo::B_LDS];
    const uint32_t chunk = blockIdx.x & 15;
    const size_t poly = blockIdx.x >> 4;
    const uint32_t limb = (uint32_t)(poly % p.L);
    const LimbDev &lm = p.limbs[limb];
    const AL ar = ArithOf<AL>::make(lm);
    const uint64_t *pt = nullptr;
    if constexpr (MODE == NS_MUL) pt = p.pt + ((p.pt_broadcast ? 0 : poly / (2 * (size_t)p.L)) * p.L + limb) * KsGeo::N;
    body_nsB<AL, MODE>(ar, ArithOf<AL>::fwd(p.tables, lm), ArithOf<AL>::inv(p.tables, lm), (MODE == NS_INV ? p.src : p.data) + poly * KsGeo::N, p.data + poly * KsGeo::N, pt,
                       (int)chunk, lds, (int)threadIdx.x, WaveSync{});
}

__global__ void __launch_bounds__(KsGeo::A_T, 4) k_nsC(NsArgs p) {
    __shared__ uint64_t lds[KsGeo::A_LDS];
    const uint32_t cb = blockIdx.x & 3;
    const size_t poly = blockIdx.x >> 2;
    const LimbDev &lm = p.limbs[poly % p.L];
    const AL ar = ArithOf<AL>::make(lm);
    body_nsC<AL>(ar, ArithOf<AL>::inv(p.tables, lm), p.data + poly * KsGeo::N, (int)cb, lds, (int)threadIdx.x, WgSync{});
}

void launch_nsA(const NsArgs &a, hipStream_t s) { hipLaunchKernelGGL(k_nsA, dim3((unsigned)(a.n * KsGeo::A_TILES)), dim3(KsGeo::A_T), 0, s, a); }
void launch_nsB(const NsArgs &a, int mode, hipStream_t s) {
    const dim3 grid((unsigned)(a.n * KsGeo::B_CHUNKS)), block(KsGeo::B_T);
    if (mode == NS_FWD) hipLaunchKernelGGL(k_nsB<NS_FWD>, grid, block, 0, s, a);
    else if (mode == NS_INV) hipLaunchKernelGGL(k_nsB<NS_INV>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(k_nsB<NS_MUL>, grid, block, 0, s, a);
}
void launch_nsC(const NsArgs &a, hipStream_t s) { hipLaunchKernelGGL(k_nsC, dim3((unsigned)(a.n * KsGeo::A_TILES)), dim3(KsGeo::A_T), 0, s, a); }

void launch_ksA(const KsSplitArgs &a, hipStream_t s) {
    const size_t grid = (size_t)a.nb * a.D * a.nJ * KsGeo::A_TILES;
    hipLaunchKernelGGL(k_ksA, dim3((unsigned)grid), dim3(KsGeo::A_T), 0, s, a);
}

void launch_ksB(const KsSplitArgs &a, hipStream_t s) {
    const size_t units = ((size_t)a.nJ * KsGeo::B_CHUNKS + 7) / 8 * 8;
    if (a.ct) hipLaunchKernelGGL(k_ksB<true>, dim3((unsigned)(units * a.nb)), dim3(KsGeo::B_T), 0, s, a);
    else hipLaunchKernelGGL(k_ksB<false>, dim3((unsigned)(units * a.nb)), dim3(KsGeo::B_T), 0, s, a);
}

void launch_ksC(const KsSplitArgs &a, hipStream_t s) {
    // limb groups (a group inverts the special prime again: 2 / 4 / 8 groups measured 13.64 / 13.83 / 13.85 ms per 256 at config 5)
#ifndef PF_KSC_GROUPS
#define PF_KSC_GROUPS 2
#endif
    uint32_t groups = PF_KSC_GROUPS;
    if (groups > a.D) groups = a.D;
    const uint32_t per = (a.D + groups - 1) / groups;
    groups = (a.D + per - 1) / per;
    hipLaunchKernelGGL(k_ksC, dim3((unsigned)((size_t)a.nb * 2 * groups * KsGeo::A_TILES)), dim3(KsGeo::A_T), 0, s, a, groups, per);
}

}  // namespace pf
