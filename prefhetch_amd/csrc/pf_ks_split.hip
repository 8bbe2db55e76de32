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
