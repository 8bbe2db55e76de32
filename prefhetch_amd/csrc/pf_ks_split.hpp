// pf_ks_split.hpp -- launch interface of the two-pass key switch at N = 32768 (kernels: pf_ks_split.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "pf_ntt_kernels.hpp"

namespace pf {

struct KsSplitArgs {
    const LimbDev *limbs;
    const void *tables;
    const uint64_t *target;      // [nb][D][N] digits of this round's ciphertexts (coefficient form)
    uint64_t *x;                 // [nb][D][nJ][N] pass A's output
    const uint64_t *ksk;         // [D][2][K][N] key, NTT form
    uint64_t *acc;               // [nb][2][K][N] accumulated products (columns J0 .. J0 + nJ - 1 written): NTT form when ct is null, else
                                 // after the first eight inverse stages (pass C finishes them)
    uint32_t D, K, nb, J0, nJ;
    uint64_t *ct;                // [nb][2][D][N] ciphertexts of this round: non-null selects the fused tail (pass B inverts half-way,
                                 // pass C finishes, divides by the special prime and adds into ct); null: NTT-form sums for k_ntt + k_ks_moddown
};

void launch_ksA(const KsSplitArgs &a, hipStream_t s);
void launch_ksB(const KsSplitArgs &a, hipStream_t s);
void launch_ksC(const KsSplitArgs &a, hipStream_t s);       // after every modulus of the round has been through passes A and B

// The split stand-alone transforms and fused ct x pt at N = 32768 (ks_split.hpp: body_nsB / body_nsC), in place on `data`:
// n limb-polynomials, polynomial i belonging to limb i % L; pt (ct x pt only): [n_pt][L][N] NTT-form plaintexts, ciphertext i / (2 L)
// multiplying plaintext (pt_broadcast ? 0 : i / (2 L)).
#ifndef PF_NS_MODES
#define PF_NS_MODES
enum { NS_FWD = 0, NS_INV = 1, NS_MUL = 2 };
#endif
struct NsArgs {
    const LimbDev *limbs;
    const void *tables;
    const uint64_t *src;         // pass A reads here (may equal data)
    uint64_t *data;
    const uint64_t *pt;
    uint32_t L, pt_broadcast;
    size_t n;
};
void launch_nsA(const NsArgs &a, hipStream_t s);
void launch_nsB(const NsArgs &a, int mode, hipStream_t s);  // NS_FWD / NS_INV / NS_MUL
void launch_nsC(const NsArgs &a, hipStream_t s);

}  // namespace pf
