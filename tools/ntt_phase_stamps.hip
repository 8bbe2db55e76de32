// Diagnostic build of the forward NTT kernel with s_memtime stamps at its phase boundaries (wave 0 of every workgroup;
// each stamp first drains the wave's outstanding memory operations so that a phase's time includes the latency it caused).
// Prints the median cycles per phase over all workgroups.  Not part of the product; numbers are for DESIGN.md.
// Build:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Wno-pass-failed -Iprefhetch_amd/csrc tools/ntt_phase_stamps.hip -o tools/ntt_phase_stamps
// Run:    tools/ntt_phase_stamps [logn=15] [polys=7680]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

__device__ unsigned long long *pf_stamp_buf;
#define PF_STAMP(id)                                                                                        \
    do {                                                                                                    \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                         \
        if (threadIdx.x == 0) pf_stamp_buf[(size_t)blockIdx.x * 64 + (id)] = __builtin_amdgcn_s_memtime();  \
    } while (0)
// inside the exchanges: a running slot counter per workgroup (wave 0, lane 0), slots 16 ..
__device__ unsigned int pf_stamp_x_slot[1 << 16];
#define PF_STAMP_X(step)                                                                                      \
    do {                                                                                                      \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                           \
        if (threadIdx.x == 0) { const unsigned s_ = pf_stamp_x_slot[blockIdx.x & 0xFFFF]++;                   \
            if (s_ < 40) pf_stamp_buf[(size_t)blockIdx.x * 64 + 16 + s_] = __builtin_amdgcn_s_memtime(); }    \
    } while (0)
#include "pf_ntt_kernels.hpp"
#include "tables.hpp"

using namespace pf;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int LOGN>
int run(size_t polys) {
    using A = ArithU64L;
    const uint32_t N = 1u << LOGN;
    const uint64_t q = 0x7FFFFFFFE90001ull;
    LimbTables t; std::string err;
    if (!build_limb_tables(N, q, t, err)) { printf("tables: %s\n", err.c_str()); return 1; }
    std::vector<uint64_t> blob;
    for (auto &e : t.fwd_u) { blob.push_back(e.w); blob.push_back(e.wq); }
    LimbDev ld{}; ld.q = q; ld.two_q = 2 * q; ld.ratio0 = t.ratio0; ld.ratio1 = t.ratio1; ld.fwd_u = 0;
    void *d_tab; LimbDev *d_l; uint64_t *d_x; unsigned long long *d_st;
    CK(hipMalloc(&d_tab, blob.size() * 8)); CK(hipMemcpy(d_tab, blob.data(), blob.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_l, sizeof ld)); CK(hipMemcpy(d_l, &ld, sizeof ld, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_x, polys * N * 8)); CK(hipMemset(d_x, 1, polys * N * 8));
    CK(hipMalloc((void **)&d_st, polys * 64 * 8)); CK(hipMemset(d_st, 0, polys * 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(pf_stamp_buf), &d_st, sizeof d_st));
    NttArgs a{}; a.limbs = d_l; a.tables = d_tab; a.src = d_x; a.dst = d_x; a.L = 1;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_ntt<LOGN, A, false>), dim3((unsigned)polys), dim3(Geo<LOGN>::T), 0, 0, a);
    CK(hipDeviceSynchronize());
    { void *sx; CK(hipGetSymbolAddress(&sx, HIP_SYMBOL(pf_stamp_x_slot))); CK(hipMemset(sx, 0, sizeof(unsigned int) << 16)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_ntt<LOGN, A, false>), dim3((unsigned)polys), dim3(Geo<LOGN>::T), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(polys * 64);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    const char *names[] = {"load (issue + wait)", "pass 0", "exchange 0->1 (+ twiddle fetch)", "pass 1", "exchange 1->2", "pass 2", "canonicalise", "store (issue + drain)"};
    printf("N = %u, %zu polynomials, %d threads x %d coefficients, stamped launch %.3f ms (s_memtime ticks = shader cycles)\n", N, polys, Geo<LOGN>::T, Geo<LOGN>::R, ms);
    double total = 0;
    for (int ph = 0; ph < 8; ++ph) {
        std::vector<double> d;
        for (size_t b = 0; b < polys; ++b) d.push_back((double)(st[b * 64 + ph + 1] - st[b * 64 + ph]));
        std::sort(d.begin(), d.end());
        printf("  %-34s median %8.0f ticks  (p10 %8.0f, p90 %8.0f)\n", names[ph], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
        total += d[d.size() / 2];
    }
    printf("  sum of medians %.0f cycles\n", total);
    const char *xn[] = {"wait at first barrier", "write half", "wait at second barrier", "read half"};
    for (int x = 0; x < 4; ++x)                       // exchange x / 2, round x % 2: five stamps each
        for (int stp = 0; stp < 4; ++stp) {
            std::vector<double> d;
            for (size_t b = 0; b < polys; ++b) d.push_back((double)(st[b * 64 + 16 + x * 5 + stp + 1] - st[b * 64 + 16 + x * 5 + stp]));
            std::sort(d.begin(), d.end());
            printf("    exchange %d round %d: %-24s median %7.0f (p90 %7.0f)\n", x / 2, x % 2, xn[stp], d[d.size() / 2], d[d.size() * 9 / 10]);
        }
    return 0;
}

// fused ct x pt at N = 8192 on the exact-FP64 butterflies (the headline kernel): one limb, `pairs` (ciphertext, limb) pairs
int run_ctpt13(size_t pairs) {
    constexpr int LOGN = 13;
    using A = ArithF64;
    const uint32_t N = 1u << LOGN;
    const uint64_t q = 0xFFFFFFFC001ull;
    LimbTables t; std::string err;
    if (!build_limb_tables(N, q, t, err) || !t.f64_ok) { printf("tables: %s\n", err.c_str()); return 1; }
    std::vector<uint64_t> blob;
    for (auto &e : t.fwd_f) blob.push_back(__builtin_bit_cast(uint64_t, e.w));
    const uint32_t inv_off = (uint32_t)blob.size();
    for (auto &e : t.inv_f) blob.push_back(__builtin_bit_cast(uint64_t, e.w));
    LimbDev ld{}; ld.q = q; ld.two_q = 2 * q; ld.ratio0 = t.ratio0; ld.ratio1 = t.ratio1; ld.qd = (double)q; ld.qinv = 1.0 / (double)q;
    ld.fwd_f = 0; ld.inv_f = inv_off;
    void *d_tab; LimbDev *d_l; uint64_t *d_ct, *d_pt, *d_out; unsigned long long *d_st;
    const size_t polys = pairs * 2;
    CK(hipMalloc(&d_tab, blob.size() * 8)); CK(hipMemcpy(d_tab, blob.data(), blob.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_l, sizeof ld)); CK(hipMemcpy(d_l, &ld, sizeof ld, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_ct, polys * N * 8)); CK(hipMemset(d_ct, 0, polys * N * 8));
    CK(hipMalloc((void **)&d_pt, pairs * N * 8)); CK(hipMemset(d_pt, 0, pairs * N * 8));
    CK(hipMalloc((void **)&d_out, polys * N * 8));
    const size_t grid = (pairs + 7) / 8 * 16;
    CK(hipMalloc((void **)&d_st, grid * 64 * 8)); CK(hipMemset(d_st, 0, grid * 64 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(pf_stamp_buf), &d_st, sizeof d_st));
    NttArgs a{}; a.limbs = d_l; a.tables = d_tab; a.src = d_ct; a.dst = d_out; a.pt = d_pt; a.n_pairs = pairs; a.L = 1;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_ctpt<LOGN, A, 0>), dim3((unsigned)grid), dim3(Geo<LOGN>::T), 0, 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_ctpt<LOGN, A, 0>), dim3((unsigned)grid), dim3(Geo<LOGN>::T), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(grid * 64);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    struct Ph { int a, b; const char *name; };
    const Ph ph[] = {{0, 1, "load ciphertext limb"}, {1, 2, "forward pass 0"}, {2, 3, "exchange 0->1"}, {3, 4, "forward pass 1"}, {4, 5, "exchange 1->2"},
                     {5, 6, "forward pass 2"}, {6, 7, "load plaintext limb"}, {7, 8, "dyadic product (+ twiddle fetch)"}, {8, 9, "inverse pass 2"},
                     {9, 10, "exchange 2->1"}, {10, 11, "inverse pass 1"}, {11, 12, "exchange 1->0"}, {12, 13, "inverse pass 0"}, {13, 14, "canonicalise + store"}};
    printf("k_ctpt<13, ArithF64, 0>: %zu pairs, stamped launch %.3f ms (every stamp drains the wave's memory operations first)\n", pairs, ms);
    double total = 0;
    for (const Ph &p : ph) {
        std::vector<double> d;
        for (size_t b = 0; b < grid; ++b) if (st[b * 64 + 14]) d.push_back((double)(st[b * 64 + p.b] - st[b * 64 + p.a]));
        std::sort(d.begin(), d.end());
        printf("  %-36s median %7.0f cycles (p10 %7.0f, p90 %7.0f)\n", p.name, d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
        total += d[d.size() / 2];
    }
    printf("  sum of medians %.0f cycles per workgroup (3 workgroups per CU run concurrently)\n", total);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && std::string(argv[1]) == "ctpt") return run_ctpt13(argc > 2 ? (size_t)atoll(argv[2]) : 4096);
    const int logn = argc > 1 ? atoi(argv[1]) : 15;
    const size_t polys = argc > 2 ? (size_t)atoll(argv[2]) : 7680;
    if (logn == 15) return run<15>(polys);
    if (logn == 13) return run<13>(polys);
    printf("logn 13 or 15\n");
    return 1;
}
