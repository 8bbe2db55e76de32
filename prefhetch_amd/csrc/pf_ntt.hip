// pf_ntt.hip -- RNS ring context, NTT / dyadic / add kernels and the fused ct x pt kernel behind the
// C ABI of include/prefhetch_hip.h.  gfx950 only.  Kernel bodies live in ntt_core.hpp.
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "pf_ks_kernels.hpp"
#include "pf_ks_split.hpp"
#include "pf_ntt_kernels.hpp"
#include "pf_common.hpp"
#include "tables.hpp"

namespace pf {

std::string &last_error_ref() {
    static thread_local std::string s;
    return s;
}

}  // namespace pf

using namespace pf;

struct pf_ctx {
    int device = 0;
    uint32_t N = 0, L = 0, logn = 0;
    bool all_f64 = false, all_lazy64 = false;   // every modulus admits the exact-FP64 / the lazy 64-bit butterflies
    int force_u64 = 0;                           // 0 automatic, 1 64-bit integer butterflies, 2 Harvey butterflies (any q < 2^61)
    int arith() const { return (all_f64 && !force_u64) ? 0 : (all_lazy64 && force_u64 != 2) ? 2 : 1; }
    int num_cus = 256;
    std::vector<LimbTables> tabs;
    LimbDev *d_limbs = nullptr;
    void *d_tables = nullptr;       // all twiddle tables, one allocation
    void *ks_ws = nullptr;          // key-switching workspace (digit NTTs + products): pf_key_switch_reserve, else grown on demand
    size_t ks_ws_bytes = 0;
    // tuning, read ONCE when the context is created (PF_KS_ROUND, PF_KS_JROUND, PF_KS_SPLIT; experiments only):
    size_t ks_round = 0;            // ciphertexts per round of the workspace (0: the default for this ring)
    uint32_t ks_jround = 0;         // two-pass path: key moduli per round (0: all)
    int ks_split = 1;               // N = 32768, lazy 64-bit family: the two-pass digit transforms of ks_split.hpp (PF_KS_SPLIT: 0 off,
                                    // 1 with the inverse transforms and the division by P fused into passes B and C, 2 with k_ntt + k_ks_moddown)
    bool split_ok() const { return logn == 15 && arith() == 2 && ks_split; }
    // the same split for the stand-alone transforms and the fused ct x pt (ks_split.hpp: body_nsB / body_nsC).  PF_NS_SPLIT: bit 0 forward,
    // bit 1 inverse, bit 2 ct x pt; PF_NS_ROUND: limb-polynomials per round.  Measured at config 5 (profiles/r04_ns_*): every pass moves the
    // whole batch through memory again at 4.2-5 TB/s with the vector pipe ~50 % busy -- forward 1.70 -> 1.57 ms (rounds of 960), inverse
    // 1.88 -> 1.81, ct x pt 3.52 -> 3.82 ms: only the forward transform takes the split by default.
    int ns_split = 1;
    size_t ns_round = 960;
    bool ns_ok(int op) const { return logn == 15 && arith() == 2 && ((ns_split >> op) & 1); }
    size_t round_size(size_t B) const {
        const uint32_t K = L, D = K - 1;
        // ciphertexts per round: a multiple of 16 with at most ~4096 digit transforms per launch up to N = 16384 (192 at N = 8192 with
        // 4 + 1 moduli = 0.4 GB of workspace) and ~8192 at N = 32768 (32 at config 5's 15 x 16 digit transforms = 2.2 GB)
        size_t sub = (logn >= 15 ? 8192 : 4096) / ((size_t)D * K) / 16 * 16;
        if (sub < 16) sub = 16;
        if (ks_round) sub = ks_round;
        return sub > B ? B : sub;
    }
    uint32_t jround() const { return (split_ok() && ks_jround && ks_jround < L) ? ks_jround : L; }
    size_t ws_bytes(size_t B) const {
        const size_t sub = round_size(B), K = L, D = K - 1;
        return (sub * D * jround() * N + sub * 2 * K * N) * 8;
    }
};

namespace {

// op: 0 forward, 1 inverse, 2 ctpt
pf_status dispatch_logn(const pf_ctx *c, int arith, int op, int flags, const NttArgs &a, size_t n, hipStream_t s) {
    const unsigned grid = (unsigned)n;          // one workgroup per limb-polynomial
    switch (c->logn) {
        case 10: launch_logn_10(arith, op, flags, a, grid, s); break;
        case 11: launch_logn_11(arith, op, flags, a, grid, s); break;
        case 12: launch_logn_12(arith, op, flags, a, grid, s); break;
        case 13: launch_logn_13(arith, op, flags, a, grid, s); break;
        case 14: launch_logn_14(arith, op, flags, a, grid, s); break;
        case 15: launch_logn_15(arith, op, flags, a, grid, s); break;
        default: return fail(PF_ERR_UNSUPPORTED, "ring degree not built (supported: 1024..32768)");
    }
    PF_HIP(hipGetLastError());
    return PF_OK;
}

// inverse of an odd g modulo 2N (a power of two): Newton's iteration doubles the correct low bits each step
uint32_t inverse_mod_2n(uint32_t g, uint32_t N) {
    uint32_t inv = g;                                         // correct to 3 bits for odd g
    for (int it = 0; it < 5; ++it) inv *= 2u - g * inv;
    return inv & (2u * N - 1);
}

pf_status run_ntt_like(pf_ctx *c, int op, int flags, const NttArgs &a, size_t n, pf_stream stream) {
    if (n == 0) return PF_OK;
    if (n > 0x7fffffffull) return fail(PF_ERR_INVALID_ARG, "too many limb-polynomials for one launch");
    PF_GUARD(c->device);
    const int arith = c->arith();
    return dispatch_logn(c, arith, op, flags, a, n, as_stream(stream));
}

pf_status run_elementwise(pf_ctx *c, int op, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n, pf_stream stream) {
    if (!c || !a || !out || (op != EW_NEG && !b)) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (n == 0) return PF_OK;
    PF_GUARD(c->device);
    const uint32_t chunk_log = c->logn < 11 ? c->logn : 11;
    const size_t blocks = (n << c->logn) >> chunk_log;
    if (blocks > 0x7fffffffull) return fail(PF_ERR_INVALID_ARG, "too many coefficients for one launch");
    EwArgs e{c->d_limbs, a, b ? b : a, out, c->L, c->logn};
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = as_stream(stream);
    switch (op) {
        case EW_MUL: hipLaunchKernelGGL((k_elementwise<EW_MUL>), grid, block, 0, s, e); break;
        case EW_ADD: hipLaunchKernelGGL((k_elementwise<EW_ADD>), grid, block, 0, s, e); break;
        case EW_SUB: hipLaunchKernelGGL((k_elementwise<EW_SUB>), grid, block, 0, s, e); break;
        default: hipLaunchKernelGGL((k_elementwise<EW_NEG>), grid, block, 0, s, e); break;
    }
    PF_HIP(hipGetLastError());
    return PF_OK;
}

}  // namespace

extern "C" {

const char *pf_status_str(pf_status s) {
    switch (s) {
        case PF_OK: return "ok";
        case PF_ERR_INVALID_ARG: return "invalid argument";
        case PF_ERR_UNSUPPORTED: return "unsupported parameters";
        case PF_ERR_HIP: return "HIP runtime error";
        case PF_ERR_NO_DEVICE: return "no usable device";
        case PF_ERR_OOM: return "out of device memory";
        default: return "unknown status";
    }
}

const char *pf_last_error(void) { return last_error_ref().c_str(); }

// what this library was built with: "" for the product build; the Makefile's $(EXTRA) otherwise, prefixed "experiment:" when the
// build carries a switch that changes results or adds debug buffers (pf_common.hpp)
#ifndef PF_BUILD_EXTRA
#define PF_BUILD_EXTRA ""
#endif
const char *pf_build_flags(void) {
#ifdef PF_EXPERIMENT_BUILD
    return "experiment: " PF_BUILD_EXTRA;
#else
    return PF_BUILD_EXTRA;
#endif
}

pf_status pf_device_count(int *count) {
    if (!count) return fail(PF_ERR_INVALID_ARG, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return fail(PF_ERR_NO_DEVICE, "hipGetDeviceCount failed"); }
    *count = n;
    return PF_OK;
}

pf_status pf_malloc(int device, void **dptr, size_t bytes) {
    if (!dptr) return fail(PF_ERR_INVALID_ARG, "null argument");
    PF_GUARD(device);
    PF_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return PF_OK;
}
pf_status pf_free(int device, void *dptr) {
    PF_GUARD(device);
    PF_HIP(hipFree(dptr));
    return PF_OK;
}
pf_status pf_memcpy_h2d(int device, void *dst, const void *src_host, size_t bytes, pf_stream stream) {
    PF_GUARD(device);
    PF_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return PF_OK;
}
pf_status pf_memcpy_d2h(int device, void *dst_host, const void *src, size_t bytes, pf_stream stream) {
    PF_GUARD(device);
    PF_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return PF_OK;
}
pf_status pf_memcpy_d2d(int device, void *dst, const void *src, size_t bytes, pf_stream stream) {
    PF_GUARD(device);
    PF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return PF_OK;
}
pf_status pf_stream_synchronize(int device, pf_stream stream) {
    PF_GUARD(device);
    PF_HIP(hipStreamSynchronize(as_stream(stream)));
    return PF_OK;
}

pf_status pf_ctx_destroy(pf_ctx *c) {
    if (!c) return PF_OK;
    {
        DeviceGuard g(c->device);
        if (c->d_tables) (void)hipFree(c->d_tables);
        if (c->ks_ws) (void)hipFree(c->ks_ws);
        if (c->d_limbs) (void)hipFree(c->d_limbs);
    }
    delete c;
    return PF_OK;
}

pf_status pf_ctx_create(pf_ctx **out, int device, uint32_t N, uint32_t L, const uint64_t *moduli) {
    if (!out || !moduli || L == 0) return fail(PF_ERR_INVALID_ARG, "null argument or L == 0");
    *out = nullptr;
    uint32_t logn = 0;
    while ((1u << logn) < N) ++logn;
    if ((1u << logn) != N || logn < 10 || logn > 15) return fail(PF_ERR_UNSUPPORTED, "N must be a power of two in [1024, 32768]");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(PF_ERR_NO_DEVICE, "no such HIP device");
    PF_GUARD(device);
    pf_ctx *c = new pf_ctx;
    c->device = device; c->N = N; c->L = L; c->logn = logn;
    if (const char *e = getenv("PF_KS_ROUND")) { const long v = atol(e); if (v > 0) c->ks_round = (size_t)v; }
    if (const char *e = getenv("PF_KS_JROUND")) { const long v = atol(e); if (v > 0) c->ks_jround = (uint32_t)v; }
    if (const char *e = getenv("PF_KS_SPLIT")) c->ks_split = atoi(e);
    if (const char *e = getenv("PF_NS_SPLIT")) c->ns_split = atoi(e);
    if (const char *e = getenv("PF_NS_ROUND")) { const long v = atol(e); if (v > 0) c->ns_round = (size_t)v; }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount;
    }
    c->tabs.resize(L);
    c->all_f64 = c->all_lazy64 = true;
    for (uint32_t l = 0; l < L; ++l) {
        std::string err;
        if (!build_limb_tables(N, moduli[l], c->tabs[l], err)) { delete c; return fail(PF_ERR_INVALID_ARG, "modulus " + std::to_string(l) + ": " + err); }
        c->all_f64 = c->all_f64 && c->tabs[l].f64_ok;
        c->all_lazy64 = c->all_lazy64 && u64_lazy_ok(moduli[l], (int)c->logn);
    }
    std::vector<LimbDev> host(L);
    static_assert(sizeof(TwU64) == 16 && sizeof(TwF64) == 8, "table entry sizes");
    std::vector<uint64_t> blob;                    // 8-byte words; u64 tables start on even words (16-byte aligned)
    auto append_u = [&](const std::vector<TwU64> &v) {
        const uint32_t off = (uint32_t)blob.size();
        for (const TwU64 &t : v) { blob.push_back(t.w); blob.push_back(t.wq); }
        return off;
    };
    auto append_f = [&](const std::vector<TwF64> &v) {
        const uint32_t off = (uint32_t)blob.size();
        for (const TwF64 &t : v) blob.push_back(__builtin_bit_cast(uint64_t, t.w));
        return off;
    };
    for (uint32_t l = 0; l < L; ++l) {
        const LimbTables &t = c->tabs[l];
        LimbDev &d = host[l];
        d = LimbDev{};
        d.q = t.q; d.two_q = 2 * t.q; d.ratio0 = t.ratio0; d.ratio1 = t.ratio1;
        d.qd = (double)t.q; d.qinv = 1.0 / (double)t.q;
        d.fwd_u = append_u(t.fwd_u); d.inv_u = append_u(t.inv_u);
        {   // key switching treats the LAST modulus as the special prime P
            const uint64_t P = c->tabs[L - 1].q;
            if (l + 1 < L) {
                d.ks_half_mod = (P >> 1) % t.q;
                d.ks_pinv = h_powmod(P % t.q, t.q - 2, t.q);
                d.ks_pinv_quot = (uint64_t)((((u128_t)d.ks_pinv) << 64) / t.q);
            }
        }
        if (t.f64_ok) { d.fwd_f = append_f(t.fwd_f); d.inv_f = append_f(t.inv_f); }
    }
    if (hipMalloc(&c->d_tables, blob.size() * sizeof(uint64_t)) != hipSuccess ||
        hipMemcpy(c->d_tables, blob.data(), blob.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) {
        pf_ctx_destroy(c);
        return fail(PF_ERR_HIP, "uploading twiddle tables failed");
    }
    if (hipMalloc((void **)&c->d_limbs, sizeof(LimbDev) * L) != hipSuccess ||
        hipMemcpy(c->d_limbs, host.data(), sizeof(LimbDev) * L, hipMemcpyHostToDevice) != hipSuccess) {
        pf_ctx_destroy(c);
        return fail(PF_ERR_HIP, "uploading limb constants failed");
    }
    *out = c;
    return PF_OK;
}

pf_status pf_ctx_info(const pf_ctx *c, uint32_t *N, uint32_t *L, uint64_t *moduli, uint64_t *psi, int32_t *path) {
    if (!c) return fail(PF_ERR_INVALID_ARG, "null context");
    if (N) *N = c->N;
    if (L) *L = c->L;
    for (uint32_t l = 0; l < c->L; ++l) {
        if (moduli) moduli[l] = c->tabs[l].q;
        if (psi) psi[l] = c->tabs[l].psi;
        if (path) path[l] = c->arith();
    }
    return PF_OK;
}

pf_status pf_ctx_force_u64(pf_ctx *c, int on) {
    if (!c) return fail(PF_ERR_INVALID_ARG, "null context");
    if (on < 0 || on > 2) return fail(PF_ERR_INVALID_ARG, "pf_ctx_force_u64: 0, 1 or 2");
    c->force_u64 = on;
    return PF_OK;
}

namespace {
// N = 32768, lazy 64-bit family: the transforms as passes of small workgroups over rounds of polynomials (ks_split.hpp), in place in dst.
// op: 0 forward, 1 inverse, 2 ct x pt (pt non-null).  A round is a multiple of 2 L polynomials so that limbs and ciphertexts stay aligned.
pf_status run_ns_split(pf_ctx *c, int op, const uint64_t *src, uint64_t *dst, const uint64_t *pt, bool pt_broadcast, size_t n, pf_stream stream) {
    if (n == 0) return PF_OK;
    PF_GUARD(c->device);
    hipStream_t s = as_stream(stream);
    const size_t unit = 2 * (size_t)c->L;
    size_t round = c->ns_round / unit * unit;
    if (round < unit) round = unit;
    for (size_t off = 0; off < n; off += round) {
        const size_t cnt = n - off < round ? n - off : round;
        NsArgs a{c->d_limbs, c->d_tables, src + off * c->N, dst + off * c->N, pt ? pt + (pt_broadcast ? 0 : off / unit * c->L * c->N) : nullptr, c->L, pt_broadcast ? 1u : 0u, cnt};
        if (op == 1) { launch_nsB(a, NS_INV, s); launch_nsC(a, s); }
        else { launch_nsA(a, s); launch_nsB(a, op == 0 ? NS_FWD : NS_MUL, s); if (op == 2) launch_nsC(a, s); }
    }
    PF_HIP(hipGetLastError());
    return PF_OK;
}
}  // namespace

pf_status pf_ntt_forward_to(pf_ctx *c, const uint64_t *src, uint64_t *dst, size_t n, pf_stream stream) {
    if (!c || ((!src || !dst) && n)) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (c->ns_ok(0) && n % c->L == 0 && src == dst) return run_ns_split(c, 0,      // (in place only: out of place the split measured 2.00 ms against 1.93)
        src, dst, nullptr, false, n, stream);
    NttArgs a{c->d_limbs, c->d_tables, src, dst, nullptr, 0, c->L, 0, 0};
    return run_ntt_like(c, 0, 0, a, n, stream);
}

pf_status pf_ntt_inverse_to(pf_ctx *c, const uint64_t *src, uint64_t *dst, size_t n, pf_stream stream) {
    if (!c || ((!src || !dst) && n)) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (c->ns_ok(1) && n % c->L == 0) return run_ns_split(c, 1, src, dst, nullptr, false, n, stream);
    NttArgs a{c->d_limbs, c->d_tables, src, dst, nullptr, 0, c->L, 0, 0};
    return run_ntt_like(c, 1, 0, a, n, stream);
}

pf_status pf_ntt_forward(pf_ctx *c, uint64_t *polys, size_t n, pf_stream stream) { return pf_ntt_forward_to(c, polys, polys, n, stream); }
pf_status pf_ntt_inverse(pf_ctx *c, uint64_t *polys, size_t n, pf_stream stream) { return pf_ntt_inverse_to(c, polys, polys, n, stream); }

pf_status pf_dyadic_mul(pf_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n, pf_stream s) { return run_elementwise(c, EW_MUL, a, b, out, n, s); }
pf_status pf_poly_add(pf_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n, pf_stream s) { return run_elementwise(c, EW_ADD, a, b, out, n, s); }
pf_status pf_poly_sub(pf_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n, pf_stream s) { return run_elementwise(c, EW_SUB, a, b, out, n, s); }
pf_status pf_poly_negate(pf_ctx *c, const uint64_t *a, uint64_t *out, size_t n, pf_stream s) { return run_elementwise(c, EW_NEG, a, nullptr, out, n, s); }

pf_status pf_ct_pt_mul(pf_ctx *c, const uint64_t *ct, const uint64_t *pt_ntt, size_t pt_count, uint64_t *out, size_t B, int flags, pf_stream stream) {
    if (!c) return fail(PF_ERR_INVALID_ARG, "null context");
    if (B == 0) return PF_OK;
    if (!ct || !pt_ntt || !out) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (pt_count != 1 && pt_count != B) return fail(PF_ERR_INVALID_ARG, "pt_count must be 1 (broadcast) or B");
    if (flags & ~7) return fail(PF_ERR_INVALID_ARG, "unknown flag bits");
    if (c->ns_ok(2) && flags == 0) return run_ns_split(c, 2, ct, out, pt_ntt, pt_count == 1, B * 2 * (size_t)c->L, stream);
    const size_t pairs = B * (size_t)c->L;
    NttArgs a{c->d_limbs, c->d_tables, ct, out, pt_ntt, pairs, c->L, pt_count == 1 ? 1u : 0u, 0};
    return run_ntt_like(c, 2, flags, a, (pairs + 7) / 8 * 16, stream);     // grid: 8 XCD streams x 2 polynomials per pair
}

pf_status pf_ct_pt_mul_fanout(pf_ctx *c, const uint64_t *ct, const uint64_t *pt_ntt, uint64_t *out, size_t B, uint32_t fanout, int flags,
                              pf_stream stream) {
    if (!c) return fail(PF_ERR_INVALID_ARG, "null context");
    if (B == 0) return PF_OK;
    if (!ct || !pt_ntt || !out) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (fanout == 0) return fail(PF_ERR_INVALID_ARG, "fanout must be at least 1");
    if (flags & ~7) return fail(PF_ERR_INVALID_ARG, "unknown flag bits");
    if (ct == out && fanout > 1) return fail(PF_ERR_INVALID_ARG, "out must not alias ct when ciphertexts fan out");
    const size_t pairs = B * (size_t)c->L;
    NttArgs a{c->d_limbs, c->d_tables, ct, out, pt_ntt, pairs, c->L, 0u, 0u, fanout};
    return run_ntt_like(c, 2, flags, a, (pairs + 7) / 8 * 16, stream);
}

pf_status pf_pack_rows_ntt(pf_ctx *c, const pf_flat *idx, const int64_t *ids, size_t n_polys, uint32_t rows_per_poly, uint64_t *out,
                           pf_stream stream) {
    if (!c || !idx) return fail(PF_ERR_INVALID_ARG, "null context or index");
    if (n_polys == 0) return PF_OK;
    if (!ids || !out) return fail(PF_ERR_INVALID_ARG, "null argument");
    size_t nb = 0; uint32_t d = 0; int dev = 0;
    const float *xb = flat_base_device(idx, &nb, &d, &dev);
    if (dev != c->device) return fail(PF_ERR_INVALID_ARG, "context and index live on different devices");
    if (rows_per_poly == 0 || (size_t)rows_per_poly * d > c->N) return fail(PF_ERR_INVALID_ARG, "rows_per_poly must be in [1, N / d]");
    NttArgs a{c->d_limbs, c->d_tables, nullptr, out, nullptr, 0, c->L, 0u, 0u, 0u, xb, ids, nb, d, rows_per_poly};
    return run_ntt_like(c, 4, 0, a, n_polys * (size_t)c->L, stream);
}

pf_status pf_ct_rows_mul(pf_ctx *c, const uint64_t *ct_ntt, const pf_flat *idx, const int64_t *ids, size_t B, uint32_t rows_per_poly,
                         uint32_t fanout, uint64_t *out, pf_stream stream) {
    if (!c || !idx) return fail(PF_ERR_INVALID_ARG, "null context or index");
    if (B == 0) return PF_OK;
    if (!ct_ntt || !ids || !out) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (fanout == 0) return fail(PF_ERR_INVALID_ARG, "fanout must be at least 1");
    if (ct_ntt == out) return fail(PF_ERR_INVALID_ARG, "out must not alias the ciphertexts");
    if (c->logn > 14) return fail(PF_ERR_UNSUPPORTED, "pf_ct_rows_mul: N <= 16384 (use pf_pack_rows_ntt + pf_ct_pt_mul_fanout above that)");
    size_t nb = 0; uint32_t d = 0; int dev = 0;
    const float *xb = flat_base_device(idx, &nb, &d, &dev);
    if (dev != c->device) return fail(PF_ERR_INVALID_ARG, "context and index live on different devices");
    if (rows_per_poly == 0 || (size_t)rows_per_poly * d > c->N) return fail(PF_ERR_INVALID_ARG, "rows_per_poly must be in [1, N / d]");
    const size_t n_ct = (B + fanout - 1) / fanout, groups = n_ct * (size_t)c->L;
    NttArgs a{c->d_limbs, c->d_tables, ct_ntt, out, nullptr, B, c->L, 0u, 0u, fanout, xb, ids, nb, d, rows_per_poly};
    return run_ntt_like(c, 5, 0, a, (groups + 7) / 8 * 8 * fanout, stream);   // 8 XCD streams x groups x fanout products
}

pf_status pf_pack_rows(pf_ctx *c, const pf_flat *idx, const int64_t *ids, size_t n_polys, uint32_t rows_per_poly, uint64_t *out,
                       pf_stream stream) {
    if (!c || !idx) return fail(PF_ERR_INVALID_ARG, "null context or index");
    if (n_polys == 0) return PF_OK;
    if (!ids || !out) return fail(PF_ERR_INVALID_ARG, "null argument");
    size_t nb = 0; uint32_t d = 0; int dev = 0;
    const float *xb = flat_base_device(idx, &nb, &d, &dev);
    if (dev != c->device) return fail(PF_ERR_INVALID_ARG, "context and index live on different devices");
    if (rows_per_poly == 0 || (size_t)rows_per_poly * d > c->N) return fail(PF_ERR_INVALID_ARG, "rows_per_poly must be in [1, N / d]");
    const size_t blocks = n_polys * c->L * (c->N / 256);
    if (blocks > 0x7fffffffull) return fail(PF_ERR_INVALID_ARG, "too many polynomials for one launch");
    PF_GUARD(c->device);
    PackArgs a{c->d_limbs, xb, ids, out, nb, d, c->L, c->logn, rows_per_poly};
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

pf_status pf_apply_galois(pf_ctx *c, const uint64_t *in, uint64_t *out, size_t n, uint32_t galois_elt, pf_stream stream) {
    if (!c) return fail(PF_ERR_INVALID_ARG, "null context");
    if (n == 0) return PF_OK;
    if (!in || !out) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (in == out) return fail(PF_ERR_INVALID_ARG, "pf_apply_galois is not an in-place operation");
    if (!(galois_elt & 1) || galois_elt >= 2 * c->N) return fail(PF_ERR_INVALID_ARG, "galois_elt must be odd and below 2N");
    const size_t blocks = n * (c->N / 256);
    if (blocks > 0x7fffffffull) return fail(PF_ERR_INVALID_ARG, "too many limb-polynomials for one launch");
    PF_GUARD(c->device);
    GaloisArgs a{c->d_limbs, in, out, c->L, c->logn, inverse_mod_2n(galois_elt, c->N), nullptr};
    hipLaunchKernelGGL(k_apply_galois, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

pf_status pf_poly_mul_monomial(pf_ctx *c, const uint64_t *in, uint64_t *out, size_t n, uint32_t exponent, pf_stream stream) {
    if (!c) return fail(PF_ERR_INVALID_ARG, "null context");
    if (n == 0) return PF_OK;
    if (!in || !out) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (in == out) return fail(PF_ERR_INVALID_ARG, "pf_poly_mul_monomial is not an in-place operation");
    if (exponent >= 2 * c->N) return fail(PF_ERR_INVALID_ARG, "exponent must be below 2N");
    const size_t blocks = n * (c->N / 256);
    if (blocks > 0x7fffffffull) return fail(PF_ERR_INVALID_ARG, "too many limb-polynomials for one launch");
    PF_GUARD(c->device);
    MonoArgs a{c->d_limbs, in, out, c->L, c->logn, exponent};
    hipLaunchKernelGGL(k_mul_monomial, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

pf_status pf_poly_addsub_monomial(pf_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *sum, uint64_t *diff, size_t n, uint32_t exponent,
                                  pf_stream stream) {
    if (!c) return fail(PF_ERR_INVALID_ARG, "null context");
    if (n == 0) return PF_OK;
    if (!a || !b || !sum || !diff) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (diff == a || diff == b || diff == sum) return fail(PF_ERR_INVALID_ARG, "pf_poly_addsub_monomial: diff must not alias the other operands");
    if (exponent >= 2 * c->N) return fail(PF_ERR_INVALID_ARG, "exponent must be below 2N");
    const size_t blocks = n * (c->N / 256);
    if (blocks > 0x7fffffffull) return fail(PF_ERR_INVALID_ARG, "too many limb-polynomials for one launch");
    PF_GUARD(c->device);
    AddSubArgs k{c->d_limbs, a, b, sum, diff, c->L, c->logn, exponent};
    hipLaunchKernelGGL(k_addsub_monomial, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), k);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

pf_status pf_ct_pt_dot(pf_ctx *c, const uint64_t *ct_ntt, size_t n_ct, const uint64_t *pt_ntt, size_t n_pt, size_t chunk, uint64_t *out, pf_stream stream) {
    if (!c) return fail(PF_ERR_INVALID_ARG, "null context");
    if (n_pt == 0) return PF_OK;
    if (!ct_ntt || !pt_ntt || !out) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (n_ct == 0 || chunk == 0 || n_ct % chunk) return fail(PF_ERR_INVALID_ARG, "pf_ct_pt_dot: chunk must divide the ciphertext count");
    if (out == ct_ntt || out == pt_ntt) return fail(PF_ERR_INVALID_ARG, "pf_ct_pt_dot: out must not alias the operands");
    unsigned bits = 0, lc = 0;
    for (uint32_t l = 0; l < c->L; ++l)
        for (unsigned b = 64; b-- > 0;)
            if (c->tabs[l].q >> b) { if (b + 1 > bits) bits = b + 1; break; }
    while ((size_t{1} << lc) < chunk) ++lc;
    if (2 * bits + lc > 128) return fail(PF_ERR_UNSUPPORTED, "pf_ct_pt_dot: chunk * q^2 must stay below 2^128 (lazy 128-bit sums)");
    const size_t groups = (n_pt + chunk - 1) / chunk, blocks = groups * c->L * (c->N / 512);
    if (blocks > 0x7fffffffull) return fail(PF_ERR_INVALID_ARG, "too many blocks for one launch");
    PF_GUARD(c->device);
    DotArgs a{c->d_limbs, ct_ntt, pt_ntt, out, c->L, c->logn, n_ct, n_pt, chunk};
    hipLaunchKernelGGL(k_ct_pt_dot, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

pf_status pf_apply_galois_ct(pf_ctx *c, const uint64_t *ct_in, uint64_t *ct_out, uint64_t *target, size_t B, uint32_t galois_elt, pf_stream stream) {
    if (!c) return fail(PF_ERR_INVALID_ARG, "null context");
    if (B == 0) return PF_OK;
    if (!ct_in || !ct_out || !target) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (ct_in == ct_out || ct_in == target) return fail(PF_ERR_INVALID_ARG, "pf_apply_galois_ct is not an in-place operation");
    if (!(galois_elt & 1) || galois_elt >= 2 * c->N) return fail(PF_ERR_INVALID_ARG, "galois_elt must be odd and below 2N");
    const size_t blocks = B * 2 * c->L * (c->N / 256);
    if (blocks > 0x7fffffffull) return fail(PF_ERR_INVALID_ARG, "too many limb-polynomials for one launch");
    PF_GUARD(c->device);
    GaloisArgs a{c->d_limbs, ct_in, ct_out, c->L, c->logn, inverse_mod_2n(galois_elt, c->N), target};
    hipLaunchKernelGGL(k_apply_galois, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

namespace {
pf_status ks_workspace(pf_ctx *c, size_t B) {
    const size_t need = c->ws_bytes(B);
    if (need > c->ks_ws_bytes) {
        if (c->ks_ws) { PF_HIP(hipFree(c->ks_ws)); c->ks_ws = nullptr; c->ks_ws_bytes = 0; }
        PF_HIP(hipMalloc(&c->ks_ws, need));
        c->ks_ws_bytes = need;
    }
    return PF_OK;
}
}  // namespace

pf_status pf_key_switch_reserve(pf_ctx *c, size_t B) {
    if (!c) return fail(PF_ERR_INVALID_ARG, "null context");
    if (c->L < 2) return fail(PF_ERR_INVALID_ARG, "key switching needs a context with the key moduli: data primes then the special prime");
    if (B == 0) return PF_OK;
    PF_GUARD(c->device);
    return ks_workspace(c, B);
}

pf_status pf_key_switch(pf_ctx *c, const uint64_t *target, const uint64_t *ksk, uint64_t *ct, size_t B, pf_stream stream) {
    if (!c) return fail(PF_ERR_INVALID_ARG, "null context");
    if (B == 0) return PF_OK;
    if (!target || !ksk || !ct) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (c->L < 2) return fail(PF_ERR_INVALID_ARG, "key switching needs a context with the key moduli: data primes then the special prime");
    const uint32_t K = c->L, D = K - 1;
    if (D > 63) return fail(PF_ERR_UNSUPPORTED, "more than 63 digits would overflow the 128-bit lazy accumulator");
    PF_GUARD(c->device);
    hipStream_t s = as_stream(stream);
    const size_t N = c->N;
    const size_t sub = c->round_size(B);
    {   // no allocation on this path once pf_key_switch_reserve(ctx, B) has run (then the call can be captured into a hipGraph)
        const pf_status st = ks_workspace(c, B);
        if (st != PF_OK) return st;
    }
    const bool split = c->split_ok();
    const uint32_t nJ = c->jround();
    const size_t x_words = sub * D * nJ * N;
    uint64_t *x = static_cast<uint64_t *>(c->ks_ws), *acc = x + x_words;
    const int arith = c->arith();
    const uint32_t chunk_log = c->logn < 11 ? c->logn : 11;
    const size_t chunks = N >> chunk_log;
    for (size_t b0 = 0; b0 < B; b0 += sub) {
        const size_t nb = B - b0 < sub ? B - b0 : sub;
        KsArgs k{c->d_limbs, x, ksk, acc, ct + b0 * 2 * D * N, D, K, c->logn, (uint32_t)nb};
        if (split) {
            // 1 + 2. digit transforms in two passes, the second one accumulating the key products (ks_split.hpp)
            const bool fused_tail = c->ks_split == 1;
            KsSplitArgs a{c->d_limbs, c->d_tables, target + b0 * D * N, x, ksk, acc, D, K, (uint32_t)nb, 0, 0, fused_tail ? ct + b0 * 2 * D * N : nullptr};
            for (uint32_t J0 = 0; J0 < K; J0 += nJ) {
                a.J0 = J0; a.nJ = K - J0 < nJ ? K - J0 : nJ;
                launch_ksA(a, s);
                launch_ksB(a, s);
            }
            if (fused_tail) {                                   // 3 + 4. rest of the inverse transforms + division by P, one pass
                launch_ksC(a, s);
                continue;
            }
        } else {
            // 1. digit NTTs: x[b][I][J] = NTT_{m_J}(target[b][I] mod m_J)
            NttArgs a{c->d_limbs, c->d_tables, target + b0 * D * N, x, nullptr, nb * D * K, K, 0, D};
            size_t ks_grid = nb * D * K;
            if (c->logn >= 15 && PF_KS_PERSIST) {               // persistent workgroups, one per CU, a multiple of K of them (k_ks_ntt)
                const size_t per = ((size_t)c->num_cus / K) * K;
                if (per && per < ks_grid) ks_grid = per;
            }
            pf_status st = dispatch_logn(c, arith, 3, 0, a, ks_grid, s);
            if (st != PF_OK) return st;
            // 2. multiply-accumulate with the key
            hipLaunchKernelGGL(k_ks_mac, dim3((unsigned)((K * chunks + 7) / 8 * 8 * nb)), dim3(256), 0, s, k);
        }
        // 3. back to coefficient form, all K limbs of both components
        NttArgs ai{c->d_limbs, c->d_tables, acc, acc, nullptr, 0, K, 0, 0};
        pf_status st = dispatch_logn(c, arith, 1, 0, ai, nb * 2 * K, s);
        if (st != PF_OK) return st;
        // 4. divide by the special prime with rounding and add into the ciphertext
        hipLaunchKernelGGL(k_ks_moddown, dim3((unsigned)(nb * 2 * D * chunks)), dim3(256), 0, s, k);
    }
    PF_HIP(hipGetLastError());
    return PF_OK;
}

}  // extern "C"
