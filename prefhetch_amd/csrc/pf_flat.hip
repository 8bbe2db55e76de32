// pf_flat.hip -- brute-force squared-L2 pre-filter (IndexFlatL2::search semantics), the gathered
// exact distances of Server::preciseSearch and the row gather of Server::preciseVectorPIR /
// retrieve_centroids.  gfx950 only.
//
// Reference call sites this stands in for:
//   faiss::IndexFlatL2 m_Quantizer           /root/reference/include/server/server_lib.h:14, src/server/server_lib.cpp:33
//   sort_nearest_centroids (executed L2 shortlist)  /root/reference/src/client/client_lib.cpp:50-81
//   Server::preciseSearch                     /root/reference/src/server/server_lib.cpp:140-167
//   Server::preciseVectorPIR / retrieve_centroids   server_lib.cpp:169-196 / 101-109
//
// Search pipeline (all launches on the caller's stream, nothing synchronises):
//   distances  k_l2_tile: LDS-tiled fp32 tiles of ||x||^2 + ||y||^2 - 2 x.y (the decomposition faiss uses for
//              nq >= 20, clamped at 0) on the f32 matrix pipe -- v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered
//              fmaf chain, so a distance is plain IEEE fp32 and can be recomputed exactly by scalar code.
//   bootstrap  the first few thousand base rows go through a [nq][chunk] slab and k_select (one workgroup per
//              query keeping a reservoir of packed (distance bits, id) keys in LDS, compacted by a bitonic sort
//              when it fills -- faiss ReservoirTopN does the same on the CPU for k >= 100).  This yields each
//              query's running top-k and its k-th distance tau.
//   streaming  the rest of the base is processed in geometrically growing chunks whose tile kernel FILTERS in its
//              epilogue: only distances <= tau are appended to the query's candidate list (survivors counted with
//              ballots, every row of a wave reserving its range in one atomic round trip; per workgroup through LDS
//              when there are few queries) -- expected k * chunk / rows_seen survivors -- and k_select merges them into the running
//              top-k and tightens tau.  No distance slab is written or re-read.  If a candidate list overflows
//              (adversarial row order), k_select re-derives that query's chunk exactly by recomputing the
//              distances with the same fmaf chain: slower, never wrong.
// Keys order by (distance, id), which fixes the tie order faiss leaves undefined.
#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>
#include "pf_common.hpp"
#include "flat_common.hpp"
#include "flat_prep.hpp"
#include "flat_tile_f32.hpp"
#include "flat_flush16.hpp"
#include "flat_tile16.hpp"
#include "flat_tile8.hpp"
#include "flat_wide16.hpp"
#include "flat_select.hpp"

namespace pf {

// ---- Server::preciseSearch: exact gathered distances ---------------------------------------------
// float dist = 0; dist += std::pow(row[k] - q[k], 2)  ==  dist = (float)((double)dist + (double)diff*(double)diff)
// with diff an fp32 subtraction (server_lib.cpp:151-162).  One lane per (query, candidate); the chain is
// inherently sequential, the row read is 512 contiguous bytes per lane.
__global__ void __launch_bounds__(256) k_l2_gathered(const float *__restrict__ xb, size_t nb, uint32_t d, const float *__restrict__ xq,
                                                      const int64_t *__restrict__ ids, size_t nq, uint32_t c, float *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nq * c) return;
    const size_t qi = t / c;
    const int64_t id = ids[t];
    if (id < 0 || (size_t)id >= nb) { out[t] = INFINITY; return; }
    const float *row = xb + (size_t)id * d, *qv = xq + qi * d;
    float dist = 0.f;
    for (uint32_t k = 0; k < d; ++k) {
        const float diff = row[k] - qv[k];
        dist = (float)((double)dist + (double)diff * (double)diff);
    }
    out[t] = dist;
}

// out[i][:] = xb[ids[i]][:]; one wave per row, 16 B per lane where the row allows
__global__ void __launch_bounds__(256) k_gather_rows(const float *__restrict__ xb, size_t nb, uint32_t d, const int64_t *__restrict__ ids,
                                                      size_t n_ids, float *__restrict__ out) {
    const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_ids) return;
    const int lane = threadIdx.x & 63;
    const int64_t id = ids[r];
    const bool ok = id >= 0 && (size_t)id < nb;
    const float *src = xb + (ok ? (size_t)id : 0) * d;
    float *dst = out + r * d;
    if ((d & 3) == 0) {
        for (uint32_t k = lane * 4; k < d; k += 256) {
            float4 v = ok ? *reinterpret_cast<const float4 *>(src + k) : make_float4(NAN, NAN, NAN, NAN);
            *reinterpret_cast<float4 *>(dst + k) = v;
        }
    } else {
        for (uint32_t k = lane; k < d; k += 64) dst[k] = ok ? src[k] : NAN;
    }
}

}  // namespace pf

using namespace pf;

struct pf_flat {
    int device = 0;
    size_t nb = 0;
    uint32_t d = 0;
    uint32_t dp = 0;              // row length of the operand images: d, or d padded with zeros to whole k-steps (image_row_length)
    float *xb = nullptr, *bn = nullptr;
    uint16_t *xb16 = nullptr;     // bf16 image of the base matrix (d = 64 or 128): rows of d values + AUX16 threshold words, nearest-even
    bool exact16 = false;         // EVERY value of the base passed the on-device exactness check: the image is the matrix itself
    float bn_max = 0.f;           // largest row norm (the margin of the bf16 tiles as a filter over inexact operands)
    bool use16 = true;            // pf_flat_exact16: the caller may switch the 16-bit operand path off
    int8_t *xb8 = nullptr;        // 8-bit data only (every value an integer in [0, 255]; d a multiple of 32 up to 128): rows of d values - 128 + AUX8 threshold bytes
    bool use8 = true;             // pf_flat_operands8: the caller may switch the int8 tiles off (the bf16 tiles then run on the same data)
    int8_t *xb8f = nullptr;       // ... the same bytes in matrix-fragment order (flat_common.hpp: frag8_offset), what the filtered launches stream
    int *c0f = nullptr;           // ... and the columns' integer threshold halves (frag8_c0_index)
    uint16_t *xbw = nullptr;      // rows longer than 256 values: the bf16 image [nb + 128][dpw] the slab tiles stream (flat_wide16.hpp); no exactness claim
    uint32_t dpw = 0;             // ... its row length: d padded with zeros to whole 64-deep slabs
    bool exactw = false;          // ... every value of the base is exactly representable in bf16 (8-bit data): the filter's margin is the accumulation's rounding only
    // workspace (grown outside graph capture)
    void *ws = nullptr;
    size_t ws_bytes = 0;
    size_t wg_slots = 1024;   // workgroups of the tile kernel resident on the device at once (CUs x occupancy)
    size_t num_cus = 256;
    // tuning knobs, read from the environment ONCE when the index is created (experiments; never on the search path)
    size_t b16_min_nq = 1;    // PF_FLAT_B16_MIN_NQ: smallest batch that takes the bf16 tiles
    size_t group_cap = 64;    // PF_FLAT_GROUP_CAP: most column tiles one workgroup of k_l2_tile16 walks
    double growth_div = 0.0;  // PF_FLAT_GROWTH_DIV: survivors per chunk as a fraction of the candidate capacity (0: the defaults)
    bool i8_old = false;      // PF_FLAT_I8_OLD: the round-3 int8 walk (A/B against tile8_walk)
};

namespace {

constexpr size_t STREAMED_WALK_MIN_NQ = 256;      // batches up to this size take the LDS-tiled int8 walk (pf_flat_search_packed)
#ifndef PF_WIDE_GROWTH_DIV
#define PF_WIDE_GROWTH_DIV 5.0
#endif
#ifndef PF_BOOT_ROWS
#define PF_BOOT_ROWS 8192
#endif
constexpr size_t BOOT_ROWS = PF_BOOT_ROWS;  // bootstrap chunk (slab path); at most 8192 (radix_bootstrap keeps the chunk in registers)

// Row length of the 16-bit / 8-bit operand images for rows of d values: every d up to 256 takes the tile path.  A row length the matrix
// instructions take as it is (a multiple of 16) stays; any other is padded with zeros -- to a multiple of 32 up to 128 values (the int8 tiles'
// k-step, so that 8-bit data of any row length runs them), of 16 beyond.  0: no image (d above 256).
uint32_t image_row_length(uint32_t d) {
    if (d > 256) return 0;
    if (d % 16 == 0) return d;
    return d <= 128 ? (d + 31) / 32 * 32 : (d + 15) / 16 * 16;
}

struct WsPlan { size_t boot, slab_ld, cap, off_qn, off_tau, off_cnt, off_scnt, off_state, off_cand, off_slab, off_q16, off_q8, off_qsx, off_qbad, total; };

WsPlan plan_ws(size_t nb, size_t nq, uint32_t k, uint32_t d /* image row length, or d */) {
    WsPlan w{};
    const size_t nb_pad = (nb + 127) / 128 * 128;
    w.boot = BOOT_ROWS < nb_pad ? BOOT_ROWS : (nb_pad ? nb_pad : 128);
    w.slab_ld = w.boot;
    w.cap = SEL_CAP - k;                    // state (<= k keys) + candidates fit one sort
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t o = 0;
    w.off_qn = o; o += up(nq * 4);
    w.off_tau = o; o += up(nq * 4);
    w.off_cnt = o; o += up(nq * 4);
    w.off_scnt = o; o += up(nq * 4);
    w.off_state = o; o += up(nq * (size_t)k * 8);
    w.off_cand = o; o += up(nq * w.cap * 8);
    w.off_slab = o; o += up(nq * w.slab_ld * 4);
    w.off_q16 = o; o += up(nq * (size_t)d * 2);
    w.off_q8 = o; o += up(nq * (size_t)d);
    w.off_qsx = o; o += up(nq * 4);
    w.off_qbad = o; o += up(((nq + 127) / 128) * 4);
    w.total = o;
    return w;
}

pf_status ensure_ws(pf_flat *f, size_t bytes) {
    if (bytes <= f->ws_bytes) return PF_OK;
    if (f->ws) { PF_HIP(hipFree(f->ws)); f->ws = nullptr; f->ws_bytes = 0; }
    PF_HIP(hipMalloc(&f->ws, bytes));
    f->ws_bytes = bytes;
    return PF_OK;
}

}  // namespace

namespace pf {
const float *flat_base_device(const pf_flat *f, size_t *nb, uint32_t *d, int *device) {
    if (!f) return nullptr;
    if (nb) *nb = f->nb;
    if (d) *d = f->d;
    if (device) *device = f->device;
    return f->xb;
}
}  // namespace pf

extern "C" {

pf_status pf_flat_destroy(pf_flat *f) {
    if (!f) return PF_OK;
    {
        DeviceGuard g(f->device);
        if (f->xb) (void)hipFree(f->xb);
        if (f->xb16) (void)hipFree(f->xb16);
        if (f->xb8) (void)hipFree(f->xb8);
        if (f->xb8f) (void)hipFree(f->xb8f);
        if (f->c0f) (void)hipFree(f->c0f);
        if (f->xbw) (void)hipFree(f->xbw);
        if (f->bn) (void)hipFree(f->bn);
        if (f->ws) (void)hipFree(f->ws);
    }
    delete f;
    return PF_OK;
}

pf_status pf_flat_create(pf_flat **out, int device, const float *xb, size_t nb, uint32_t d) {
    if (!out || (!xb && nb) || d == 0) return fail(PF_ERR_INVALID_ARG, "null argument or d == 0");
    *out = nullptr;
    if (nb >= (1ull << 32) - 1) return fail(PF_ERR_UNSUPPORTED, "nb must be below 2^32-1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(PF_ERR_NO_DEVICE, "no such HIP device");
    PF_GUARD(device);
    pf_flat *f = new pf_flat;
    f->device = device; f->nb = nb; f->d = d; f->dp = image_row_length(d);
    const uint32_t dp = f->dp;
    if (const char *v = getenv("PF_FLAT_B16_MIN_NQ")) f->b16_min_nq = (size_t)atoi(v);
    if (const char *v = getenv("PF_FLAT_GROUP_CAP")) { if (atoi(v) > 0) f->group_cap = (size_t)atoi(v); }
    if (const char *v = getenv("PF_FLAT_GROWTH_DIV")) f->growth_div = atof(v);
    if (const char *v = getenv("PF_FLAT_I8_OLD")) f->i8_old = atoi(v) != 0;
    const size_t bytes = (nb ? nb : 1) * (size_t)d * 4;
    hipError_t e = hipMalloc((void **)&f->xb, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&f->bn, (nb ? nb : 1) * 4);
    if (e == hipSuccess && nb) e = hipMemcpy(f->xb, xb, nb * (size_t)d * 4, hipMemcpyDefault);
    if (e == hipSuccess && nb) {
        // row norms; and, where the shape allows the bf16 loop, the 16-bit image with its value-by-value exactness check
        uint32_t *flag = nullptr;
        // every row length that is a multiple of the matrix instruction's k-step up to 256 (k_rows_prep stages whole rows in LDS
        // up to PREP_MAX_D; beyond 128 values the tiles are 128 x 64: Geo16W)
        const bool try16 = dp != 0 && getenv("PF_FLAT_NO_BF16") == nullptr;
        // image rows carry AUX16 threshold words behind their d values; one tile of zero rows pads the end (k_l2_tile16 copies whole tiles)
        const size_t bytes16 = (nb + 128) * (size_t)(dp + AUX16) * 2;
        if (try16 && (hipMalloc((void **)&f->xb16, bytes16) != hipSuccess || hipMemset(f->xb16, 0, bytes16) != hipSuccess ||
                      hipMalloc((void **)&flag, 4) != hipSuccess || hipMemset(flag, 0, 4) != hipSuccess)) {
            (void)hipGetLastError();                                  // no room for the image: the fp32 path needs none
            if (f->xb16) { (void)hipFree(f->xb16); f->xb16 = nullptr; }
            if (flag) { (void)hipFree(flag); flag = nullptr; }
        }
        // 8-bit data (SIFT, the reference's dataset): an int8 image for the integer matrix instruction where the rows are whole 32-deep k-steps
        // (same zero-row padding; rows of d values - 128 + AUX8 bytes).  Kept only if EVERY value is an integer in [0, 255] (flag bit 2).
        const size_t bytes8 = (nb + 128) * (size_t)(dp + AUX8);
        if (f->xb16 && dp % 32 == 0 && dp <= 128 && getenv("PF_FLAT_NO_I8") == nullptr &&
            (hipMalloc((void **)&f->xb8, bytes8) != hipSuccess || hipMemset(f->xb8, 0, bytes8) != hipSuccess)) {
            (void)hipGetLastError();
            if (f->xb8) { (void)hipFree(f->xb8); f->xb8 = nullptr; }
        }
        // the fragment-order image: whole steps of 32 rows, two steps of zero rows behind the end (a walk prefetches one step ahead, clamped to its last)
        const size_t rows8f = (nb + 31) / 32 * 32 + 64, bytes8f = rows8f / 16 * frag8_ksteps(dp) * 1024, bytesc0 = rows8f * sizeof(int);
        if (f->xb8 && (hipMalloc((void **)&f->xb8f, bytes8f) != hipSuccess || hipMemset(f->xb8f, 0, bytes8f) != hipSuccess ||
                       hipMalloc((void **)&f->c0f, bytesc0) != hipSuccess || hipMemset(f->c0f, 0, bytesc0) != hipSuccess)) {
            (void)hipGetLastError();
            if (f->xb8f) { (void)hipFree(f->xb8f); f->xb8f = nullptr; }
            if (f->c0f) { (void)hipFree(f->c0f); f->c0f = nullptr; }
        }
        if (d <= 128) hipLaunchKernelGGL((k_rows_prep<64, 128>), dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, nullptr, f->xb, nb, d, f->bn, f->xb16, dp + AUX16, true, flag, 0u,
                                         f->xb8, dp + (uint32_t)AUX8, f->c0f ? f->xb8f : nullptr, f->c0f, (int *)nullptr, dp);
        else if (d <= PREP_MAX_D) hipLaunchKernelGGL((k_rows_prep<32, PREP_MAX_D>), dim3((unsigned)((nb + 31) / 32)), dim3(64), 0, nullptr, f->xb, nb, d, f->bn, f->xb16, dp + AUX16, true, flag, 0u,
                                                     (int8_t *)nullptr, 0u, (int8_t *)nullptr, (int *)nullptr, (int *)nullptr, dp);
        else {
            hipLaunchKernelGGL(k_row_norms, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, nullptr, f->xb, nb, d, f->bn);
            // rows beyond the register-resident tiles: a bf16 image for the slab tiles (one tile of zero rows behind the end)
            f->dpw = wide_row_length(d);
            const size_t bytesw = (nb + 128) * (size_t)f->dpw * 2, threads = nb * (size_t)(f->dpw / 8);
            if (getenv("PF_FLAT_NO_BF16") == nullptr && threads / 256 < (1ull << 31) &&
                (hipMalloc((void **)&f->xbw, bytesw) != hipSuccess || hipMemset(f->xbw, 0, bytesw) != hipSuccess)) {
                (void)hipGetLastError();                              // no room for the image: the fp32 path needs none
                if (f->xbw) { (void)hipFree(f->xbw); f->xbw = nullptr; }
            }
            if (f->xbw && (hipMalloc((void **)&flag, 4) != hipSuccess || hipMemset(flag, 0, 4) != hipSuccess)) {
                (void)hipGetLastError();
                (void)hipFree(f->xbw); f->xbw = nullptr;
                if (flag) { (void)hipFree(flag); flag = nullptr; }
            }
            if (f->xbw) hipLaunchKernelGGL(k_rows_bf16, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, nullptr, f->xb, nb, d, f->xbw, f->dpw, flag, 0u);
        }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (flag) {
            uint32_t inexact = 1;
            if (e == hipSuccess) e = hipMemcpy(&inexact, flag, 4, hipMemcpyDeviceToHost);
            (void)hipFree(flag);
            f->exact16 = f->xb16 && !(inexact & 1u);       // inexact values: the image stays, as the operand of a conservative filter
            f->exactw = f->xbw && !(inexact & 1u);
            if (f->xb8 && (inexact & 5u)) {                                                 // some value is not an integer in [0, 255]
                (void)hipFree(f->xb8); f->xb8 = nullptr;
                if (f->xb8f) { (void)hipFree(f->xb8f); f->xb8f = nullptr; }
                if (f->c0f) { (void)hipFree(f->c0f); f->c0f = nullptr; }
            }
            if (e == hipSuccess && f->xb16 && !f->exact16) {
                hipLaunchKernelGGL(k_aux_margin, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, nullptr, f->xb16, f->bn, nb, dp, dp + AUX16);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipDeviceSynchronize();
            }
        }
        if (e == hipSuccess && f->xbw) {             // a norm that is not finite rules the slab tiles' filter out
            std::vector<float> norms(nb);
            e = hipMemcpy(norms.data(), f->bn, nb * 4, hipMemcpyDeviceToHost);
            bool finite = true;
            for (float v : norms) { finite = finite && std::isfinite(v); if (v > f->bn_max) f->bn_max = v; }
            if (!finite) { (void)hipFree(f->xbw); f->xbw = nullptr; f->exactw = false; }
        }
        if (e == hipSuccess && f->xb16) {            // largest row norm; a norm that is not finite rules the filter out
            std::vector<float> norms(nb);
            e = hipMemcpy(norms.data(), f->bn, nb * 4, hipMemcpyDeviceToHost);
            bool finite = true;
            for (float v : norms) { finite = finite && std::isfinite(v); if (v > f->bn_max) f->bn_max = v; }
            if (!finite) {
                (void)hipFree(f->xb16); f->xb16 = nullptr; f->exact16 = false;
                if (f->xb8) { (void)hipFree(f->xb8); f->xb8 = nullptr; }
                if (f->xb8f) { (void)hipFree(f->xb8f); f->xb8f = nullptr; }
                if (f->c0f) { (void)hipFree(f->c0f); f->c0f = nullptr; }
            }
        }
    }
    if (e != hipSuccess) { pf_flat_destroy(f); return fail(e == hipErrorOutOfMemory ? PF_ERR_OOM : PF_ERR_HIP, std::string("pf_flat_create: ") + hipGetErrorString(e)); }
    {
        hipDeviceProp_t prop{};
        int occ = 0;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_l2_tile<true, GeoBatch, true>, 256, 0) == hipSuccess && occ > 0)
            f->wg_slots = (size_t)prop.multiProcessorCount * (size_t)occ;
        if (prop.multiProcessorCount > 0) f->num_cus = (size_t)prop.multiProcessorCount;
        (void)hipGetLastError();
    }
    *out = f;
    return PF_OK;
}

pf_status pf_flat_exact16(pf_flat *f, int mode, int *active) {
    if (!f || mode < -1 || mode > 1) return fail(PF_ERR_INVALID_ARG, "pf_flat_exact16: index, and mode -1 (query), 0 (off) or 1 (on where exact)");
    if (mode >= 0) f->use16 = mode == 1;
    if (active) *active = f->xb16 && f->use16 ? (f->exact16 ? 2 : 1) : (f->xbw && f->use16 ? 1 : 0);
    return PF_OK;
}

pf_status pf_flat_operands8(pf_flat *f, int mode, int *active) {
    if (!f || mode < -1 || mode > 1) return fail(PF_ERR_INVALID_ARG, "pf_flat_operands8: index, and mode -1 (query), 0 (off) or 1 (on where the data is 8-bit)");
    if (mode >= 0) f->use8 = mode == 1;
    if (active) *active = f->xb8 && f->use8 && f->xb16 && f->use16 ? 1 : 0;
    return PF_OK;
}

pf_status pf_flat_info(const pf_flat *f, size_t *nb, uint32_t *d) {
    if (!f) return fail(PF_ERR_INVALID_ARG, "null index");
    if (nb) *nb = f->nb;
    if (d) *d = f->d;
    return PF_OK;
}

pf_status pf_flat_reserve(pf_flat *f, size_t nq_max, uint32_t k_max) {
    if (!f || nq_max == 0 || k_max == 0) return fail(PF_ERR_INVALID_ARG, "bad argument");
    PF_GUARD(f->device);
    return ensure_ws(f, plan_ws(f->nb, nq_max, k_max, f->dp ? f->dp : (f->xbw ? f->dpw : f->d)).total);
}

pf_status pf_flat_search(pf_flat *f, const float *xq, size_t nq, uint32_t k, float *D, int64_t *I, pf_stream stream) {
    if (nq && (!D || !I)) return fail(PF_ERR_INVALID_ARG, "null argument");
    return pf_flat_search_packed(f, xq, nq, k, D, I, nullptr, stream);
}

pf_status pf_flat_search_packed(pf_flat *f, const float *xq, size_t nq, uint32_t k, float *D, int64_t *I, uint32_t *packed, pf_stream stream) {
    if (!f) return fail(PF_ERR_INVALID_ARG, "null index");
    if (nq == 0) return PF_OK;
    if (!xq || (!packed && (!D || !I))) return fail(PF_ERR_INVALID_ARG, "null argument");
    if (k == 0 || k > K_MAX) return fail(PF_ERR_UNSUPPORTED, "k must be in [1, 1024]");
    if (nq > (1u << 20)) return fail(PF_ERR_INVALID_ARG, "nq too large for one call (at most 2^20 queries)");
    PF_GUARD(f->device);
    hipStream_t s = as_stream(stream);
    const uint32_t dp = f->dp ? f->dp : (f->xbw ? f->dpw : f->d);     // the images' row length
    const WsPlan w = plan_ws(f->nb, nq, k, dp);
    pf_status st = ensure_ws(f, w.total);
    if (st != PF_OK) return st;
    char *base = static_cast<char *>(f->ws);
    float *qn = reinterpret_cast<float *>(base + w.off_qn);
    float *tau = reinterpret_cast<float *>(base + w.off_tau);
    uint32_t *ccnt = reinterpret_cast<uint32_t *>(base + w.off_cnt);
    uint32_t *scnt = reinterpret_cast<uint32_t *>(base + w.off_scnt);
    uint64_t *state = reinterpret_cast<uint64_t *>(base + w.off_state);
    uint64_t *cand = reinterpret_cast<uint64_t *>(base + w.off_cand);
    float *slab = reinterpret_cast<float *>(base + w.off_slab);
    // Any batch size: for a few queries the 128-row tiles are mostly padding, but the scan is then bound by the bytes of the
    // base it streams, and the bf16 image is half the fp32 matrix (1 query over 1M x 128: 0.22 -> 0.16 ms, 64 queries 0.36 -> 0.23)
    const bool b16 = f->xb16 && f->use16 && nq >= f->b16_min_nq;
    uint16_t *q16 = reinterpret_cast<uint16_t *>(base + w.off_q16);
    uint32_t *qbad = reinterpret_cast<uint32_t *>(base + w.off_qbad);
    const bool wide = f->xbw && f->use16 && nq > 64;                // rows longer than 256 values, batches: the slab tiles (flat_wide16.hpp)
    if (b16 || wide) PF_HIP(hipMemsetAsync(qbad, 0, ((nq + 127) / 128) * 4, s));
    const bool b8 = b16 && f->xb8 && f->use8;                       // int8 tiles for the query tiles that turn out to be 8-bit too (flag bit 2, set by the kernel below)
    int8_t *q8 = reinterpret_cast<int8_t *>(base + w.off_q8);
    int *qsx = reinterpret_cast<int *>(base + w.off_qsx);
    if (f->d <= PREP_MAX_D) hipLaunchKernelGGL((k_rows_prep<4, PREP_MAX_D>), dim3((unsigned)((nq + 3) / 4)), dim3(64), 0, s, xq, nq, f->d, qn, b16 ? q16 : nullptr, dp,
                                               false, b16 ? qbad : nullptr, 128u, b8 ? q8 : nullptr, dp, (int8_t *)nullptr, (int *)nullptr, b8 ? qsx : nullptr, dp);
    else hipLaunchKernelGGL(k_row_norms_wave, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, xq, nq, f->d, qn);
    // rows longer than 256 values, batches: the filtered chunks run the slab tiles over bf16 images (flat_wide16.hpp); the queries' image is built here
    if (wide) hipLaunchKernelGGL(k_rows_bf16, dim3((unsigned)((nq * (size_t)(f->dpw / 8) + 255) / 256)), dim3(256), 0, s, xq, nq, f->d, q16, f->dpw, qbad, 128u);
    TileArgs t{};
    t.xq16 = q16; t.xb16 = f->xb16; t.q_inexact = qbad; t.base_exact = (wide ? f->exactw : f->exact16) ? 1u : 0u; t.bn_max = f->bn_max;
    // The streamed int8 walk wants at least three query tiles: every 1 KiB piece of the base is then asked for by three or more waves at about the
    // same time and all but the first find it in L2.  Below (one or two tiles) a wave's walk is a string of trips to memory, and the LDS-tiled walk
    // (tile16_walk<.., I8>), which shares a column tile among its four waves, is up to 1.7x faster -- 1M x 128, k = 200, streamed | LDS-tiled:
    // 64 queries 0.286 | 0.182 ms, 256 0.315 | 0.201, 384 0.245 | 0.263, 512 0.227 | 0.232, 1024 0.335 | 0.354.
    t.xq8 = b8 ? q8 : nullptr; t.xb8 = b8 ? f->xb8 : nullptr; t.i8_old = (f->i8_old || !f->xb8f || nq <= STREAMED_WALK_MIN_NQ) ? 1u : 0u;
    t.xb8f = f->xb8f; t.c0f = f->c0f; t.qsx8 = qsx;
    t.xq = xq; t.xb = f->xb; t.qn = qn; t.bn = f->bn; t.slab = slab; t.nq = (uint32_t)nq; t.d = f->d; t.slab_ld = (uint32_t)w.slab_ld;
    t.tau = tau; t.cand_cnt = ccnt; t.cand = cand; t.cap = (uint32_t)w.cap;
    SelArgs a{};
    a.slab = slab; a.slab_ld = (uint32_t)w.slab_ld; a.state = state; a.state_cnt = scnt; a.tau = tau; a.cand_cnt = ccnt; a.cand = cand;
    a.cap = (uint32_t)w.cap; a.xq = xq; a.xb = f->xb; a.qn = qn; a.bn = f->bn; a.d = f->d; a.k = k; a.D = D; a.I = I; a.packed = packed; a.q_flags = b16 || wide ? qbad : nullptr; a.bn_max = f->bn_max; a.base_exact = t.base_exact;
    // tile geometry by batch size: 128-row query tiles for batches, 32 / 64-row tiles when a 128-row tile would be
    // mostly padding (the scan of the base is then HBM-bound instead of MFMA-bound)
    const int geo = b16 ? 2 : nq <= 32 ? 0 : nq <= 64 ? 1 : 2;
    const bool wide16 = b16 && dp > 128;                                               // Geo16W
    const size_t TM = geo == 0 ? 32 : geo == 1 ? 64 : 128, TN = b16 ? (size_t)(wide16 ? Geo16W::TN : Geo16::TN) : geo == 2 ? 128 : 256;
    const size_t slots = b16 ? f->num_cus * (size_t)(wide16 ? 2 : B16_WG_PER_CU) : f->wg_slots;       // workgroups of the tile kernel resident at once
    t.n_qtiles = (uint32_t)((nq + TM - 1) / TM);
    auto launch_tile = [&](bool filter, size_t cols) {
        const size_t nct = (cols + TN - 1) / TN;
        if (b16) {
            // Column tiles per workgroup: the launch should take the fewest whole rounds of resident workgroups (2 per CU) that walks of
            // at most group_cap tiles allow, and fill them -- a walk pays a prologue and a flush (about three tiles' worth), and a
            // round that is a quarter full takes as long as a full one.  (Before: two rounds whatever the chunk and walks of at most 8
            // tiles; 16 k columns ran as 2 x 1 tile, 213 k as 3.25 rounds of 8.  Measured over the cap: 8 0.555, 16 0.537, 32 0.520,
            // 64 0.514, 128 0.510 ms per search -- most chunks are then one round of workgroups that walk their whole share.)
            const size_t group_cap = f->group_cap;
            const size_t per_round = slots / t.n_qtiles ? slots / t.n_qtiles : 1;       // column groups of one round
            const size_t rounds = (nct + per_round * group_cap - 1) / (per_round * group_cap);
            size_t group = (nct + per_round * rounds - 1) / (per_round * rounds);
            group = group < 1 ? 1 : group;
            const size_t n_groups = (nct + group - 1) / group;
            const dim3 grid16((unsigned)(((n_groups + 7) / 8) * 8 * t.n_qtiles));
            const uint32_t g32 = (uint32_t)group, n32 = (uint32_t)n_groups;
            switch (dp) {
#define PF_T16P(DD, PAD) do { if (filter && DD % 32 == 0 && DD <= 128 && PF_B16_TN == 128 && b8 && !t.i8_old) \
                                hipLaunchKernelGGL((k_l2_tile16<true, DD, (DD % 32 == 0 && DD <= 128 && PF_B16_TN == 128), PAD>), grid16, dim3(256), 0, s, t, g32, n32); \
                            else if (filter) hipLaunchKernelGGL((k_l2_tile16<true, DD, false, PAD>), grid16, dim3(256), 0, s, t, g32, n32); \
                            else hipLaunchKernelGGL((k_l2_tile16<false, DD, false, PAD>), grid16, dim3(256), 0, s, t, g32, n32); } while (0)
            // (rows padded to the image's length take their own instantiation: flat_flush16.hpp, PADDED)
#define PF_T16(DD) case DD: if (dp != f->d) PF_T16P(DD, true); else PF_T16P(DD, false); break;
#ifdef PF_DEV_ONLY_D128   // development builds (compile time, ISA inspection): rows of 128 values only -- other row lengths are NOT searched (experiment switch)
                PF_T16(128)
#else
                PF_T16(16) PF_T16(32) PF_T16(48) PF_T16(64) PF_T16(80) PF_T16(96) PF_T16(112) PF_T16(128)
                PF_T16(144) PF_T16(160) PF_T16(176) PF_T16(192) PF_T16(208) PF_T16(224) PF_T16(240) PF_T16(256)
#endif
#undef PF_T16
#undef PF_T16P
                default: break;                                       // (pf_flat_create keeps an image for these row lengths only)
            }
            return;
        }
        const dim3 grid((unsigned)(((nct + 7) / 8) * 8 * t.n_qtiles));
        if (wide && filter) {
            // the two accumulations' rounding, dist-level and relative to |x|^2 + |y|^2: the fp32 chain's d roundings of at most half an ulp of a
            // partial sum <= (|x|^2 + |y|^2) / 2, the matrix pipe's d of at most a whole one (should it truncate), doubled by the -2: 3 d 2^-24; taken as d 2^-20
            const float acc_term = (float)f->d * 0x1p-20f;
            hipLaunchKernelGGL(k_l2_wide16, grid, dim3(256), 0, s, t, (const uint16_t *)q16, (const uint16_t *)f->xbw, f->dpw, WIDE_MARGIN + acc_term, acc_term);
            const uint32_t per_q = (uint32_t)((w.cap + 255) / 256);
            hipLaunchKernelGGL(k_wide_fixup, dim3((unsigned)(nq * per_q)), dim3(256), 0, s, t, per_q);
            // the query tiles the filter does not separate for (their word's bit 1): fp32 tiles, the same launch geometry, every other workgroup leaves at once
            TileArgs tf = t;
            tf.only_flagged = 1u;
            if (f->d % TK == 0) hipLaunchKernelGGL((k_l2_tile<true, GeoBatch, true, false>), grid, dim3(256), 0, s, tf);
            else hipLaunchKernelGGL((k_l2_tile<true, GeoBatch, false, false>), grid, dim3(256), 0, s, tf);
            return;
        }
        const bool fast = f->d % TK == 0;
#define PF_TILE(FILTER, GEO, AGG) do { if (fast) hipLaunchKernelGGL((k_l2_tile<FILTER, GEO, true, AGG>), grid, dim3(256), 0, s, t); \
                                       else hipLaunchKernelGGL((k_l2_tile<FILTER, GEO, false, AGG>), grid, dim3(256), 0, s, t); } while (0)
        // per-workgroup aggregation of survivors pays once several queries share the hot counters (measured: nq = 1 0.27 ->
        // 0.25 ms without it, nq = 32 0.30 -> 0.33 ms without it)
        const bool agg = nq > 4;
        switch (geo * 2 + (filter ? 1 : 0)) {
            case 0: PF_TILE(false, GeoSmall32, false); break;
            case 1: if (agg) PF_TILE(true, GeoSmall32, true); else PF_TILE(true, GeoSmall32, false); break;
            case 2: PF_TILE(false, GeoSmall64, false); break;
            case 3: PF_TILE(true, GeoSmall64, true); break;
            case 4: PF_TILE(false, GeoBatch, false); break;
            default: PF_TILE(true, GeoBatch, false); break;
        }
#undef PF_TILE
    };
    auto launch_select = [&]() {
        if (nq <= 256) hipLaunchKernelGGL(k_select<1024>, dim3((unsigned)nq), dim3(1024), 0, s, a);
        else if (a.mode == 1 && !a.first && !a.last) hipLaunchKernelGGL(k_merge4, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, s, a, (uint32_t)nq);   // a wave per query
        else hipLaunchKernelGGL(k_select<256>, dim3((unsigned)nq), dim3(256), 0, s, a);
    };
    // bootstrap chunk through the slab
    const size_t boot = f->nb < w.boot ? f->nb : w.boot;
    t.nb_first = 0; t.nb_count = boot;
    if (boot) launch_tile(false, boot);
    a.nb_first = 0; a.nb_count = boot; a.mode = 0; a.first = 1; a.last = boot == f->nb;
    launch_select();
    // streaming chunks: sized so that the expected survivors per query, k * chunk / rows_seen, stay at a fraction 1/div of
    // the candidate capacity
    const double growth_div = f->growth_div;
    size_t pos = boot;
    // Batches: rows_seen may grow by at most g_max = 1 + cap / (div * k) per chunk.  Taking g_max every time ends in a short
    // last chunk that still costs a launch and a merge; instead the number of chunks n is the smallest that g_max allows and
    // all of them grow by the same ratio (nb / boot)^(1/n) -- 1M rows after 8192: four chunks of ratio 3.3 at div = 3 carry
    // the survivors per chunk that five greedy ones at div = 4 carried.
    double ratio = 0.0;
    int n_chunks = 0, chunk_no = 0;
    if (geo == 2 && f->nb > boot) {
        // (the streamed int8 walk pays more per launch and less per survivor than the LDS-tiled walks: 2.3 makes 1M rows three chunks of ratio 5 -- 0.354 -> 0.345 ms)
        // (the slab tiles of rows beyond 256 values pay for every survivor with a row of the fp32 matrix re-read: more, smaller steps carry fewer
        // survivors in all -- n (r - 1) falls towards ln(nb / boot) -- and a launch is cheap next to their tiles)
        const double div = growth_div > 0.0 ? growth_div : (b8 && !t.i8_old ? 2.3 : wide ? PF_WIDE_GROWTH_DIV : 3.0), g_max = 1.0 + (double)w.cap / (div * (double)k), span = (double)f->nb / (double)boot;
        n_chunks = (int)ceil(log(span) / log(g_max) - 1e-9);
        if (n_chunks < 1) n_chunks = 1;
        ratio = pow(span, 1.0 / n_chunks);
    }
    while (pos < f->nb) {
        size_t chunk;
        if (geo == 2) {
            ++chunk_no;
            const double end = (double)boot * pow(ratio, chunk_no);
            chunk = chunk_no >= n_chunks || end >= (double)f->nb ? f->nb - pos : ((size_t)end - pos) / 1024 * 1024;
            if (chunk < 4096) chunk = 4096;
            // whole rounds of resident workgroups: the short early chunks take as long as their rounds, however full the last one is
            const size_t round_cols = (slots / t.n_qtiles ? slots / t.n_qtiles : 1) * TN;
            if (chunk > round_cols && chunk < f->nb - pos) chunk = (chunk + round_cols / 2) / round_cols * round_cols;
            if (chunk > f->nb - pos || f->nb - pos - chunk < 4096) chunk = f->nb - pos;
        } else {
            const double div = growth_div > 0.0 ? growth_div : 4.0;
            chunk = (size_t)((double)pos * (double)w.cap / (div * (double)k));
            chunk = chunk / 256 * 256;
            if (chunk < 4096) chunk = 4096;
            // whole rounds of resident workgroups: a chunk that fills the device 1.2 times takes as long as one that fills it twice
            const size_t round_cols = (f->wg_slots / t.n_qtiles ? f->wg_slots / t.n_qtiles : 1) * TN;
            if (chunk > round_cols) chunk = chunk / round_cols * round_cols;
            if (chunk > f->nb - pos) chunk = f->nb - pos;
            // a short tail is not worth a launch and a merge of its own (the 4x margin on the candidate capacity absorbs it)
            if (f->nb - pos - chunk < chunk / 2) chunk = f->nb - pos;
        }
        t.nb_first = pos; t.nb_count = chunk;
        launch_tile(true, chunk);
        a.nb_first = pos; a.nb_count = chunk; a.mode = 1; a.first = 0; a.last = pos + chunk == f->nb;
        launch_select();
        pos += chunk;
    }
    PF_HIP(hipGetLastError());
    return PF_OK;
}

pf_status pf_l2_gathered(pf_flat *f, const float *xq, const int64_t *ids, size_t nq, uint32_t c, float *D, pf_stream stream) {
    if (!f) return fail(PF_ERR_INVALID_ARG, "null index");
    if (nq == 0 || c == 0) return PF_OK;
    if (!xq || !ids || !D) return fail(PF_ERR_INVALID_ARG, "null argument");
    PF_GUARD(f->device);
    const size_t total = nq * (size_t)c;
    hipLaunchKernelGGL(k_l2_gathered, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), f->xb, f->nb, f->d, xq, ids, nq, c, D);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

pf_status pf_gather_rows(pf_flat *f, const int64_t *ids, size_t n_ids, float *out, pf_stream stream) {
    if (!f) return fail(PF_ERR_INVALID_ARG, "null index");
    if (n_ids == 0) return PF_OK;
    if (!ids || !out) return fail(PF_ERR_INVALID_ARG, "null argument");
    PF_GUARD(f->device);
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n_ids + 3) / 4)), dim3(256), 0, as_stream(stream), f->xb, f->nb, f->d, ids, n_ids, out);
    PF_HIP(hipGetLastError());
    return PF_OK;
}

}  // extern "C"

#ifdef PF_FLAT_STAMPS
extern "C" int pf_flat_debug_stamps(unsigned long long *out, size_t n) {
    const size_t have = sizeof(pf::pf_flat_stamp_buf) / 8;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::pf_flat_stamp_buf), (n < have ? n : have) * 8);
}
extern "C" int pf_flat_debug_flush_stamps(unsigned long long *out, size_t n) {
    const size_t have = sizeof(pf::pf_flat_flush_stamp_buf) / 8;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pf::pf_flat_flush_stamp_buf), (n < have ? n : have) * 8);
}
#endif
