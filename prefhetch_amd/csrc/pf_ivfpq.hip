// pf_ivfpq.hip -- IVF-PQ coarse stage: the index behind Server::coarseSearch
// (/root/reference/src/server/server_lib.cpp:111-138), i.e. faiss::IndexIVFPQ(quantizer, d=128, nlist=256, M=32, nbits=8)
// (server_lib.cpp:33-36) and the fork-only IndexIVFPQ::search_encrypted (server_lib.cpp:126-130; PreFHEtch-faiss @
// 49c5b57c, source absent).  Semantics restated from the call site and its consumer (client_lib.cpp:122-156) plus
// faiss's published IVFADC design [Jegou et al. 2011; faiss IndexIVFPQ, by_residual = true]:
//   * add:     vector x -> list c = nearest coarse centroid (squared L2, tie -> smaller id); code[m] = nearest of the
//              256 sub-centroids of sub-quantizer m to the residual (x - centroid_c) restricted to dims [m*dsub, (m+1)*dsub).
//   * search_encrypted(n, x, list ids [n][nprobe]) -> for every query, for every GIVEN list in the given order, for
//              every stored vector in insertion order: the asymmetric distance
//                  sum_m || (x - centroid_list)_m - subcentroid[m][code[m]] ||^2
//              and its id; results of a query are concatenated, list_sizes[q] = number of results of query q.
// Arithmetic contract (shared with oracle/pf_oracle.c): residual r = x - c in fp32; table entry
// T[m][j] = sum_{t<dsub} (r_t - s_t)^2 as fp32 mul then add in index order (no FMA contraction); distance = fp32 sum
// of T[m][code[m]] for m = 0..M-1 in order, starting from 0.  gfx950 only.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string>
#include <vector>
#include "pf_common.hpp"

namespace pf {

constexpr uint32_t KSUB = 256;           // 8-bit sub-quantizers, as the reference configures (SUB_QUANTIZER_SIZE = 8)

struct ScanArgs {
    const float *xq;          // [nq][d]
    const float *centroids;   // [nlist][d]
    const float *codebooks;   // [M][KSUB][dsub]
    const uint8_t *codes;     // [ntotal][M], list-contiguous
    const int64_t *ids;       // [ntotal]
    const uint64_t *list_off; // [nlist+1] offsets into codes/ids
    const int64_t *probe;     // [nq][nprobe] list ids (device copy)
    const uint64_t *out_off;  // [nq*nprobe] output offset of every (query, probe)
    float *D; int64_t *I;
    uint32_t d, M, dsub, nprobe, nlist;
};

// One workgroup per (query, probed list): build the M x 256 fp32 look-up table of the query's residual in LDS,
// then every thread scans codes of the list, 16 B (16 sub-codes) per load.
// [r2] Both loops keep several independent loads in flight: thread t owns sub-centroid t of every sub-quantizer, so the
// residual is workgroup-uniform (scalar loads) and the table is built eight sub-quantizers at a time with their eight
// sub-centroid loads issued first (dsub = 4: one 16-byte load each); the scan takes two codes per iteration.  Before, every
// table entry and every code waited for its own load: a workgroup lived ~55 us for ~1100 instructions per thread.
__global__ void __launch_bounds__(256) k_ivfpq_scan(ScanArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lut[];          // [M][KSUB]
    const size_t qp = blockIdx.x;
    const size_t q = qp / p.nprobe;
    const int64_t list = p.probe[qp];
    if (list < 0 || (uint64_t)list >= p.nlist) return;                     // faiss convention: -1 = no list
    const float *x = p.xq + q * p.d, *c = p.centroids + (size_t)list * p.d;
    const uint32_t tid = threadIdx.x;
    if (p.dsub == 4 && (p.M & 7) == 0) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x), *c4 = reinterpret_cast<const float4 *>(c);
        const float4 *book = reinterpret_cast<const float4 *>(p.codebooks);
        for (uint32_t m0 = 0; m0 < p.M; m0 += 8) {
            float4 s[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = book[(size_t)(m0 + u) * KSUB + tid];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 xv = x4[m0 + u], cv = c4[m0 + u];              // workgroup-uniform
                const float r0 = xv.x - cv.x, r1 = xv.y - cv.y, r2 = xv.z - cv.z, r3 = xv.w - cv.w;
                const float d0 = r0 - s[u].x, d1 = r1 - s[u].y, d2 = r2 - s[u].z, d3 = r3 - s[u].w;
                float acc = 0.f;
                acc = acc + d0 * d0; acc = acc + d1 * d1; acc = acc + d2 * d2; acc = acc + d3 * d3;   // contraction is off: mul, then add
                lut[(m0 + u) * KSUB + tid] = acc;
            }
        }
    } else {
        for (uint32_t e = tid; e < p.M * KSUB; e += 256) {
            const uint32_t m = e / KSUB, j = e % KSUB;
            const float *s = p.codebooks + ((size_t)m * KSUB + j) * p.dsub;
            float acc = 0.f;
            for (uint32_t t = 0; t < p.dsub; ++t) {
                const float r = x[m * p.dsub + t] - c[m * p.dsub + t];
                const float diff = r - s[t];
                acc = acc + diff * diff;                                       // contraction is off: mul, then add
            }
            lut[e] = acc;
        }
    }
    __syncthreads();
    const uint64_t first = p.list_off[list], count = p.list_off[list + 1] - first;
    const uint64_t out0 = p.out_off[qp];
    if (p.M == 32) {
        // two codes per iteration: their four 16-byte loads and two ids are requested before the first table look-up
        auto adc = [&](const uint4 &lo, const uint4 &hi) {
            const uint32_t ws[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            float dis = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) dis = dis + lut[i * KSUB + ((ws[i >> 2] >> (8 * (i & 3))) & 255)];
            return dis;
        };
        uint64_t v = tid;
        for (; v + 256 < count; v += 512) {
            const uint4 *ca = reinterpret_cast<const uint4 *>(p.codes + (first + v) * 32), *cb = reinterpret_cast<const uint4 *>(p.codes + (first + v + 256) * 32);
            const uint4 a0 = ca[0], a1 = ca[1], b0 = cb[0], b1 = cb[1];
            const int64_t ia = p.ids[first + v], ib = p.ids[first + v + 256];
            p.D[out0 + v] = adc(a0, a1); p.I[out0 + v] = ia;
            p.D[out0 + v + 256] = adc(b0, b1); p.I[out0 + v + 256] = ib;
        }
        if (v < count) {
            const uint4 *ca = reinterpret_cast<const uint4 *>(p.codes + (first + v) * 32);
            const uint4 a0 = ca[0], a1 = ca[1];
            p.D[out0 + v] = adc(a0, a1); p.I[out0 + v] = p.ids[first + v];
        }
        return;
    }
    for (uint64_t v = tid; v < count; v += 256) {
        const uint8_t *code = p.codes + (first + v) * p.M;
        float dis = 0.f;
        if ((p.M & 15) == 0) {
            for (uint32_t m0 = 0; m0 < p.M; m0 += 16) {
                const uint4 w = *reinterpret_cast<const uint4 *>(code + m0);
                const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int i = 0; i < 16; ++i) dis = dis + lut[(m0 + i) * KSUB + ((ws[i >> 2] >> (8 * (i & 3))) & 255)];
            }
        } else {
            for (uint32_t m = 0; m < p.M; ++m) dis = dis + lut[m * KSUB + code[m]];
        }
        p.D[out0 + v] = dis;
        p.I[out0 + v] = p.ids[first + v];
    }
}

}  // namespace pf

using namespace pf;

struct pf_ivfpq {
    int device = 0;
    uint32_t d = 0, nlist = 0, M = 0, dsub = 0;
    float *centroids = nullptr, *codebooks = nullptr;
    // inverted lists: host masters (append order per list), flattened device copy rebuilt lazily after add
    std::vector<std::vector<uint8_t>> h_codes;
    std::vector<std::vector<int64_t>> h_ids;
    std::vector<uint64_t> h_off;
    uint8_t *d_codes = nullptr;
    int64_t *d_ids = nullptr;
    uint64_t *d_off = nullptr;
    bool dirty = true;
    size_t ntotal = 0;
    // per-call staging (probe ids, output offsets), grown on demand: device copies and the pinned host buffers they are
    // filled from; `staged` marks the end of the last call's copies (the next call waits for it before it refills the
    // buffers -- in practice it has long passed: no stream synchronisation per call)
    int64_t *d_probe = nullptr;
    uint64_t *d_outoff = nullptr;
    int64_t *pin_probe = nullptr;
    uint64_t *pin_outoff = nullptr;
    hipEvent_t staged = nullptr;
    hipEvent_t scanned = nullptr;     // end of the last call's scan kernel: the next call's stream waits for it before it overwrites d_probe / d_outoff
    size_t stage_cap = 0;
};

namespace {

pf_status flush_lists(pf_ivfpq *h) {
    if (!h->dirty) return PF_OK;
    h->h_off.assign(h->nlist + 1, 0);
    for (uint32_t l = 0; l < h->nlist; ++l) h->h_off[l + 1] = h->h_off[l] + h->h_ids[l].size();
    const size_t n = h->h_off[h->nlist];
    std::vector<uint8_t> codes(n * h->M ? n * h->M : 1);
    std::vector<int64_t> ids(n ? n : 1);
    for (uint32_t l = 0; l < h->nlist; ++l) {
        if (h->h_ids[l].empty()) continue;
        std::copy(h->h_codes[l].begin(), h->h_codes[l].end(), codes.begin() + h->h_off[l] * h->M);
        std::copy(h->h_ids[l].begin(), h->h_ids[l].end(), ids.begin() + h->h_off[l]);
    }
    if (h->d_codes) { PF_HIP(hipFree(h->d_codes)); h->d_codes = nullptr; }
    if (h->d_ids) { PF_HIP(hipFree(h->d_ids)); h->d_ids = nullptr; }
    if (!h->d_off) PF_HIP(hipMalloc((void **)&h->d_off, (h->nlist + 1) * 8));
    PF_HIP(hipMalloc((void **)&h->d_codes, codes.size()));
    PF_HIP(hipMalloc((void **)&h->d_ids, ids.size() * 8));
    PF_HIP(hipMemcpy(h->d_codes, codes.data(), codes.size(), hipMemcpyHostToDevice));
    PF_HIP(hipMemcpy(h->d_ids, ids.data(), ids.size() * 8, hipMemcpyHostToDevice));
    PF_HIP(hipMemcpy(h->d_off, h->h_off.data(), (h->nlist + 1) * 8, hipMemcpyHostToDevice));
    h->ntotal = n;
    h->dirty = false;
    return PF_OK;
}

}  // namespace

extern "C" {

pf_status pf_ivfpq_destroy(pf_ivfpq *h) {
    if (!h) return PF_OK;
    {
        DeviceGuard g(h->device);
        for (void *p : {(void *)h->centroids, (void *)h->codebooks, (void *)h->d_codes, (void *)h->d_ids, (void *)h->d_off,
                        (void *)h->d_probe, (void *)h->d_outoff})
            if (p) (void)hipFree(p);
        if (h->pin_probe) (void)hipHostFree(h->pin_probe);
        if (h->pin_outoff) (void)hipHostFree(h->pin_outoff);
        if (h->staged) (void)hipEventDestroy(h->staged);
        if (h->scanned) (void)hipEventDestroy(h->scanned);
    }
    delete h;
    return PF_OK;
}

pf_status pf_ivfpq_create(pf_ivfpq **out, int device, uint32_t d, uint32_t nlist, uint32_t M, const float *centroids_host,
                          const float *codebooks_host) {
    if (!out || !centroids_host || !codebooks_host) return fail(PF_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (d == 0 || nlist == 0 || M == 0 || d % M) return fail(PF_ERR_INVALID_ARG, "d must be a positive multiple of M");
    if ((size_t)M * KSUB * 4 > 64 * 1024) return fail(PF_ERR_UNSUPPORTED, "M > 64 sub-quantizers: look-up table exceeds the 64 KiB dynamic LDS default");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(PF_ERR_NO_DEVICE, "no such HIP device");
    PF_GUARD(device);
    pf_ivfpq *h = new pf_ivfpq;
    h->device = device; h->d = d; h->nlist = nlist; h->M = M; h->dsub = d / M;
    h->h_codes.resize(nlist); h->h_ids.resize(nlist);
    const size_t cb = (size_t)nlist * d * 4, kb = (size_t)M * KSUB * h->dsub * 4;
    hipError_t e = hipMalloc((void **)&h->centroids, cb);
    if (e == hipSuccess) e = hipMalloc((void **)&h->codebooks, kb);
    if (e == hipSuccess) e = hipMemcpy(h->centroids, centroids_host, cb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->codebooks, codebooks_host, kb, hipMemcpyHostToDevice);
    if (e != hipSuccess) { pf_ivfpq_destroy(h); return fail(PF_ERR_HIP, std::string("pf_ivfpq_create: ") + hipGetErrorString(e)); }
    *out = h;
    return PF_OK;
}

// Appends n encoded vectors: list_ids_host[i] = inverted list, codes_host[i][M] = PQ code, ids_host[i] = label.
// (Assignment and encoding are the caller's: see IvfPqTrainer in server_lib.cpp, which runs them on the GPU flat index.)
pf_status pf_ivfpq_add_encoded(pf_ivfpq *h, size_t n, const int64_t *list_ids_host, const uint8_t *codes_host, const int64_t *ids_host) {
    if (!h || (n && (!list_ids_host || !codes_host || !ids_host))) return fail(PF_ERR_INVALID_ARG, "null argument");
    for (size_t i = 0; i < n; ++i)
        if (list_ids_host[i] < 0 || (uint64_t)list_ids_host[i] >= h->nlist) return fail(PF_ERR_INVALID_ARG, "list id out of range");
    for (size_t i = 0; i < n; ++i) {
        auto &lc = h->h_codes[list_ids_host[i]];
        lc.insert(lc.end(), codes_host + i * h->M, codes_host + (i + 1) * h->M);
        h->h_ids[list_ids_host[i]].push_back(ids_host[i]);
    }
    h->dirty = true;
    return PF_OK;
}

pf_status pf_ivfpq_info(const pf_ivfpq *h, uint32_t *d, uint32_t *nlist, uint32_t *M, size_t *ntotal, uint64_t *list_sizes_host) {
    if (!h) return fail(PF_ERR_INVALID_ARG, "null index");
    if (d) *d = h->d;
    if (nlist) *nlist = h->nlist;
    if (M) *M = h->M;
    size_t n = 0;
    for (uint32_t l = 0; l < h->nlist; ++l) { n += h->h_ids[l].size(); if (list_sizes_host) list_sizes_host[l] = h->h_ids[l].size(); }
    if (ntotal) *ntotal = n;
    return PF_OK;
}

// Copies out the stored entries of one list (host buffers sized by pf_ivfpq_info): test / serialisation hook.
pf_status pf_ivfpq_get_list(const pf_ivfpq *h, uint32_t list, uint8_t *codes_host, int64_t *ids_host) {
    if (!h || list >= h->nlist) return fail(PF_ERR_INVALID_ARG, "bad list");
    if (codes_host) std::copy(h->h_codes[list].begin(), h->h_codes[list].end(), codes_host);
    if (ids_host) std::copy(h->h_ids[list].begin(), h->h_ids[list].end(), ids_host);
    return PF_OK;
}

// IndexIVFPQ::search_encrypted.  probe_host [nq][nprobe] list ids (host: they arrive in the request), xq device.
// D/I device buffers of `capacity` entries; list_sizes_host[q] = results of query q, written back to back.
// Returns PF_ERR_INVALID_ARG if capacity is too small (nothing is launched).
pf_status pf_ivfpq_search_lists(pf_ivfpq *h, const float *xq, const int64_t *probe_host, size_t nq, uint32_t nprobe, float *D, int64_t *I,
                                size_t capacity, uint64_t *list_sizes_host, pf_stream stream) {
    if (!h) return fail(PF_ERR_INVALID_ARG, "null index");
    if (nq == 0 || nprobe == 0) return PF_OK;
    if (!xq || !probe_host || !D || !I || !list_sizes_host) return fail(PF_ERR_INVALID_ARG, "null argument");
    PF_GUARD(h->device);
    pf_status st = flush_lists(h);
    if (st != PF_OK) return st;
    const size_t need = nq * nprobe;
    if (need > h->stage_cap) {
        if (h->staged) PF_HIP(hipEventSynchronize(h->staged));
        if (h->scanned) PF_HIP(hipEventSynchronize(h->scanned));      // the last scan still reads the buffers freed below
        if (h->d_probe) PF_HIP(hipFree(h->d_probe));
        if (h->d_outoff) PF_HIP(hipFree(h->d_outoff));
        if (h->pin_probe) PF_HIP(hipHostFree(h->pin_probe));
        if (h->pin_outoff) PF_HIP(hipHostFree(h->pin_outoff));
        h->d_probe = nullptr; h->d_outoff = nullptr; h->pin_probe = nullptr; h->pin_outoff = nullptr; h->stage_cap = 0;
        PF_HIP(hipMalloc((void **)&h->d_probe, need * 8));
        PF_HIP(hipMalloc((void **)&h->d_outoff, need * 8));
        PF_HIP(hipHostMalloc((void **)&h->pin_probe, need * 8, hipHostMallocDefault));
        PF_HIP(hipHostMalloc((void **)&h->pin_outoff, need * 8, hipHostMallocDefault));
        h->stage_cap = need;
    }
    if (!h->staged) PF_HIP(hipEventCreateWithFlags(&h->staged, hipEventDisableTiming));
    else PF_HIP(hipEventSynchronize(h->staged));                   // the previous call's copies out of the pinned buffers are done
    uint64_t total = 0;
    for (size_t q = 0; q < nq; ++q) {
        uint64_t per_q = 0;
        for (uint32_t j = 0; j < nprobe; ++j) {
            const int64_t l = probe_host[q * nprobe + j];
            h->pin_probe[q * nprobe + j] = l;
            h->pin_outoff[q * nprobe + j] = total + per_q;
            if (l >= 0 && (uint64_t)l < h->nlist) per_q += h->h_ids[l].size();
        }
        list_sizes_host[q] = per_q;
        total += per_q;
    }
    if (total > capacity) return fail(PF_ERR_INVALID_ARG, "output capacity too small for the probed lists");
    hipStream_t s = as_stream(stream);
    // the DEVICE staging buffers are read by the previous call's scan kernel, possibly on another stream: this call's stream
    // waits for that kernel (a device-side wait, the host does not block) before the copies overwrite them
    if (h->scanned) PF_HIP(hipStreamWaitEvent(s, h->scanned, 0));
    else PF_HIP(hipEventCreateWithFlags(&h->scanned, hipEventDisableTiming));
    PF_HIP(hipMemcpyAsync(h->d_probe, h->pin_probe, need * 8, hipMemcpyHostToDevice, s));
    PF_HIP(hipMemcpyAsync(h->d_outoff, h->pin_outoff, need * 8, hipMemcpyHostToDevice, s));
    PF_HIP(hipEventRecord(h->staged, s));
    ScanArgs a{xq, h->centroids, h->codebooks, h->d_codes, h->d_ids, h->d_off, h->d_probe, h->d_outoff, D, I, h->d, h->M, h->dsub, nprobe, h->nlist};
    hipLaunchKernelGGL(k_ivfpq_scan, dim3((unsigned)need), dim3(256), h->M * KSUB * 4, s, a);
    PF_HIP(hipGetLastError());
    PF_HIP(hipEventRecord(h->scanned, s));
    return PF_OK;
}

}  // extern "C"
