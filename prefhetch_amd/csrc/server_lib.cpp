// server_lib.cpp -- class Server (include/server/server_lib.h) over the C ABI of libprefhetch_hip.so.
// Host C++ only: no HIP headers, no faiss, no spdlog.  Mirrors the behaviour of
// /root/reference/src/server/server_lib.cpp: same outputs for the same inputs, std::runtime_error for
// unusable state, out-parameters resized by the callee.
#include "../../include/server/server_lib.h"

#include <cstring>
#include <filesystem>
#include <mutex>
#include <stdexcept>
#include <string>

#include "../../include/prefhetch_hip.h"

#ifdef PREFHETCH_WITH_DROGON
#include <drogon/drogon.h>
#endif

namespace {

// Dataset locations of the reference deployment (src/server/server_lib.cpp:22-27), relative to build/.
const char *const kBasePath = "../sift/siftsmall/siftsmall_base.fvecs";
const char *const kCentroidCachePath = "NBASE10000_IVF256_centroids.fvecs";
const char *const kListenAddress = "0.0.0.0";
constexpr int kListenPort = 8080;

void check(pf_status st, const char *what) {
    if (st != PF_OK) throw std::runtime_error(std::string(what) + ": " + pf_status_str(st) + " (" + pf_last_error() + ")");
}

// a device buffer owned through the C ABI
struct DevBuf {
    int device = 0;
    void *ptr = nullptr;
    size_t bytes = 0;
    void reserve(int dev, size_t n) {
        if (n <= bytes) return;
        release();
        check(pf_malloc(dev, &ptr, n), "pf_malloc");
        device = dev; bytes = n;
    }
    void release() { if (ptr) pf_free(device, ptr); ptr = nullptr; bytes = 0; }
    ~DevBuf() { release(); }
};

}  // namespace

struct Server::Impl {
    int device = 0;
    pf_flat *base = nullptr;        // NBASE x 128 fp32 in HBM (reference: m_DatasetBase)
    pf_flat *centroids = nullptr;   // NLIST x 128 fp32 in HBM (reference: m_Quantizer / m_Index->quantizer)
    size_t nb = 0, nlist = 0;
    // the reference's handlers run on one Drogon loop thread; this lock makes concurrent const calls safe anyway
    mutable std::mutex lock;
    mutable DevBuf d_query, d_ids, d_out, d_out2;

    ~Impl() {
        if (base) pf_flat_destroy(base);
        if (centroids) pf_flat_destroy(centroids);
    }
    void require_ready() const {
        if (!base || !centroids) throw std::runtime_error("Server: index not initialised (call init_index or init_from_memory)");
    }
};

Server::Server() : m_Impl(std::make_unique<Impl>()) {}
Server::~Server() = default;

void Server::init_from_memory(const float *base, size_t nb, const float *centroids, size_t nlist, int device) {
    if (!base || !centroids || nb == 0 || nlist == 0) throw std::runtime_error("Server::init_from_memory: empty input");
    std::lock_guard<std::mutex> g(m_Impl->lock);
    if (m_Impl->base) { pf_flat_destroy(m_Impl->base); m_Impl->base = nullptr; }
    if (m_Impl->centroids) { pf_flat_destroy(m_Impl->centroids); m_Impl->centroids = nullptr; }
    m_Impl->device = device;
    check(pf_flat_create(&m_Impl->base, device, base, nb, PRECISE_VECTOR_DIMENSIONS), "pf_flat_create(base)");
    check(pf_flat_create(&m_Impl->centroids, device, centroids, nlist, PRECISE_VECTOR_DIMENSIONS), "pf_flat_create(centroids)");
    m_Impl->nb = nb; m_Impl->nlist = nlist;
}

void Server::init_index() {
    size_t d = 0, nb = 0;
    std::vector<float> base;
    vecs_read<float>(kBasePath, d, nb, base);                  // aborts when the dataset is missing, like the reference
    if (d != static_cast<size_t>(PRECISE_VECTOR_DIMENSIONS))
        throw std::runtime_error("Incorrect dimensions for base set, not the same as PRECISE_VECTOR_DIMENSIONS");
    if (!std::filesystem::exists(kCentroidCachePath))
        throw std::runtime_error(std::string("no centroid cache ") + kCentroidCachePath +
                                 ": IVFPQ training is outside this build; produce the centroids with the reference and export them as fvecs");
    size_t dc = 0, nc = 0;
    std::vector<float> cent;
    vecs_read<float>(kCentroidCachePath, dc, nc, cent);
    if (dc != static_cast<size_t>(PRECISE_VECTOR_DIMENSIONS) || nc != static_cast<size_t>(NLIST))
        throw std::runtime_error("centroid cache does not hold NLIST x PRECISE_VECTOR_DIMENSIONS values");
    init_from_memory(base.data(), nb, cent.data(), nc, 0);
}

void Server::run_webserver() {
#ifdef PREFHETCH_WITH_DROGON
    drogon::app().addListener(kListenAddress, kListenPort);
    drogon::app().run();
#else
    (void)kListenAddress; (void)kListenPort;
    throw std::runtime_error("Server::run_webserver: built without Drogon (-DPREFHETCH_WITH_DROGON)");
#endif
}

void Server::retrieve_centroids(std::vector<std::array<float, PRECISE_VECTOR_DIMENSIONS>> &centroids) const {
    const Impl &im = *m_Impl;
    std::lock_guard<std::mutex> g(im.lock);
    im.require_ready();
    centroids.resize(NLIST);
    std::vector<int64_t> ids(NLIST);
    for (int64_t i = 0; i < NLIST; ++i) ids[i] = i;
    const size_t out_bytes = static_cast<size_t>(NLIST) * PRECISE_VECTOR_DIMENSIONS * sizeof(float);
    im.d_ids.reserve(im.device, ids.size() * sizeof(int64_t));
    im.d_out.reserve(im.device, out_bytes);
    check(pf_memcpy_h2d(im.device, im.d_ids.ptr, ids.data(), ids.size() * sizeof(int64_t), nullptr), "h2d");
    check(pf_gather_rows(im.centroids, static_cast<const int64_t *>(im.d_ids.ptr), ids.size(), static_cast<float *>(im.d_out.ptr), nullptr), "pf_gather_rows");
    check(pf_memcpy_d2h(im.device, centroids.data(), im.d_out.ptr, out_bytes, nullptr), "d2h");
    check(pf_stream_synchronize(im.device, nullptr), "sync");
}

void Server::coarseSearch(const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &,
                          const std::array<std::array<faiss::idx_t, NPROBE>, NQUERY> &, std::vector<float> &,
                          std::vector<faiss::idx_t> &, std::array<size_t, NQUERY> &) const {
    // faiss::IndexIVFPQ::search_encrypted of the PreFHEtch-faiss fork (reference src/server/server_lib.cpp:126-130):
    // the IVF-PQ coarse stage is the first "next" row of SURVEY.md section 8f and is not built in this round.
    throw std::runtime_error("Server::coarseSearch: the IVFPQ coarse stage is not part of this build yet");
}

void Server::preciseSearch(const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &precise_query,
                           const std::array<std::array<faiss::idx_t, COARSE_PROBE>, NQUERY> &nearest_coarse_vector_idx,
                           std::array<std::array<float, COARSE_PROBE>, NQUERY> &precise_distance_scores) const {
    const Impl &im = *m_Impl;
    std::lock_guard<std::mutex> g(im.lock);
    im.require_ready();
    im.d_query.reserve(im.device, sizeof precise_query);
    im.d_ids.reserve(im.device, sizeof nearest_coarse_vector_idx);
    im.d_out.reserve(im.device, sizeof precise_distance_scores);
    check(pf_memcpy_h2d(im.device, im.d_query.ptr, precise_query.data(), sizeof precise_query, nullptr), "h2d");
    check(pf_memcpy_h2d(im.device, im.d_ids.ptr, nearest_coarse_vector_idx.data(), sizeof nearest_coarse_vector_idx, nullptr), "h2d");
    check(pf_l2_gathered(im.base, static_cast<const float *>(im.d_query.ptr), static_cast<const int64_t *>(im.d_ids.ptr), NQUERY,
                         COARSE_PROBE, static_cast<float *>(im.d_out.ptr), nullptr), "pf_l2_gathered");
    check(pf_memcpy_d2h(im.device, precise_distance_scores.data(), im.d_out.ptr, sizeof precise_distance_scores, nullptr), "d2h");
    check(pf_stream_synchronize(im.device, nullptr), "sync");
}

void Server::preciseVectorPIR(const std::array<std::array<faiss_idx_t, K>, NQUERY> &k_nearest_precise_vectors_idx,
                              std::array<std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, K>, NQUERY> &query_results) {
    Impl &im = *m_Impl;
    std::lock_guard<std::mutex> g(im.lock);
    im.require_ready();
    im.d_ids.reserve(im.device, sizeof k_nearest_precise_vectors_idx);
    im.d_out.reserve(im.device, sizeof query_results);
    check(pf_memcpy_h2d(im.device, im.d_ids.ptr, k_nearest_precise_vectors_idx.data(), sizeof k_nearest_precise_vectors_idx, nullptr), "h2d");
    check(pf_gather_rows(im.base, static_cast<const int64_t *>(im.d_ids.ptr), static_cast<size_t>(NQUERY) * K, static_cast<float *>(im.d_out.ptr), nullptr), "pf_gather_rows");
    check(pf_memcpy_d2h(im.device, query_results.data(), im.d_out.ptr, sizeof query_results, nullptr), "d2h");
    check(pf_stream_synchronize(im.device, nullptr), "sync");
}

void Server::nearestCentroids(const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &precise_query,
                              std::array<std::array<faiss::idx_t, NPROBE>, NQUERY> &nearest_centroid_idx,
                              std::array<std::array<float, NPROBE>, NQUERY> &nearest_centroid_dist) const {
    const Impl &im = *m_Impl;
    std::lock_guard<std::mutex> g(im.lock);
    im.require_ready();
    im.d_query.reserve(im.device, sizeof precise_query);
    im.d_out.reserve(im.device, sizeof nearest_centroid_dist);
    im.d_out2.reserve(im.device, sizeof nearest_centroid_idx);
    check(pf_memcpy_h2d(im.device, im.d_query.ptr, precise_query.data(), sizeof precise_query, nullptr), "h2d");
    check(pf_flat_search(im.centroids, static_cast<const float *>(im.d_query.ptr), NQUERY, NPROBE, static_cast<float *>(im.d_out.ptr),
                         static_cast<int64_t *>(im.d_out2.ptr), nullptr), "pf_flat_search");
    check(pf_memcpy_d2h(im.device, nearest_centroid_dist.data(), im.d_out.ptr, sizeof nearest_centroid_dist, nullptr), "d2h");
    check(pf_memcpy_d2h(im.device, nearest_centroid_idx.data(), im.d_out2.ptr, sizeof nearest_centroid_idx, nullptr), "d2h");
    check(pf_stream_synchronize(im.device, nullptr), "sync");
}
