// server_lib.cpp -- class Server (include/server/server_lib.h) over the C ABI of libprefhetch_hip.so.
// Host C++ only: no HIP headers, no faiss, no spdlog.  Mirrors the behaviour of
// /root/reference/src/server/server_lib.cpp: same outputs for the same inputs, std::runtime_error for
// unusable state, out-parameters resized by the callee.
#include "../../include/server/server_lib.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <string>

#include "../../include/prefhetch_hip.h"
#include "../../include/server/http.h"
#include "../../include/client/pir.h"
#include <memory>
#include <cstdlib>

#ifdef PREFHETCH_WITH_DROGON
#include <drogon/drogon.h>
#endif

namespace {

// Dataset locations of the reference deployment (src/server/server_lib.cpp:22-27), relative to build/.
const char *const kTrainPath = "../sift/siftsmall/siftsmall_learn.fvecs";
const char *const kBasePath = "../sift/siftsmall/siftsmall_base.fvecs";
// the reference names its cache NBASE10000_PRECISE_DIMENSIONS_IVF256_PQ32_SUB_QUANTIZER_SIZE8.faiss (server_lib.cpp:38-42)
const char *const kIndexCachePath = "NBASE10000_PRECISE_DIMENSIONS_IVF256_PQ32_SUB_QUANTIZER_SIZE8.pfivfpq";
constexpr int kKmeansIters = 25;         // faiss ClusteringParameters default
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
        if (ptr && dev == device && n <= bytes) return;           // a buffer of another device is never reused
        release();
        check(pf_malloc(dev, &ptr, n), "pf_malloc");
        device = dev; bytes = n;
    }
    void release() { if (ptr) pf_free(device, ptr); ptr = nullptr; bytes = 0; }
    ~DevBuf() { release(); }
};

// Nearest row of `table` [k][d] for every row of x_dev [n][d] (device), through the GPU flat-L2 index
// (squared L2, tie -> smaller id): the assignment step of k-means, IVF list assignment and PQ encoding.
std::vector<int64_t> nearest_rows(int dev, const float *table, size_t k, uint32_t d, const float *x_dev, size_t n) {
    pf_flat *idx = nullptr;
    check(pf_flat_create(&idx, dev, table, k, d), "pf_flat_create");
    std::vector<int64_t> out(n);
    DevBuf dD, dI;
    const size_t step = 1u << 16;
    dD.reserve(dev, step * sizeof(float));
    dI.reserve(dev, step * sizeof(int64_t));
    for (size_t i0 = 0; i0 < n; i0 += step) {
        const size_t m = std::min(step, n - i0);
        pf_status st = pf_flat_search(idx, x_dev + i0 * d, m, 1, static_cast<float *>(dD.ptr), static_cast<int64_t *>(dI.ptr), nullptr);
        if (st == PF_OK) st = pf_memcpy_d2h(dev, out.data() + i0, dI.ptr, m * sizeof(int64_t), nullptr);
        if (st == PF_OK) st = pf_stream_synchronize(dev, nullptr);
        if (st != PF_OK) { pf_flat_destroy(idx); check(st, "nearest_rows"); }
    }
    pf_flat_destroy(idx);
    return out;
}

// Lloyd k-means, deterministic: centroids start at evenly spaced training rows, kKmeansIters iterations, means in
// double, an emptied cluster keeps its previous centre.  (faiss seeds from a random permutation; a trained index is
// not reproducible across libraries anyway -- parity for the coarse stage is defined GIVEN the index content.)
std::vector<float> kmeans(int dev, const std::vector<float> &x, size_t n, uint32_t d, size_t k) {
    if (n < k) throw std::runtime_error("k-means: fewer training vectors than centroids");
    std::vector<float> cent(k * d);
    for (size_t c = 0; c < k; ++c) std::copy_n(x.data() + (c * n / k) * d, d, cent.data() + c * d);
    DevBuf dx;
    dx.reserve(dev, n * d * sizeof(float));
    check(pf_memcpy_h2d(dev, dx.ptr, x.data(), n * d * sizeof(float), nullptr), "h2d");
    check(pf_stream_synchronize(dev, nullptr), "sync");
    std::vector<double> sum(k * d);
    std::vector<size_t> cnt(k);
    for (int it = 0; it < kKmeansIters; ++it) {
        const std::vector<int64_t> a = nearest_rows(dev, cent.data(), k, d, static_cast<const float *>(dx.ptr), n);
        std::fill(sum.begin(), sum.end(), 0.0);
        std::fill(cnt.begin(), cnt.end(), 0);
        for (size_t i = 0; i < n; ++i) {
            ++cnt[a[i]];
            for (uint32_t t = 0; t < d; ++t) sum[a[i] * d + t] += x[i * d + t];
        }
        for (size_t c = 0; c < k; ++c)
            if (cnt[c]) for (uint32_t t = 0; t < d; ++t) cent[c * d + t] = static_cast<float>(sum[c * d + t] / static_cast<double>(cnt[c]));
    }
    return cent;
}

constexpr uint32_t kD = PRECISE_VECTOR_DIMENSIONS, kM = SUB_QUANTIZERS, kDsub = kD / kM, kKsub = 1u << SUB_QUANTIZER_SIZE;

// residual sub-vectors of sub-quantizer m: (x_i - centroid[list_i]) restricted to dims [m*dsub, (m+1)*dsub)
std::vector<float> residual_slice(const float *x, size_t n, const std::vector<int64_t> &list, const std::vector<float> &cent, uint32_t m) {
    std::vector<float> r(n * kDsub);
    for (size_t i = 0; i < n; ++i)
        for (uint32_t t = 0; t < kDsub; ++t) r[i * kDsub + t] = x[i * kD + m * kDsub + t] - cent[list[i] * kD + m * kDsub + t];
    return r;
}

}  // namespace

// private row retrieval (include/client/pir.h): BFV context with a special prime + the packed base, built on first use
struct PirState {
    bfv::Context ctx;
    pir::Database db;
    PirState(int device, const float *rows, size_t n_rows)
        : ctx(bfv::Params::seal_default(Server::ENC_RING_DEGREE, Server::PIR_PLAIN_MODULUS, device)), db(ctx, rows, n_rows, static_cast<uint32_t>(PRECISE_VECTOR_DIMENSIONS)) {}
};

struct Server::Impl {
    int device = 0;
    pf_flat *base = nullptr;        // NBASE x 128 fp32 in HBM (reference: m_DatasetBase)
    pf_flat *centroids = nullptr;   // NLIST x 128 fp32 in HBM (reference: m_Quantizer / m_Index->quantizer)
    pf_ivfpq *ivfpq = nullptr;      // inverted lists + PQ codes (reference: m_Index)
    pf_ctx *ring = nullptr;         // RNS ring of the encrypted round: N = 8192, SEAL BFVDefault data primes
    std::vector<float> h_centroids, h_codebooks;
    size_t nb = 0, nlist = 0;
    // the reference's handlers run on one Drogon loop thread; this lock makes concurrent const calls safe anyway
    mutable std::mutex lock;
    mutable DevBuf d_query, d_ids, d_out, d_out2;
    mutable std::unique_ptr<PirState> pir;
    PirState &pir_state() const;

    ~Impl() {
        reset();
    }
    void reset() {
        if (base) pf_flat_destroy(base);
        if (centroids) pf_flat_destroy(centroids);
        if (ivfpq) pf_ivfpq_destroy(ivfpq);
        if (ring) pf_ctx_destroy(ring);
        pir.reset();
        base = nullptr; centroids = nullptr; ivfpq = nullptr; ring = nullptr;
        d_query.release(); d_ids.release(); d_out.release(); d_out2.release();   // they belong to the old device
    }
    // installs trained tables and creates the device objects
    void install(int dev, const float *base_rows, size_t n_base, std::vector<float> cent, std::vector<float> books) {
        reset();
        device = dev;
        h_centroids = std::move(cent); h_codebooks = std::move(books);
        nlist = h_centroids.size() / kD; nb = n_base;
        check(pf_flat_create(&base, dev, base_rows, n_base, kD), "pf_flat_create(base)");
        check(pf_flat_create(&centroids, dev, h_centroids.data(), nlist, kD), "pf_flat_create(centroids)");
        check(pf_ivfpq_create(&ivfpq, dev, kD, static_cast<uint32_t>(nlist), kM, h_centroids.data(), h_codebooks.data()), "pf_ivfpq_create");
        check(pf_ctx_create(&ring, dev, Server::ENC_RING_DEGREE, Server::ENC_LIMBS, Server::ENC_MODULI), "pf_ctx_create");
    }
    // IndexIVFPQ::add: assign to the nearest coarse centroid, encode the residual, append (ids = row numbers)
    void add(const float *x, size_t n) {
        DevBuf dx;
        dx.reserve(device, n * kD * sizeof(float));
        check(pf_memcpy_h2d(device, dx.ptr, x, n * kD * sizeof(float), nullptr), "h2d");
        check(pf_stream_synchronize(device, nullptr), "sync");
        const std::vector<int64_t> list = nearest_rows(device, h_centroids.data(), nlist, kD, static_cast<const float *>(dx.ptr), n);
        std::vector<uint8_t> codes(n * kM);
        DevBuf dr;
        dr.reserve(device, n * kDsub * sizeof(float));
        for (uint32_t m = 0; m < kM; ++m) {
            const std::vector<float> r = residual_slice(x, n, list, h_centroids, m);
            check(pf_memcpy_h2d(device, dr.ptr, r.data(), r.size() * sizeof(float), nullptr), "h2d");
            check(pf_stream_synchronize(device, nullptr), "sync");
            const std::vector<int64_t> c = nearest_rows(device, h_codebooks.data() + static_cast<size_t>(m) * kKsub * kDsub, kKsub, kDsub,
                                                        static_cast<const float *>(dr.ptr), n);
            for (size_t i = 0; i < n; ++i) codes[i * kM + m] = static_cast<uint8_t>(c[i]);
        }
        std::vector<int64_t> ids(n);
        std::iota(ids.begin(), ids.end(), static_cast<int64_t>(0));
        check(pf_ivfpq_add_encoded(ivfpq, n, list.data(), codes.data(), ids.data()), "pf_ivfpq_add_encoded");
    }
    void require_ready() const {
        if (!base || !centroids || !ivfpq) throw std::runtime_error("Server: index not initialised (call init_index or init_from_memory)");
    }
};

Server::Server() : m_Impl(std::make_unique<Impl>()) {}
Server::~Server() = default;

void Server::init_from_memory(const float *base, size_t nb, const float *train, size_t nt, int device) {
    if (!base || nb == 0) throw std::runtime_error("Server::init_from_memory: empty input");
    if (!train) { train = base; nt = nb; }
    std::lock_guard<std::mutex> g(m_Impl->lock);
    // IndexIVFPQ::train: coarse quantizer, then the product quantizer on the residuals of the training set
    const std::vector<float> xt(train, train + nt * kD);
    std::vector<float> cent = kmeans(device, xt, nt, kD, static_cast<size_t>(NLIST));
    std::vector<int64_t> list;
    {
        DevBuf dx;
        dx.reserve(device, nt * kD * sizeof(float));
        check(pf_memcpy_h2d(device, dx.ptr, train, nt * kD * sizeof(float), nullptr), "h2d");
        check(pf_stream_synchronize(device, nullptr), "sync");
        list = nearest_rows(device, cent.data(), NLIST, kD, static_cast<const float *>(dx.ptr), nt);
    }
    std::vector<float> books(static_cast<size_t>(kM) * kKsub * kDsub);
    for (uint32_t m = 0; m < kM; ++m) {
        const std::vector<float> sub = kmeans(device, residual_slice(train, nt, list, cent, m), nt, kDsub, kKsub);
        std::copy(sub.begin(), sub.end(), books.begin() + static_cast<size_t>(m) * kKsub * kDsub);
    }
    m_Impl->install(device, base, nb, std::move(cent), std::move(books));
    m_Impl->add(base, nb);
}

namespace {
constexpr char kMagic[8] = {'P', 'F', 'I', 'V', 'F', 'P', 'Q', '1'};
}

void Server::init_index() {
    size_t d = 0, nb = 0;
    std::vector<float> base;
    if (!std::filesystem::exists(kIndexCachePath)) {
        size_t nt = 0;
        std::vector<float> xt;
        vecs_read<float>(kTrainPath, d, nt, xt);                  // aborts when the dataset is missing, like the reference
        if (d != static_cast<size_t>(PRECISE_VECTOR_DIMENSIONS))
            throw std::runtime_error("Incorrect dimensions for train set, not the same as PRECISE_VECTOR_DIMENSIONS");
        size_t d2 = 0;
        vecs_read<float>(kBasePath, d2, nb, base);
        if (d2 != d) throw std::runtime_error("dataset does not have same dimension as train set");
        init_from_memory(base.data(), nb, xt.data(), nt, 0);
        // cache the trained index (the reference: faiss::write_index, server_lib.cpp:82)
        std::vector<float> cent, books; std::vector<uint8_t> codes; std::vector<faiss::idx_t> ids; std::vector<uint64_t> off;
        export_index(cent, books, codes, ids, off);
        std::ofstream f(kIndexCachePath, std::ios::binary);
        const uint64_t hdr[3] = {cent.size(), books.size(), ids.size()};
        f.write(kMagic, 8); f.write(reinterpret_cast<const char *>(hdr), sizeof hdr);
        f.write(reinterpret_cast<const char *>(cent.data()), cent.size() * 4);
        f.write(reinterpret_cast<const char *>(books.data()), books.size() * 4);
        f.write(reinterpret_cast<const char *>(off.data()), off.size() * 8);
        f.write(reinterpret_cast<const char *>(ids.data()), ids.size() * 8);
        f.write(reinterpret_cast<const char *>(codes.data()), codes.size());
        return;
    }
    vecs_read<float>(kBasePath, d, nb, base);
    std::ifstream f(kIndexCachePath, std::ios::binary);
    char magic[8]; uint64_t hdr[3];
    f.read(magic, 8); f.read(reinterpret_cast<char *>(hdr), sizeof hdr);
    if (!f || std::memcmp(magic, kMagic, 8) != 0 || hdr[0] != static_cast<uint64_t>(NLIST) * kD || hdr[1] != static_cast<uint64_t>(kM) * kKsub * kDsub)
        throw std::runtime_error("Loaded index is not of type IndexIVFPQ");       // the reference's message for an unusable cache
    // the file is not trusted: the vector count sizes two allocations and the list offsets index into them
    if (hdr[2] > nb) throw std::runtime_error("Loaded index is not of type IndexIVFPQ");       // more stored vectors than base rows
    std::vector<float> cent(hdr[0]), books(hdr[1]);
    std::vector<uint64_t> off(NLIST + 1); std::vector<int64_t> ids(hdr[2]); std::vector<uint8_t> codes(hdr[2] * kM);
    f.read(reinterpret_cast<char *>(cent.data()), cent.size() * 4);
    f.read(reinterpret_cast<char *>(books.data()), books.size() * 4);
    f.read(reinterpret_cast<char *>(off.data()), off.size() * 8);
    f.read(reinterpret_cast<char *>(ids.data()), ids.size() * 8);
    f.read(reinterpret_cast<char *>(codes.data()), codes.size());
    if (!f) throw std::runtime_error("index cache is truncated");
    bool sane = off[0] == 0 && off[NLIST] == hdr[2];
    for (size_t l = 0; sane && l < static_cast<size_t>(NLIST); ++l) sane = off[l] <= off[l + 1];
    for (size_t i = 0; sane && i < ids.size(); ++i) sane = ids[i] >= 0 && static_cast<uint64_t>(ids[i]) < nb;
    if (!sane) throw std::runtime_error("Loaded index is not of type IndexIVFPQ");       // inconsistent offsets or labels outside the base
    std::lock_guard<std::mutex> g(m_Impl->lock);
    m_Impl->install(0, base.data(), nb, std::move(cent), std::move(books));
    std::vector<int64_t> list(ids.size());
    for (size_t l = 0; l < static_cast<size_t>(NLIST); ++l) std::fill(list.begin() + off[l], list.begin() + off[l + 1], static_cast<int64_t>(l));
    check(pf_ivfpq_add_encoded(m_Impl->ivfpq, ids.size(), list.data(), codes.data(), ids.data()), "pf_ivfpq_add_encoded");
}

void Server::export_index(std::vector<float> &centroids, std::vector<float> &codebooks, std::vector<uint8_t> &codes,
                          std::vector<faiss::idx_t> &ids, std::vector<uint64_t> &list_offsets) const {
    const Impl &im = *m_Impl;
    im.require_ready();
    centroids = im.h_centroids; codebooks = im.h_codebooks;
    std::vector<uint64_t> sizes(im.nlist);
    size_t ntotal = 0;
    check(pf_ivfpq_info(im.ivfpq, nullptr, nullptr, nullptr, &ntotal, sizes.data()), "pf_ivfpq_info");
    list_offsets.assign(im.nlist + 1, 0);
    for (size_t l = 0; l < im.nlist; ++l) list_offsets[l + 1] = list_offsets[l] + sizes[l];
    codes.resize(ntotal * kM); ids.resize(ntotal);
    for (size_t l = 0; l < im.nlist; ++l)
        check(pf_ivfpq_get_list(im.ivfpq, static_cast<uint32_t>(l), codes.data() + list_offsets[l] * kM, ids.data() + list_offsets[l]), "pf_ivfpq_get_list");
}

// Reference: drogon::app().addListener(SERVER_ADDRESS, SERVER_PORT); run() (/root/reference/src/server/server_lib.cpp:48-53).
// Here the four routes (and the encrypted one) are served by the POSIX-socket listener of include/server/http.h; blocks
// for the life of the process like Drogon's run().  PREFHETCH_LISTEN_PORT overrides the port.
void Server::run_webserver() {
    int port = kListenPort;
    if (const char *e = std::getenv("PREFHETCH_LISTEN_PORT")) port = std::atoi(e);
    wire::HttpListener listener(*this, kListenAddress, static_cast<uint16_t>(port));
    listener.serve();
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

void Server::coarseSearch(const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &precise_query,
                          const std::array<std::array<faiss::idx_t, NPROBE>, NQUERY> &nearest_centroid_idx,
                          std::vector<float> &coarse_distance_scores, std::vector<faiss::idx_t> &coarse_distance_indexes,
                          std::array<size_t, NQUERY> &list_sizes_per_query) const {
    // reference src/server/server_lib.cpp:111-138: search_encrypted over the client-chosen lists, outputs sized
    // NBASE * NQUERY first and trimmed to the number of scanned vectors afterwards
    const Impl &im = *m_Impl;
    std::lock_guard<std::mutex> g(im.lock);
    im.require_ready();
    const size_t cap = static_cast<size_t>(NBASE) * NQUERY > im.nb * NQUERY ? static_cast<size_t>(NBASE) * NQUERY : im.nb * NQUERY;
    im.d_query.reserve(im.device, sizeof precise_query);
    im.d_out.reserve(im.device, cap * sizeof(float));
    im.d_out2.reserve(im.device, cap * sizeof(int64_t));
    check(pf_memcpy_h2d(im.device, im.d_query.ptr, precise_query.data(), sizeof precise_query, nullptr), "h2d");
    uint64_t sizes[NQUERY];
    check(pf_ivfpq_search_lists(im.ivfpq, static_cast<const float *>(im.d_query.ptr), nearest_centroid_idx.data()->data(), NQUERY, NPROBE,
                                static_cast<float *>(im.d_out.ptr), static_cast<int64_t *>(im.d_out2.ptr), cap, sizes, nullptr), "pf_ivfpq_search_lists");
    size_t total = 0;
    for (int i = 0; i < NQUERY; ++i) { list_sizes_per_query[i] = sizes[i]; total += sizes[i]; }
    coarse_distance_scores.resize(total);
    coarse_distance_indexes.resize(total);
    if (total) {
        check(pf_memcpy_d2h(im.device, coarse_distance_scores.data(), im.d_out.ptr, total * sizeof(float), nullptr), "d2h");
        check(pf_memcpy_d2h(im.device, coarse_distance_indexes.data(), im.d_out2.ptr, total * sizeof(int64_t), nullptr), "d2h");
    }
    check(pf_stream_synchronize(im.device, nullptr), "sync");
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

// The encrypted form of preciseSearch.  Candidate rows are packed 64 to a plaintext polynomial (as pf_pack_rows does), brought
// to NTT form and multiplied into the query ciphertext (pf_ct_rows_mul: all of it in registers): ENC_POLYS_PER_QUERY = ceil(COARSE_PROBE * 128 / N) = 4
// ciphertext x plaintext products per query, one fused launch for all of them.
void Server::preciseSearchEncrypted(const uint64_t *query_ct_device, const std::array<std::array<faiss::idx_t, COARSE_PROBE>, NQUERY> &ids,
                                    uint64_t *result_ct_device) const {
    const Impl &im = *m_Impl;
    std::lock_guard<std::mutex> g(im.lock);
    im.require_ready();
    if (!query_ct_device || !result_ct_device) throw std::invalid_argument("preciseSearchEncrypted: null device pointer");
    constexpr size_t polys = static_cast<size_t>(NQUERY) * ENC_POLYS_PER_QUERY;
    std::vector<int64_t> padded(polys * ENC_ROWS_PER_POLY, -1);                 // -1: zero row
    for (size_t q = 0; q < static_cast<size_t>(NQUERY); ++q)
        for (size_t j = 0; j < static_cast<size_t>(COARSE_PROBE); ++j) padded[q * ENC_POLYS_PER_QUERY * ENC_ROWS_PER_POLY + j] = ids[q][j];
    im.d_ids.reserve(im.device, padded.size() * 8);
    check(pf_memcpy_h2d(im.device, im.d_ids.ptr, padded.data(), padded.size() * 8, nullptr), "h2d");
    // the query ciphertexts are transformed once (into a scratch buffer), not once per plaintext block
    constexpr size_t ct_words = static_cast<size_t>(NQUERY) * 2 * ENC_LIMBS * ENC_RING_DEGREE;
    im.d_query.reserve(im.device, ct_words * 8);
    uint64_t *qn = static_cast<uint64_t *>(im.d_query.ptr);
    check(pf_ntt_forward_to(im.ring, query_ct_device, qn, static_cast<size_t>(NQUERY) * 2 * ENC_LIMBS, nullptr), "pf_ntt_forward_to");
    // plaintexts: packed from the candidate rows, transformed and multiplied into both ciphertext components in one kernel
    check(pf_ct_rows_mul(im.ring, qn, im.base, static_cast<const int64_t *>(im.d_ids.ptr), polys, ENC_ROWS_PER_POLY, ENC_POLYS_PER_QUERY,
                         result_ct_device, nullptr), "pf_ct_rows_mul");
    check(pf_stream_synchronize(im.device, nullptr), "sync");
}

void Server::preciseSearchEncryptedHost(const uint64_t *query_ct_host, const std::array<std::array<faiss::idx_t, COARSE_PROBE>, NQUERY> &ids,
                                        uint64_t *result_ct_host, std::array<std::array<float, COARSE_PROBE>, NQUERY> &row_norms) const {
    if (!query_ct_host || !result_ct_host) throw std::invalid_argument("preciseSearchEncryptedHost: null buffer");
    constexpr size_t ct_words = static_cast<size_t>(NQUERY) * 2 * ENC_LIMBS * ENC_RING_DEGREE;
    constexpr size_t out_words = ct_words * ENC_POLYS_PER_QUERY;
    int dev = 0;
    {
        std::lock_guard<std::mutex> g(m_Impl->lock);
        m_Impl->require_ready();
        dev = m_Impl->device;
    }
    DevBuf d_in, d_res;                              // per call: the device form below takes the lock itself
    d_in.reserve(dev, ct_words * 8);
    d_res.reserve(dev, out_words * 8);
    check(pf_memcpy_h2d(dev, d_in.ptr, query_ct_host, ct_words * 8, nullptr), "h2d");
    check(pf_stream_synchronize(dev, nullptr), "sync");
    preciseSearchEncrypted(static_cast<const uint64_t *>(d_in.ptr), ids, static_cast<uint64_t *>(d_res.ptr));
    check(pf_memcpy_d2h(dev, result_ct_host, d_res.ptr, out_words * 8, nullptr), "d2h");
    check(pf_stream_synchronize(dev, nullptr), "sync");
    // ||x||^2 = the reference's distance chain against the zero vector (server_lib.cpp:151-162): same accumulation
    const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> zero{};
    preciseSearch(zero, ids, row_norms);
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

size_t Server::pirRows() const {
    std::lock_guard<std::mutex> g(m_Impl->lock);
    m_Impl->require_ready();
    return m_Impl->nb;
}

uint32_t Server::pirLevels() const {
    std::lock_guard<std::mutex> g(m_Impl->lock);
    m_Impl->require_ready();
    return pir::Layout::make(ENC_RING_DEGREE, static_cast<uint32_t>(PRECISE_VECTOR_DIMENSIONS), m_Impl->nb).levels;
}

size_t Server::pirCols() const {
    std::lock_guard<std::mutex> g(m_Impl->lock);
    m_Impl->require_ready();
    return pir::Layout::make(ENC_RING_DEGREE, static_cast<uint32_t>(PRECISE_VECTOR_DIMENSIONS), m_Impl->nb).n_cols;
}

void Server::preciseVectorPIRPrivateHost(const uint64_t *query_ct_host, size_t count, const uint64_t *galois_keys_host, uint64_t *reply_ct_host) const {
    if (count == 0) return;
    if (!query_ct_host || !galois_keys_host || !reply_ct_host) throw std::invalid_argument("preciseVectorPIRPrivateHost: null buffer");
    const Impl &im = *m_Impl;
    std::lock_guard<std::mutex> g(im.lock);
    im.require_ready();
    PirState &ps = im.pir_state();
    const size_t N = ENC_RING_DEGREE, D = ENC_LIMBS, per = 2 * D * N, key_words = D * 2 * (D + 1) * N;
    const uint32_t levels = ps.db.layout().levels;
    const std::vector<uint32_t> elts = pir::galois_elements(ENC_RING_DEGREE, levels);
    std::vector<bfv::SwitchKey> keys(levels);
    for (uint32_t j = 0; j < levels; ++j) {
        keys[j].ksk = bfv::DeviceWords(im.device, key_words);
        keys[j].ksk.upload(galois_keys_host + j * key_words, key_words);
        keys[j].galois_elt = elts[j];
    }
    bfv::Ciphertexts query, reply;
    query.count = count;
    query.data = bfv::DeviceWords(im.device, count * per);
    query.data.upload(query_ct_host, count * per);
    pir::answer(ps.ctx, ps.db, query, keys, reply);
    reply.data.download(reply_ct_host, reply.count * per);                             // count x n_cols ciphertexts
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


// caller holds the lock
PirState &Server::Impl::pir_state() const {
    const Impl &im = *this;
    if (!im.pir) {
        // the base lives in HBM (pf_flat): one gather brings the rows back for packing
        std::vector<int64_t> ids(im.nb);
        for (size_t i = 0; i < im.nb; ++i) ids[i] = static_cast<int64_t>(i);
        std::vector<float> rows(im.nb * kD);
        DevBuf d_ids, d_rows;
        d_ids.reserve(im.device, ids.size() * 8);
        d_rows.reserve(im.device, rows.size() * sizeof(float));
        check(pf_memcpy_h2d(im.device, d_ids.ptr, ids.data(), ids.size() * 8, nullptr), "h2d");
        check(pf_gather_rows(im.base, static_cast<const int64_t *>(d_ids.ptr), im.nb, static_cast<float *>(d_rows.ptr), nullptr), "pf_gather_rows");
        check(pf_memcpy_d2h(im.device, rows.data(), d_rows.ptr, rows.size() * sizeof(float), nullptr), "d2h");
        check(pf_stream_synchronize(im.device, nullptr), "sync");
        im.pir = std::make_unique<PirState>(im.device, rows.data(), im.nb);
    }
    return *im.pir;
}
