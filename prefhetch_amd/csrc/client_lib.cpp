// client_lib.cpp -- client side of the PreFHEtch protocol (include/client/client_lib.h), restating
// /root/reference/src/client/client_lib.cpp:16-337 over a wire::Transport and the C ABI of libprefhetch_hip.so.
#include "../../include/client/client_lib.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <stdexcept>

#include "../../include/client/bfv.h"
#include "../../include/client/pir.h"
#include "../../include/prefhetch_hip.h"
#include "../../include/server/wire.h"

namespace {

const char *const kQueryPath = "../sift/siftsmall/siftsmall_query.fvecs";                  // reference :12
const char *const kGroundTruthPath = "../sift/siftsmall/siftsmall_groundtruth.ivecs";      // reference :13-14

wire::Transport *g_transport = nullptr;
int g_device = 0;

wire::Transport &transport() {
    if (!g_transport) throw std::runtime_error("client: no transport set (set_transport)");
    return *g_transport;
}

void check(pf_status st, const char *what) {
    if (st != PF_OK) throw std::runtime_error(std::string(what) + ": " + pf_status_str(st) + " (" + pf_last_error() + ")");
}

void sort_by_distance(DistanceIndexData *b, DistanceIndexData *e) {
    std::stable_sort(b, e, [](const DistanceIndexData &a, const DistanceIndexData &c) { return a.distance < c.distance; });
}

template <size_t ROWS, size_t COLS>
void append_float_matrix(std::string &out, const std::array<std::array<float, COLS>, ROWS> &m) {
    out.push_back('[');
    for (size_t r = 0; r < ROWS; ++r) {
        if (r) out.push_back(',');
        out.push_back('[');
        for (size_t c = 0; c < COLS; ++c) {
            if (c) out.push_back(',');
            wire::append_float(out, m[r][c]);
        }
        out.push_back(']');
    }
    out.push_back(']');
}

template <size_t ROWS, size_t COLS>
void append_id_matrix(std::string &out, const std::array<std::array<faiss_idx_t, COLS>, ROWS> &m) {
    out.push_back('[');
    for (size_t r = 0; r < ROWS; ++r) {
        if (r) out.push_back(',');
        out.push_back('[');
        for (size_t c = 0; c < COLS; ++c) {
            if (c) out.push_back(',');
            wire::append_int(out, m[r][c]);
        }
        out.push_back(']');
    }
    out.push_back(']');
}

struct DevBuf {
    void *p = nullptr;
    DevBuf(size_t bytes) { check(pf_malloc(g_device, &p, bytes), "pf_malloc"); }
    ~DevBuf() { if (p) pf_free(g_device, p); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

}  // namespace

void set_transport(wire::Transport *t) { g_transport = t; }
void set_client_device(int device) { g_device = device; }

void ping_server() { (void)transport().get("query"); }

void get_query(std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &query) {
    size_t nq = 0, d2 = 0;
    std::vector<float> xq;
    vecs_read<float>(kQueryPath, d2, nq, xq);
    if (d2 != (size_t)PRECISE_VECTOR_DIMENSIONS || nq < (size_t)NQUERY) {      // assert()s in the reference (:24-27)
        std::fprintf(stderr, "%s: %zu queries of dimension %zu, need %lld of %lld\n", kQueryPath, nq, d2, (long long)NQUERY,
                     (long long)PRECISE_VECTOR_DIMENSIONS);
        std::abort();
    }
    for (size_t i = 0; i < (size_t)NQUERY; ++i)
        for (size_t j = 0; j < (size_t)PRECISE_VECTOR_DIMENSIONS; ++j) query[i][j] = xq[i * PRECISE_VECTOR_DIMENSIONS + j];
}

void get_centroids(std::vector<std::array<float, PRECISE_VECTOR_DIMENSIONS>> &centroids) {
    const wire::Json resp = wire::parse(transport().get("query"));
    if (resp.kind != wire::Json::Array) throw wire::TypeError("centroid response is not an array");
    centroids.resize(resp.arr.size());
    for (size_t i = 0; i < resp.arr.size(); ++i)
        for (size_t k = 0; k < (size_t)PRECISE_VECTOR_DIMENSIONS; ++k) centroids[i][k] = resp.arr[i].at(k).as_float();
}

// Reference :50-81: distance[i][j] = sum_k pow(query[i][k] - centroid[j][k], 2) accumulated in a float through
// double, every centroid kept, ascending by distance.  The distances come from pf_l2_gathered (same chain, bit for
// bit) over an index built from the centroids; the sort stays on the host.
void sort_nearest_centroids(const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &precise_query,
                            const std::vector<std::array<float, PRECISE_VECTOR_DIMENSIONS>> &centroids,
                            std::array<std::vector<DistanceIndexData>, NQUERY> &nearest_centroids) {
    const size_t nc = centroids.size();
    for (auto &v : nearest_centroids) v.clear();
    if (nc == 0) return;
    pf_flat *index = nullptr;
    check(pf_flat_create(&index, g_device, centroids[0].data(), nc, (uint32_t)PRECISE_VECTOR_DIMENSIONS), "pf_flat_create");
    try {
        std::vector<int64_t> ids((size_t)NQUERY * nc);
        for (size_t i = 0; i < (size_t)NQUERY; ++i)
            for (size_t j = 0; j < nc; ++j) ids[i * nc + j] = (int64_t)j;
        std::vector<float> dist((size_t)NQUERY * nc);
        DevBuf dq(sizeof precise_query), di(ids.size() * 8), dd(dist.size() * 4);
        check(pf_memcpy_h2d(g_device, dq.p, precise_query.data(), sizeof precise_query, nullptr), "h2d");
        check(pf_memcpy_h2d(g_device, di.p, ids.data(), ids.size() * 8, nullptr), "h2d");
        check(pf_l2_gathered(index, static_cast<const float *>(dq.p), static_cast<const int64_t *>(di.p), (size_t)NQUERY, (uint32_t)nc,
                             static_cast<float *>(dd.p), nullptr), "pf_l2_gathered");
        check(pf_memcpy_d2h(g_device, dist.data(), dd.p, dist.size() * 4, nullptr), "d2h");
        check(pf_stream_synchronize(g_device, nullptr), "sync");
        for (size_t i = 0; i < (size_t)NQUERY; ++i) {
            nearest_centroids[i].reserve(nc);
            for (size_t j = 0; j < nc; ++j) nearest_centroids[i].push_back(DistanceIndexData{dist[i * nc + j], (faiss_idx_t)j});
            sort_by_distance(nearest_centroids[i].data(), nearest_centroids[i].data() + nc);
        }
    } catch (...) {
        pf_flat_destroy(index);
        throw;
    }
    pf_flat_destroy(index);
}

void get_coarse_scores(const std::array<std::vector<DistanceIndexData>, NQUERY> &sorted_centroids,
                       const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &precise_query,
                       std::vector<float> &coarse_scores, std::vector<faiss_idx_t> &coarse_vectors_idx,
                       std::array<size_t, NQUERY> &list_sizes_per_query) {
    std::array<std::array<faiss_idx_t, NPROBE>, NQUERY> nearest_centroids_id;
    for (size_t i = 0; i < (size_t)NQUERY; ++i) {
        if ((size_t)NPROBE > sorted_centroids[i].size()) throw std::runtime_error("Centroids count is not equal to NPROBE");   // reference :97-100
        for (size_t j = 0; j < (size_t)NPROBE; ++j) nearest_centroids_id[i][j] = sorted_centroids[i][j].idx;
    }
    std::string body = "{\"nearestCentroidIndexes\":";          // nlohmann dumps object keys in sorted order
    append_id_matrix(body, nearest_centroids_id);
    body += ",\"preciseQuery\":";
    append_float_matrix(body, precise_query);
    body.push_back('}');

    const wire::Json resp = wire::parse(transport().post("coarsesearch", body));
    const wire::Json &scores = resp.at("coarseDistanceScores"), &idx = resp.at("coarseVectorIndexes"), &sizes = resp.at("listSizesPerQuery");
    if (scores.kind != wire::Json::Array || idx.kind != wire::Json::Array) throw wire::TypeError("coarse response fields must be arrays");
    coarse_scores.resize(scores.arr.size());
    for (size_t i = 0; i < scores.arr.size(); ++i) coarse_scores[i] = scores.arr[i].as_float();
    coarse_vectors_idx.resize(idx.arr.size());
    for (size_t i = 0; i < idx.arr.size(); ++i) coarse_vectors_idx[i] = idx.arr[i].as_int();
    for (size_t i = 0; i < (size_t)NQUERY; ++i) list_sizes_per_query[i] = (size_t)sizes.at(i).as_int();
}

void compute_nearest_coarse_vectors(const std::vector<float> &coarse_distance_scores, const std::vector<faiss_idx_t> &coarse_vector_indexes,
                                    const std::array<size_t, NQUERY> &list_sizes_per_query,
                                    std::array<std::vector<DistanceIndexData>, NQUERY> &nearest_coarse_vectors) {
    size_t current = 0;
    for (size_t i = 0; i < (size_t)NQUERY; ++i) {
        if (list_sizes_per_query[i] < (size_t)COARSE_PROBE)
            throw std::runtime_error("Number of computed coarse scores is lesser than COARSE_PROBE");          // reference :131-137
        if (current + list_sizes_per_query[i] > coarse_distance_scores.size() || current + list_sizes_per_query[i] > coarse_vector_indexes.size())
            throw std::out_of_range("list sizes exceed the number of coarse scores");     // the reference would read past the end
        nearest_coarse_vectors[i].clear();
        nearest_coarse_vectors[i].reserve(list_sizes_per_query[i]);
        for (size_t j = 0; j < list_sizes_per_query[i]; ++j)
            nearest_coarse_vectors[i].push_back(DistanceIndexData{coarse_distance_scores[current + j], coarse_vector_indexes[current + j]});
        current += list_sizes_per_query[i];
        sort_by_distance(nearest_coarse_vectors[i].data(), nearest_coarse_vectors[i].data() + nearest_coarse_vectors[i].size());
    }
}

void get_precise_scores(const std::array<std::vector<DistanceIndexData>, NQUERY> &sorted_coarse_vectors,
                        const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &precise_query,
                        std::array<std::array<float, COARSE_PROBE>, NQUERY> &precise_scores) {
    std::array<std::array<faiss_idx_t, COARSE_PROBE>, NQUERY> ids;
    for (size_t i = 0; i < (size_t)NQUERY; ++i) {
        if (sorted_coarse_vectors[i].size() < (size_t)COARSE_PROBE) throw std::out_of_range("fewer than COARSE_PROBE coarse candidates");
        for (size_t j = 0; j < (size_t)COARSE_PROBE; ++j) ids[i][j] = sorted_coarse_vectors[i][j].idx;
    }
    std::string body = "{\"nearestCoarseVectorIndexes\":";
    append_id_matrix(body, ids);
    body += ",\"preciseQuery\":";
    append_float_matrix(body, precise_query);
    body.push_back('}');
    const wire::Json resp = wire::parse(transport().post("precisesearch", body));
    const wire::Json &scores = resp.at("preciseDistanceScores");
    for (size_t i = 0; i < (size_t)NQUERY; ++i)
        for (size_t j = 0; j < (size_t)COARSE_PROBE; ++j) precise_scores[i][j] = scores.at(i).at(j).as_float();
}

void get_precise_scores_encrypted(const RankedLists &sorted_coarse_vectors, const QueryBatch &precise_query, const bfv::Context &ctx,
                                  bfv::Encryptor &encryptor, bfv::Decryptor &decryptor, PreciseScores &precise_scores) {
    constexpr size_t D = (size_t)PRECISE_VECTOR_DIMENSIONS, NQ = (size_t)NQUERY, C = (size_t)COARSE_PROBE;
    const size_t N = ctx.N(), L = ctx.L();
    if (N != Server::ENC_RING_DEGREE || L != Server::ENC_LIMBS) throw std::invalid_argument("encrypted round: ring degree 8192 and 4 primes expected");
    for (size_t l = 0; l < L; ++l)
        if (ctx.params().moduli[l] != Server::ENC_MODULI[l]) throw std::invalid_argument("encrypted round: SEAL BFVDefault(8192) data primes expected");
    std::array<std::array<faiss_idx_t, COARSE_PROBE>, NQUERY> ids;
    for (size_t i = 0; i < NQ; ++i) {
        if (sorted_coarse_vectors[i].size() < C) throw std::out_of_range("fewer than COARSE_PROBE coarse candidates");
        for (size_t j = 0; j < C; ++j) ids[i][j] = sorted_coarse_vectors[i][j].idx;
    }
    // encode and encrypt
    std::vector<uint64_t> plain(NQ * N);
    for (size_t i = 0; i < NQ; ++i) bfv::encode_query(precise_query[i].data(), (uint32_t)D, (uint32_t)N, ctx.t(), plain.data() + i * N);
    bfv::Ciphertexts qct;
    encryptor.encrypt(plain.data(), NQ, qct);
    std::vector<uint64_t> words(NQ * 2 * L * N);
    qct.data.download(words.data(), words.size());
    std::string body = "{\"nearestCoarseVectorIndexes\":";
    append_id_matrix(body, ids);
    body += ",\"queryCiphertexts\":\"";
    body += wire::base64_encode(words.data(), words.size() * 8);
    body += "\"}";
    const wire::Json resp = wire::parse(transport().post("precisesearch-encrypted", body));
    // decrypt the ENC_POLYS_PER_QUERY result ciphertexts of every query
    const wire::Json &blob = resp.at("resultCiphertexts");
    if (blob.kind != wire::Json::String) throw wire::TypeError("resultCiphertexts must be a base64 string");
    const std::vector<uint8_t> raw = wire::base64_decode(blob.s);
    const size_t blocks = NQ * Server::ENC_POLYS_PER_QUERY;
    if (raw.size() != blocks * 2 * L * N * 8) throw std::out_of_range("resultCiphertexts has the wrong size");
    bfv::Ciphertexts rct;
    rct.count = blocks;
    rct.data = bfv::DeviceWords(ctx.params().device, blocks * 2 * L * N);
    rct.data.upload(reinterpret_cast<const uint64_t *>(raw.data()), blocks * 2 * L * N);
    std::vector<uint64_t> rplain;
    decryptor.decrypt(rct, rplain);
    const wire::Json &norms = resp.at("rowNorms");
    for (size_t i = 0; i < NQ; ++i) {
        double qn = 0;
        for (size_t k = 0; k < D; ++k) { const double v = (double)llrintf(precise_query[i][k]); qn += v * v; }
        for (size_t b = 0; b < Server::ENC_POLYS_PER_QUERY; ++b) {
            int64_t ip[Server::ENC_ROWS_PER_POLY];
            bfv::decode_inner_products(rplain.data() + (i * Server::ENC_POLYS_PER_QUERY + b) * N, (uint32_t)D, Server::ENC_ROWS_PER_POLY, ctx.t(), ip);
            for (size_t j = 0; j < Server::ENC_ROWS_PER_POLY; ++j) {
                const size_t cand = b * Server::ENC_ROWS_PER_POLY + j;
                if (cand >= C) break;
                precise_scores[i][cand] = static_cast<float>(qn - 2.0 * (double)ip[j] + (double)norms.at(i).at(cand).as_float());
            }
        }
    }
}

void compute_nearest_precise_vectors(const std::array<std::array<float, COARSE_PROBE>, NQUERY> &precise_scores,
                                     const std::array<std::vector<DistanceIndexData>, NQUERY> &sorted_coarse_vectors,
                                     std::array<std::array<DistanceIndexData, COARSE_PROBE>, NQUERY> &nearest_precise_vectors) {
    for (size_t i = 0; i < (size_t)NQUERY; ++i) {
        if (sorted_coarse_vectors[i].size() < (size_t)COARSE_PROBE) throw std::out_of_range("fewer than COARSE_PROBE coarse candidates");
        for (size_t j = 0; j < (size_t)COARSE_PROBE; ++j)
            nearest_precise_vectors[i][j] = DistanceIndexData{precise_scores[i][j], sorted_coarse_vectors[i][j].idx};
        sort_by_distance(nearest_precise_vectors[i].data(), nearest_precise_vectors[i].data() + COARSE_PROBE);
    }
}

void get_precise_vectors_pir(const std::array<std::array<DistanceIndexData, COARSE_PROBE>, NQUERY> &nearest_precise_vectors,
                             std::array<std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, K>, NQUERY> &query_results,
                             std::array<std::array<faiss_idx_t, K>, NQUERY> &query_results_idx) {
    static_assert(K <= COARSE_PROBE, "K greater than COARSE_PROBE");               // a run-time throw in the reference (:214-217)
    for (size_t i = 0; i < (size_t)NQUERY; ++i)
        for (size_t j = 0; j < (size_t)K; ++j) query_results_idx[i][j] = nearest_precise_vectors[i][j].idx;
    std::string body = "{\"nearestPreciseVectorIndexes\":";
    append_id_matrix(body, query_results_idx);
    body.push_back('}');
    const wire::Json resp = wire::parse(transport().post("precise-vector-pir", body));
    const wire::Json &rows = resp.at("queryResults");
    for (size_t i = 0; i < (size_t)NQUERY; ++i)
        for (size_t j = 0; j < (size_t)K; ++j)
            for (size_t k = 0; k < (size_t)PRECISE_VECTOR_DIMENSIONS; ++k) query_results[i][j][k] = rows.at(i).at(j).at(k).as_float();
}

void get_precise_vectors_pir_private(const PreciseRanking &nearest_precise_vectors, const bfv::Context &ctx, bfv::KeyGenerator &keygen,
                                     bfv::Encryptor &encryptor, bfv::Decryptor &decryptor, ResultVectors &query_results,
                                     ResultIds &query_results_idx, size_t results_per_query) {
    constexpr size_t D = (size_t)PRECISE_VECTOR_DIMENSIONS, NQ = (size_t)NQUERY;
    if (results_per_query > (size_t)K) throw std::invalid_argument("private retrieval: at most K rows per query");
    const wire::Json lay_json = wire::parse(transport().post("pir-layout", "{}"));
    const pir::Layout lay = pir::Layout::make(ctx.N(), (uint32_t)D, (size_t)lay_json.at("rows").as_int());
    if ((int64_t)lay.levels != lay_json.at("levels").as_int() || (int64_t)lay.n_cols != lay_json.at("cols").as_int() || (int64_t)ctx.N() != lay_json.at("ringDegree").as_int() ||
        (int64_t)ctx.t() != lay_json.at("plainModulus").as_int())
        throw std::invalid_argument("private retrieval: the server packs its rows for other parameters");
    const size_t N = ctx.N(), L = ctx.L(), per = 2 * L * N, key_words = L * 2 * (L + 1) * N;
    for (size_t i = 0; i < NQ; ++i)
        for (size_t j = 0; j < (size_t)K; ++j) query_results_idx[i][j] = nearest_precise_vectors[i][j].idx;
    // Galois keys of the expansion rounds, in pf_key_switch's layout
    std::vector<uint64_t> keys((size_t)lay.levels * key_words);
    {
        size_t j = 0;
        for (uint32_t g : pir::galois_elements((uint32_t)N, lay.levels)) {
            const bfv::SwitchKey gk = keygen.create_galois_key(g);
            gk.ksk.download(keys.data() + j++ * key_words, key_words);
        }
    }
    bool keys_sent = false;
    std::vector<uint64_t> plain(results_per_query * N), words(results_per_query * per), back;
    for (size_t i = 0; i < NQ; ++i) {                                                 // one request per query
        if (results_per_query == 0) break;
        for (size_t j = 0; j < results_per_query; ++j) pir::encode_query(lay, ctx.t(), (size_t)query_results_idx[i][j], plain.data() + j * N);
        bfv::Ciphertexts qct;
        encryptor.encrypt(plain.data(), results_per_query, qct);
        qct.data.download(words.data(), words.size());
        std::string body = "{\"count\":" + std::to_string(results_per_query) + ",\"queryCiphertexts\":\"" + wire::base64_encode(words.data(), words.size() * 8) + "\"";
        if (!keys_sent) { body += ",\"galoisKeys\":\"" + wire::base64_encode(keys.data(), keys.size() * 8) + "\""; keys_sent = true; }
        body.push_back('}');
        const wire::Json resp = wire::parse(transport().post("precise-vector-pir-private", body));
        const wire::Json &blob = resp.at("replyCiphertexts");
        if (blob.kind != wire::Json::String) throw wire::TypeError("replyCiphertexts must be a base64 string");
        const std::vector<uint8_t> raw = wire::base64_decode(blob.s);
        const size_t n_reply = results_per_query * lay.n_cols;                         // a retrieval's columns together
        if (raw.size() != n_reply * per * 8) throw std::out_of_range("replyCiphertexts has the wrong size");
        bfv::Ciphertexts rct;
        rct.count = n_reply;
        rct.data = bfv::DeviceWords(ctx.params().device, n_reply * per);
        rct.data.upload(reinterpret_cast<const uint64_t *>(raw.data()), n_reply * per);
        decryptor.decrypt(rct, back);
        for (size_t j = 0; j < results_per_query; ++j)
            pir::decode_row(lay, back.data() + j * lay.n_cols * N, (size_t)query_results_idx[i][j], query_results[i][j].data());
    }
}

RecallStats compute_recall_stats(const std::array<std::array<faiss_idx_t, K>, NQUERY> &observed, const std::vector<int> &ground_truth,
                                 size_t gt_nn_per_query) {
    if ((size_t)K > gt_nn_per_query)
        throw std::runtime_error("K greater than nearest neigbours per query in ground truth dataset");       // reference :261-267
    if (ground_truth.size() < (size_t)NQUERY * gt_nn_per_query) throw std::out_of_range("ground truth holds fewer than NQUERY rows");
    float mrr_1 = 0, mrr_10 = 0, mrr_100 = 0;
    int nq_recall_1 = 0, nq_recall_10 = 0, nq_recall_100 = 0;
    for (size_t i = 0; i < (size_t)NQUERY; ++i) {
        for (size_t j = 0; j < (size_t)K; ++j) {
            for (size_t k = 0; k < (size_t)K; ++k) {
                if ((faiss_idx_t)ground_truth[i * gt_nn_per_query + j] != observed[i][k]) continue;
                if (k < 1) nq_recall_1++;
                if (k < 10) nq_recall_10++;
                if (k < 100) nq_recall_100++;
                if (j == 0) {                                   // MRR looks at the first ground-truth neighbour only
                    const float rr = 1.0f / static_cast<float>(k + 1);
                    if (k < 1) mrr_1 += rr;
                    if (k < 10) mrr_10 += rr;
                    if (k < 100) mrr_100 += rr;
                }
                break;
            }
        }
    }
    RecallStats s;
    s.recall_1 = static_cast<float>(nq_recall_1) / (1 * NQUERY);
    s.recall_10 = static_cast<float>(nq_recall_10) / (10 * NQUERY);
    s.recall_100 = static_cast<float>(nq_recall_100) / (100 * NQUERY);
    s.mrr_1 = mrr_1 / NQUERY;
    s.mrr_10 = mrr_10 / NQUERY;
    s.mrr_100 = mrr_100 / NQUERY;
    return s;
}

void benchmark_results(const std::array<std::array<faiss_idx_t, K>, NQUERY> &observed_query_results_idx) {
    size_t gt_nn_per_query = 0, gt_nq = 0;
    std::vector<int> ground_truth;
    vecs_read(kGroundTruthPath, gt_nn_per_query, gt_nq, ground_truth);
    const RecallStats s = compute_recall_stats(observed_query_results_idx, ground_truth, gt_nn_per_query);
    std::printf("\n\nBENCHMARK RESULTS\nTotal Query Benchmark Results\n");
    std::printf("Parameters: NPROBE = %lld, COARSE_PROBE = %lld, K = %lld\n", (long long)NPROBE, (long long)COARSE_PROBE, (long long)K);
    std::printf("Parameters: NQUERY = %lld, NLIST = %lld\n", (long long)NQUERY, (long long)NLIST);
    std::printf("Parameters: SUB_QUANTIZERS = %lld, SUB_VECTOR_SIZE = %lld\n", (long long)SUB_QUANTIZERS, (long long)SUB_QUANTIZER_SIZE);
    std::printf("Recall@1 = %g, Recall@10 = %g, Recall@100 = %g\n", s.recall_1, s.recall_10, s.recall_100);
    std::printf("MRR@1 = %g, MRR@10 = %g, MRR@100 = %g\n\n\n", s.mrr_1, s.mrr_10, s.mrr_100);
    for (int i = 0; i < 100; ++i) std::printf("-");
    std::printf("\n");
}
