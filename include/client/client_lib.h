// client_lib.h -- the client side of the PreFHEtch protocol, source-compatible with the reference header of the same
// path (/root/reference/include/client/client_lib.h:1-72: same free functions, argument meaning and error behaviour;
// bodies restated from /root/reference/src/client/client_lib.cpp:16-337).
//
// What differs from the reference:
//   * requests go through a wire::Transport (include/server/wire.h) chosen with set_transport() instead of cpr over
//     "http://localhost:8080/": the JSON bodies are the reference's, so an HTTP transport is a ten-line adapter;
//   * the L2 shortlist over the centroids (sort_nearest_centroids) is evaluated by the HIP kernel
//     that reproduces the reference's float += pow(diff, 2) chain bit for bit (pf_l2_gathered), not by a host loop;
//   * every sort is a stable sort by (distance, position), where the reference's std::ranges::sort leaves the
//     order of equal distances unspecified;
//   * the recall / MRR figures are also returned (RecallStats) and can be computed on in-memory ground truth.
#pragma once

#include <array>
#include <string>
#include <vector>

#include "client_server_utils.h"

namespace wire { struct Transport; }
namespace bfv { class Context; class Encryptor; class Decryptor; class KeyGenerator; }

const std::string server_addr = "http://localhost:8080/";     // reference :7 (used by HTTP transports)

struct DistanceIndexData {
    float distance;
    faiss_idx_t idx;
};

// Shapes of the protocol (aliases only: the types are the reference's std::array / std::vector types, so its callers
// compile unchanged)
using QueryBatch = std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY>;         // NQUERY x 128 query vectors
using CentroidTable = std::vector<std::array<float, PRECISE_VECTOR_DIMENSIONS>>;             // NLIST x 128
using RankedLists = std::array<std::vector<DistanceIndexData>, NQUERY>;                      // per query, ascending by distance
using PreciseScores = std::array<std::array<float, COARSE_PROBE>, NQUERY>;
using PreciseRanking = std::array<std::array<DistanceIndexData, COARSE_PROBE>, NQUERY>;
using ResultIds = std::array<std::array<faiss_idx_t, K>, NQUERY>;
using ResultVectors = std::array<std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, K>, NQUERY>;
using ListSizes = std::array<size_t, NQUERY>;

// The connection the functions below use; not owned.  Without one every request throws std::runtime_error.
void set_transport(wire::Transport *transport);
// Device the centroid shortlist runs on (default 0).
void set_client_device(int device);

// GET /query; throws when the server does not answer
void ping_server();
// Reads ../sift/siftsmall/siftsmall_query.fvecs (aborts when the file is missing, like the reference).
void get_query(QueryBatch &query);

// round 1: centroids from the server, shortlist of inverted lists on the client (reference :22-26)
void get_centroids(CentroidTable &centroids);
void sort_nearest_centroids(const QueryBatch &precise_query, const CentroidTable &centroids, RankedLists &nearest_centroids);

// round 2: PQ scores of every vector of the NPROBE nearest lists, ranked on the client (reference :32-48)
void get_coarse_scores(const RankedLists &sorted_centroids, const QueryBatch &precise_query, std::vector<float> &coarse_scores,
                       std::vector<faiss_idx_t> &coarse_vectors_idx, ListSizes &list_sizes_per_query_coarse);
void compute_nearest_coarse_vectors(const std::vector<float> &coarse_distance_scores, const std::vector<faiss_idx_t> &coarse_vector_indexes,
                                    const ListSizes &list_sizes_per_query_coarse, RankedLists &nearest_coarse_vectors_idx);

// round 3: exact distances of the COARSE_PROBE best candidates, ranked on the client (reference :50-62)
void get_precise_scores(const RankedLists &sorted_coarse_vectors, const QueryBatch &precise_query, PreciseScores &precise_scores);
// Round 3 with the query ENCRYPTED -- the step the reference marks as TODO (reference :14, 28-30): the queries travel as
// BFV ciphertexts of sum_i q_i X^i (route "precisesearch-encrypted", include/server/wire.h), the server multiplies them
// with the packed candidate rows, the client decrypts the inner products and finishes ||q||^2 - 2 <q, x> + ||x||^2 with
// the row norms the server returns.  Vector entries are rounded to integers by the encoding: on integer-valued data
// (SIFT) the scores equal get_precise_scores' bit for bit.  `ctx` must use ring degree 8192 and SEAL's BFVDefault(8192)
// data primes (bfv::Params::seal_default(8192, t)) with t above twice the largest inner product.
void get_precise_scores_encrypted(const RankedLists &sorted_coarse_vectors, const QueryBatch &precise_query, const bfv::Context &ctx,
                                  bfv::Encryptor &encryptor, bfv::Decryptor &decryptor, PreciseScores &precise_scores);
void compute_nearest_precise_vectors(const PreciseScores &precise_scores, const RankedLists &sorted_coarse_vectors,
                                     PreciseRanking &nearest_precise_vectors);

// round 4: the K best vectors themselves (reference :64-69)
void get_precise_vectors_pir(const PreciseRanking &nearest_precise_vectors, ResultVectors &query_results, ResultIds &query_results_idx);

// round 4 with the ids kept private (include/client/pir.h; routes "pir-layout" and "precise-vector-pir-private"): one BFV
// ciphertext per wanted row goes up, one comes back; the rows equal get_precise_vectors_pir's bit for bit.  `ctx` =
// bfv::Params::seal_default(8192, 65537); the Galois keys of the expansion are generated here and sent with the first request.
// Only the first `results_per_query` (<= K) rows of every query are fetched (a retrieval costs the server an expansion of
// 2^levels key switches); the others are left untouched.
void get_precise_vectors_pir_private(const PreciseRanking &nearest_precise_vectors, const bfv::Context &ctx, bfv::KeyGenerator &keygen,
                                     bfv::Encryptor &encryptor, bfv::Decryptor &decryptor, ResultVectors &query_results,
                                     ResultIds &query_results_idx, size_t results_per_query = (size_t)K);

// Recall@{1,10,100} and MRR@{1,10,100} exactly as the reference counts them (client_lib.cpp:243-337): ground-truth
// neighbour j < K of query i is a hit at the position k < K where it appears in the observed results; recall@R counts
// hits with k < R over R * NQUERY; MRR@R adds 1/(k+1) for the FIRST ground-truth neighbour only.
struct RecallStats {
    float recall_1, recall_10, recall_100;
    float mrr_1, mrr_10, mrr_100;
};
RecallStats compute_recall_stats(const ResultIds &observed_query_results_idx, const std::vector<int> &ground_truth, size_t gt_nn_per_query);
// Reads ../sift/siftsmall/siftsmall_groundtruth.ivecs and prints the reference's report.
void benchmark_results(const ResultIds &query_results_idx);
