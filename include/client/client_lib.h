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

const std::string server_addr = "http://localhost:8080/";     // reference :7 (used by HTTP transports)

struct DistanceIndexData {
    float distance;
    faiss_idx_t idx;
};

// The connection the functions below use; not owned.  Without one every request throws std::runtime_error.
void set_transport(wire::Transport *transport);
// Device the centroid shortlist runs on (default 0).
void set_client_device(int device);

void ping_server();                                            // GET /query; throws when the server does not answer

// Reads ../sift/siftsmall/siftsmall_query.fvecs (aborts when the file is missing, like the reference).
void get_query(std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &query);

void get_centroids(std::vector<std::array<float, PRECISE_VECTOR_DIMENSIONS>> &centroids);
void sort_nearest_centroids(const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &precise_query,
                            const std::vector<std::array<float, PRECISE_VECTOR_DIMENSIONS>> &centroids,
                            std::array<std::vector<DistanceIndexData>, NQUERY> &nearest_centroids);

void get_coarse_scores(const std::array<std::vector<DistanceIndexData>, NQUERY> &sorted_centroids,
                       const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &precise_query,
                       std::vector<float> &coarse_scores, std::vector<faiss_idx_t> &coarse_vectors_idx,
                       std::array<size_t, NQUERY> &list_sizes_per_query_coarse);

void compute_nearest_coarse_vectors(const std::vector<float> &coarse_distance_scores,
                                    const std::vector<faiss_idx_t> &coarse_vector_indexes,
                                    const std::array<size_t, NQUERY> &list_sizes_per_query_coarse,
                                    std::array<std::vector<DistanceIndexData>, NQUERY> &nearest_coarse_vectors_idx);

void get_precise_scores(const std::array<std::vector<DistanceIndexData>, NQUERY> &sorted_coarse_vectors,
                        const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &precise_query,
                        std::array<std::array<float, COARSE_PROBE>, NQUERY> &precise_scores);

void compute_nearest_precise_vectors(const std::array<std::array<float, COARSE_PROBE>, NQUERY> &precise_scores,
                                     const std::array<std::vector<DistanceIndexData>, NQUERY> &sorted_coarse_vectors,
                                     std::array<std::array<DistanceIndexData, COARSE_PROBE>, NQUERY> &nearest_precise_vectors);

void get_precise_vectors_pir(const std::array<std::array<DistanceIndexData, COARSE_PROBE>, NQUERY> &nearest_precise_vectors,
                             std::array<std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, K>, NQUERY> &query_results,
                             std::array<std::array<faiss_idx_t, K>, NQUERY> &query_results_idx);

// Recall@{1,10,100} and MRR@{1,10,100} exactly as the reference counts them (client_lib.cpp:243-337): ground-truth
// neighbour j < K of query i is a hit at the position k < K where it appears in the observed results; recall@R counts
// hits with k < R over R * NQUERY; MRR@R adds 1/(k+1) for the FIRST ground-truth neighbour only.
struct RecallStats {
    float recall_1, recall_10, recall_100;
    float mrr_1, mrr_10, mrr_100;
};
RecallStats compute_recall_stats(const std::array<std::array<faiss_idx_t, K>, NQUERY> &observed_query_results_idx,
                                 const std::vector<int> &ground_truth, size_t gt_nn_per_query);
// Reads ../sift/siftsmall/siftsmall_groundtruth.ivecs and prints the reference's report.
void benchmark_results(const std::array<std::array<faiss_idx_t, K>, NQUERY> &query_results_idx);
