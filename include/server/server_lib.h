// server_lib.h -- class Server, the drop-in boundary of the PreFHEtch server
// (/root/reference/include/server/server_lib.h:12-50).  Public names, signatures and constness are the
// reference's, so src/server/controllers/Query.cc:15-18,48-51,86-88,116-117 and src/server/server.cpp:9-11
// compile against this header unchanged.  What differs is behind the pimpl: the faiss objects are replaced
// by handles of libprefhetch_hip.so (include/prefhetch_hip.h) -- the base matrix and the centroids live in
// MI355X HBM and every distance stage runs as a HIP kernel.
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "client_server_utils.h"

// The reference spells some index types faiss::idx_t (server_lib.h:33,36,41) and others faiss_idx_t;
// both are int64_t.  faiss itself is not a dependency of this build.
namespace faiss {
using idx_t = int64_t;
}

class Server {
  private:
    struct Impl;                       // device handles, stream, staging buffers (src: prefhetch_amd/csrc/server_lib.cpp)
    std::unique_ptr<Impl> m_Impl;

  public:
    Server();
    ~Server();
    // Singleton for static access across all controllers (reference :20-23)
    static std::shared_ptr<Server> &getInstance() {
        static std::shared_ptr<Server> server = std::make_shared<Server>();
        return server;
    }

    // Reference src/server/server_lib.cpp:55-99: when no cached index file exists, trains the IVF-PQ index
    // (NLIST=256 coarse centroids, 32 x 8-bit sub-quantizers) on ../sift/siftsmall/siftsmall_learn.fvecs, adds
    // ../sift/siftsmall/siftsmall_base.fvecs and writes the cache; otherwise loads base + cache.  The cache is this
    // build's own format (the reference's is faiss::write_index).  k-means assignments run on the GPU flat index.
    void init_index();
    // Serves the routes over HTTP/1.1 on 0.0.0.0:8080 (reference :48-53) with the POSIX-socket listener of http.h; blocks.
    void run_webserver();

    void retrieve_centroids(std::vector<std::array<float, PRECISE_VECTOR_DIMENSIONS>> &centroids) const;
    void coarseSearch(const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &precise_query,
                      const std::array<std::array<faiss::idx_t, NPROBE>, NQUERY> &nearest_centroid_idx,
                      std::vector<float> &coarse_distance_scores, std::vector<faiss::idx_t> &coarse_distance_indexes,
                      std::array<size_t, NQUERY> &list_sizes_per_query) const;
    void preciseSearch(const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &precise_query,
                       const std::array<std::array<faiss::idx_t, COARSE_PROBE>, NQUERY> &nearest_coarse_vector_idx,
                       std::array<std::array<float, COARSE_PROBE>, NQUERY> &precise_distance_scores) const;
    void preciseVectorPIR(const std::array<std::array<faiss_idx_t, K>, NQUERY> &k_nearest_precise_vectors_idx,
                          std::array<std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, K>, NQUERY> &query_results);

    // ---- additions of this build (not in the reference) ---------------------------------------------------
    // Dataset-free init: trains on `train` [nt][128] (on the base vectors when train == nullptr) and adds base [nb][128].
    void init_from_memory(const float *base, size_t nb, const float *train = nullptr, size_t nt = 0, int device = 0);
    // Trained index content in flat form (test / serialisation hook): centroids [NLIST][128], codebooks [32][256][4],
    // list-contiguous codes [ntotal][32] and ids [ntotal], list offsets [NLIST+1].
    void export_index(std::vector<float> &centroids, std::vector<float> &codebooks, std::vector<uint8_t> &codes,
                      std::vector<faiss::idx_t> &ids, std::vector<uint64_t> &list_offsets) const;
    // The encrypted form of preciseSearch -- the step the reference's TODOs leave in the clear
    // (include/client/client_lib.h:14,28-30): the query arrives as BFV ciphertexts of the polynomial sum_i q_i X^i
    // (include/client/bfv.h: encode_query), ring degree 8192 over SEAL's BFVDefault(8192) data primes, SEAL's
    // Ciphertext::data() layout [2][L][N], already in HBM.  result block b of query i is a ciphertext whose plaintext
    // carries <q_i, x_{ids[i][64 b + j]}> at coefficient 128 j (decode_inner_products); the client adds its own
    // ||q||^2 and the row norms to obtain the squared distances preciseSearch returns.
    static constexpr uint32_t ENC_RING_DEGREE = 8192, ENC_LIMBS = 4;
    static constexpr uint64_t ENC_MODULI[ENC_LIMBS] = {0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001};
    static constexpr uint64_t ENC_SPECIAL_PRIME = 0xFFFFFEBC001;     // the key modulus behind them (SEAL BFVDefault(8192)'s fifth prime): key-switching keys carry a limb of it
    static constexpr uint32_t ENC_ROWS_PER_POLY = ENC_RING_DEGREE / PRECISE_VECTOR_DIMENSIONS;
    static constexpr uint32_t ENC_POLYS_PER_QUERY = (COARSE_PROBE + ENC_ROWS_PER_POLY - 1) / ENC_ROWS_PER_POLY;
    void preciseSearchEncrypted(const uint64_t *query_ct_device /* [NQUERY][2][4][8192] */,
                                const std::array<std::array<faiss::idx_t, COARSE_PROBE>, NQUERY> &nearest_coarse_vector_idx,
                                uint64_t *result_ct_device /* [NQUERY][ENC_POLYS_PER_QUERY][2][4][8192] */) const;
    // The same with host buffers (what a transport has in hand): uploads the query ciphertexts, runs the device form,
    // downloads the results, and returns ||x||^2 of every candidate row -- the client needs them to finish the distances
    // ||q||^2 - 2 <q, x> + ||x||^2, and they do not depend on the query.
    void preciseSearchEncryptedHost(const uint64_t *query_ct_host, const std::array<std::array<faiss::idx_t, COARSE_PROBE>, NQUERY> &nearest_coarse_vector_idx,
                                    uint64_t *result_ct_host, std::array<std::array<float, COARSE_PROBE>, NQUERY> &row_norms) const;
    // The private form of preciseVectorPIR (include/client/pir.h: SealPIR-style retrieval built from pf_ct_pt_mul and
    // pf_key_switch): the ids stay with the client.  Ring degree 8192, BFVDefault(8192) data primes + its special prime,
    // plaintext modulus PIR_PLAIN_MODULUS.  `count` query ciphertexts [count][2][4][8192] in, `count` x pirCols() reply ciphertexts out (a retrieval's columns together); the client's
    // Galois keys for the pirLevels() expansion rounds (pir::galois_elements), [levels][4][2][5][8192] words in
    // pf_key_switch's layout.  The packed database is built from the base rows on first use.
    static constexpr uint64_t PIR_PLAIN_MODULUS = 65537;
    size_t pirRows() const;
    uint32_t pirLevels() const;
    size_t pirCols() const;           // ciphertexts per retrieval in the reply (pir::Layout::n_cols: 4 at 1M rows)
    void preciseVectorPIRPrivateHost(const uint64_t *query_ct_host, size_t count, const uint64_t *galois_keys_host, uint64_t *reply_ct_host) const;
    // The executed flat-L2 shortlist of the protocol (client sort_nearest_centroids, src/client/client_lib.cpp:50-81)
    // on the server's IndexFlatL2 over the centroids: top-NPROBE centroid ids (and squared distances) per query.
    void nearestCentroids(const std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> &precise_query,
                          std::array<std::array<faiss::idx_t, NPROBE>, NQUERY> &nearest_centroid_idx,
                          std::array<std::array<float, NPROBE>, NQUERY> &nearest_centroid_dist) const;
};
