// test_server.cpp -- exercises class Server (the reference's own interface, include/server/server_lib.h) on a
// real GPU with the reference's shapes (NBASE=10000, NQUERY=5, COARSE_PROBE=200, K=100, NLIST=256) and checks
// every output against literal restatements of the reference loops (src/server/server_lib.cpp:140-196,
// src/client/client_lib.cpp:50-81).  Also pins the public signatures to the reference's at compile time.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <type_traits>

#include "../../include/server/server_lib.h"
#include "../../include/server/http.h"
#include "../../include/server/wire.h"
#include <thread>
#include "../../include/client/client_lib.h"

// ---- signature pins (reference include/server/server_lib.h:19-49) ----
using Q = std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY>;
static_assert(std::is_same_v<decltype(&Server::getInstance), std::shared_ptr<Server> &(*)()>);
static_assert(std::is_same_v<decltype(&Server::init_index), void (Server::*)()>);
static_assert(std::is_same_v<decltype(&Server::run_webserver), void (Server::*)()>);
static_assert(std::is_same_v<decltype(&Server::retrieve_centroids), void (Server::*)(std::vector<std::array<float, 128>> &) const>);
static_assert(std::is_same_v<decltype(&Server::coarseSearch),
                             void (Server::*)(const Q &, const std::array<std::array<faiss::idx_t, NPROBE>, NQUERY> &, std::vector<float> &,
                                              std::vector<faiss::idx_t> &, std::array<size_t, NQUERY> &) const>);
static_assert(std::is_same_v<decltype(&Server::preciseSearch),
                             void (Server::*)(const Q &, const std::array<std::array<faiss::idx_t, COARSE_PROBE>, NQUERY> &,
                                              std::array<std::array<float, COARSE_PROBE>, NQUERY> &) const>);
static_assert(std::is_same_v<decltype(&Server::preciseVectorPIR),
                             void (Server::*)(const std::array<std::array<faiss_idx_t, K>, NQUERY> &,
                                              std::array<std::array<std::array<float, 128>, K>, NQUERY> &)>);
static_assert(PRECISE_VECTOR_DIMENSIONS == 128 && NPROBE == 20 && COARSE_PROBE == 200 && K == 100 && NBASE == 10000 &&
              NQUERY == 5 && NLIST == 256 && SUB_QUANTIZERS == 32 && SUB_QUANTIZER_SIZE == 8);

static int fails = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++fails; } } while (0)

int main(int argc, char **argv) {
    const bool gaussian = argc > 1 && std::string(argv[1]) == "gaussian";
    std::mt19937 rng(20250801);
    std::uniform_int_distribution<int> pix(0, 255);
    std::normal_distribution<float> gauss(0.f, 1.f);
    auto draw = [&] { return gaussian ? gauss(rng) : static_cast<float>(pix(rng)); };
    const size_t NT = 6000;                                    // training vectors (the reference trains on siftsmall_learn)
    std::vector<float> base(static_cast<size_t>(NBASE) * 128), train(NT * 128);
    for (float &v : base) v = draw();
    for (float &v : train) v = draw();
    auto &srv = Server::getInstance();
    EXPECT(srv.get() == Server::getInstance().get());
    bool threw = false;
    try { std::vector<std::array<float, 128>> c; srv->retrieve_centroids(c); } catch (const std::runtime_error &) { threw = true; }
    EXPECT(threw);                                   // not initialised yet -> runtime_error, like an unusable index in the reference
    srv->init_from_memory(base.data(), NBASE, train.data(), NT);
    std::vector<float> cent, books; std::vector<uint8_t> codes; std::vector<faiss::idx_t> stored_ids; std::vector<uint64_t> off;
    srv->export_index(cent, books, codes, stored_ids, off);
    EXPECT(cent.size() == static_cast<size_t>(NLIST) * 128 && books.size() == 32u * 256 * 4 && stored_ids.size() == static_cast<size_t>(NBASE));
    {   // every base row is stored exactly once, in the list of its nearest centroid (ties -> smaller id), fp64 check with slack
        std::vector<int> seen(NBASE, 0);
        for (size_t l = 0; l < static_cast<size_t>(NLIST); ++l)
            for (uint64_t v = off[l]; v < off[l + 1]; ++v) {
                const int64_t id = stored_ids[v];
                ++seen[id];
                double dl = 0, best = 1e300;
                for (size_t c = 0; c < static_cast<size_t>(NLIST); ++c) {
                    double dd = 0;
                    for (int t = 0; t < 128; ++t) { const double df = double(base[id * 128 + t]) - double(cent[c * 128 + t]); dd += df * df; }
                    if (c == l) dl = dd;
                    best = std::min(best, dd);
                }
                EXPECT(dl <= best * (1 + 1e-5));
            }
        EXPECT(std::all_of(seen.begin(), seen.end(), [](int c) { return c == 1; }));
    }

    Q query;
    for (auto &q : query) for (float &v : q) v = draw();

    // retrieve_centroids: server_lib.cpp:101-109
    std::vector<std::array<float, 128>> got_c;
    srv->retrieve_centroids(got_c);
    EXPECT(got_c.size() == static_cast<size_t>(NLIST));
    EXPECT(std::memcmp(got_c.data(), cent.data(), cent.size() * 4) == 0);

    // preciseSearch: literal restatement of server_lib.cpp:151-164
    std::uniform_int_distribution<int64_t> pick(0, NBASE - 1);
    std::array<std::array<faiss::idx_t, COARSE_PROBE>, NQUERY> ids;
    for (auto &row : ids) for (auto &v : row) v = pick(rng);
    std::array<std::array<float, COARSE_PROBE>, NQUERY> got_d, ref_d;
    srv->preciseSearch(query, ids, got_d);
    for (int i = 0; i < NQUERY; i++)
        for (int j = 0; j < COARSE_PROBE; j++) {
            float dist = 0.0;
            const float *row = base.data() + ids[i][j] * PRECISE_VECTOR_DIMENSIONS;
            for (int k = 0; k < PRECISE_VECTOR_DIMENSIONS; k++) dist += std::pow((row[k] - query[i][k]), 2);
            ref_d[i][j] = dist;
        }
    EXPECT(std::memcmp(got_d.data(), ref_d.data(), sizeof got_d) == 0);      // bit-exact, also on gaussian data

    // preciseVectorPIR: server_lib.cpp:169-196
    std::array<std::array<faiss_idx_t, K>, NQUERY> kid;
    for (auto &row : kid) for (auto &v : row) v = pick(rng);
    auto res = std::make_unique<std::array<std::array<std::array<float, 128>, K>, NQUERY>>();
    srv->preciseVectorPIR(kid, *res);
    for (int i = 0; i < NQUERY; i++)
        for (int j = 0; j < K; j++) EXPECT(std::memcmp((*res)[i][j].data(), base.data() + kid[i][j] * 128, 512) == 0);

    // nearestCentroids == client sort_nearest_centroids (client_lib.cpp:50-81) truncated to NPROBE
    std::array<std::array<faiss::idx_t, NPROBE>, NQUERY> cid;
    std::array<std::array<float, NPROBE>, NQUERY> cdist;
    srv->nearestCentroids(query, cid, cdist);
    for (int i = 0; i < NQUERY; i++) {
        std::vector<std::pair<float, int64_t>> all;
        for (int j = 0; j < NLIST; j++) {
            float distance = 0.0;
            for (int k = 0; k < 128; k++) distance += std::pow(query[i][k] - cent[j * 128 + k], 2);
            all.push_back({distance, j});
        }
        std::sort(all.begin(), all.end());
        for (int j = 0; j < NPROBE; j++) {
            // trained centroids are real-valued, so fp32 distances are held to the north-star tolerance (1e-5 relative);
            // an id may only differ from the reference order inside that band
            EXPECT(std::fabs(cdist[i][j] - all[j].first) <= 1e-5f * all[j].first);
            if (cid[i][j] != all[j].second) {
                float dref = -1.f;
                for (auto &pr : all) if (pr.second == cid[i][j]) dref = pr.first;
                EXPECT(std::fabs(dref - all[j].first) <= 2e-5f * all[j].first);
            }
        }
    }

    // coarseSearch: literal restatement of the ADC scan over the client-chosen lists (server_lib.cpp:111-138; arithmetic
    // contract of pf_ivfpq.hip) on the exported index content, bit-exact
    {
        std::array<std::array<faiss::idx_t, NPROBE>, NQUERY> probe = cid;       // the lists a client would pick
        std::vector<float> s; std::vector<faiss::idx_t> l; std::array<size_t, NQUERY> sz{};
        srv->coarseSearch(query, probe, s, l, sz);
        std::vector<float> rs; std::vector<faiss::idx_t> rl; std::array<size_t, NQUERY> rsz{};
        std::vector<float> lut(32 * 256);
        for (int i = 0; i < NQUERY; i++) {
            for (int j = 0; j < NPROBE; j++) {
                const int64_t li = probe[i][j];
                for (int m = 0; m < 32; m++)
                    for (int c = 0; c < 256; c++) {
                        float acc = 0.f;
                        for (int t = 0; t < 4; t++) {
                            volatile float r = query[i][m * 4 + t] - cent[li * 128 + m * 4 + t];
                            volatile float diff = r - books[(m * 256 + c) * 4 + t];
                            volatile float sq = diff * diff;
                            acc = acc + sq;
                        }
                        lut[m * 256 + c] = acc;
                    }
                for (uint64_t v = off[li]; v < off[li + 1]; ++v) {
                    float dis = 0.f;
                    for (int m = 0; m < 32; m++) dis = dis + lut[m * 256 + codes[v * 32 + m]];
                    rs.push_back(dis); rl.push_back(stored_ids[v]); ++rsz[i];
                }
            }
        }
        EXPECT(sz == rsz);
        EXPECT(l == rl);
        EXPECT(s.size() == rs.size() && std::memcmp(s.data(), rs.data(), rs.size() * 4) == 0);
        size_t total = 0;
        for (size_t v : sz) total += v;
        EXPECT(total == s.size() && total > 0);
    }

    // ---- the whole 4-round protocol of the reference client (src/client/client.cpp:7-80) against this Server, on
    // clustered synthetic data, scored like benchmark_results (src/client/client_lib.cpp:243-337): Recall@k and MRR@k
    // against exact ground truth.  SIFT is not available offline; a Gaussian mixture stands in for it.
    if (gaussian) {
        std::mt19937 r2(7);
        std::normal_distribution<float> unit(0.f, 1.f);
        const int n_clusters = 64;
        std::vector<float> centers(n_clusters * 128);
        for (float &v : centers) v = 40.f * unit(r2);
        auto sample = [&](float *dst) {
            const int c = static_cast<int>(r2() % n_clusters);
            for (int t = 0; t < 128; t++) dst[t] = centers[c * 128 + t] + 4.f * unit(r2);
        };
        std::vector<float> b2(static_cast<size_t>(NBASE) * 128), t2v(NT * 128);
        for (size_t i = 0; i < static_cast<size_t>(NBASE); i++) sample(b2.data() + i * 128);
        for (size_t i = 0; i < NT; i++) sample(t2v.data() + i * 128);
        Q q2;
        for (auto &q : q2) sample(q.data());
        srv->init_from_memory(b2.data(), NBASE, t2v.data(), NT);
        // round 1: centroids -> client-side shortlist of NPROBE lists (client_lib.cpp:50-81,100-102)
        std::vector<std::array<float, 128>> cents;
        srv->retrieve_centroids(cents);
        std::array<std::array<faiss::idx_t, NPROBE>, NQUERY> probe;
        for (int i = 0; i < NQUERY; i++) {
            std::vector<std::pair<float, int64_t>> all;
            for (int j = 0; j < NLIST; j++) {
                float distance = 0.0;
                for (int k = 0; k < 128; k++) distance += std::pow(q2[i][k] - cents[j][k], 2);
                all.push_back({distance, j});
            }
            std::sort(all.begin(), all.end());
            for (int j = 0; j < NPROBE; j++) probe[i][j] = all[j].second;
        }
        // round 2: coarse scores -> top COARSE_PROBE per query (client_lib.cpp:122-156)
        std::vector<float> cs; std::vector<faiss::idx_t> ci; std::array<size_t, NQUERY> sz{};
        srv->coarseSearch(q2, probe, cs, ci, sz);
        std::array<std::array<faiss::idx_t, COARSE_PROBE>, NQUERY> coarse_ids;
        size_t at = 0;
        bool enough = true;
        for (int i = 0; i < NQUERY; i++) {
            std::vector<std::pair<float, int64_t>> v;
            for (size_t j = 0; j < sz[i]; j++) v.push_back({cs[at + j], ci[at + j]});
            at += sz[i];
            std::sort(v.begin(), v.end());
            if (v.size() < static_cast<size_t>(COARSE_PROBE)) { enough = false; break; }
            for (int j = 0; j < COARSE_PROBE; j++) coarse_ids[i][j] = v[j].second;
        }
        EXPECT(enough);
        if (enough) {
            // round 3: exact distances of the shortlist -> top K (client_lib.cpp:158-207)
            std::array<std::array<float, COARSE_PROBE>, NQUERY> pd;
            srv->preciseSearch(q2, coarse_ids, pd);
            double recall_at[3] = {0, 0, 0}, mrr10 = 0;
            const int ks[3] = {1, 10, 100};
            for (int i = 0; i < NQUERY; i++) {
                std::vector<std::pair<float, int64_t>> v;
                for (int j = 0; j < COARSE_PROBE; j++) v.push_back({pd[i][j], coarse_ids[i][j]});
                std::sort(v.begin(), v.end());
                std::vector<std::pair<double, int64_t>> gt;                      // exact ground truth
                for (int64_t j = 0; j < NBASE; j++) {
                    double dd = 0;
                    for (int t = 0; t < 128; t++) { const double df = double(b2[j * 128 + t]) - double(q2[i][t]); dd += df * df; }
                    gt.push_back({dd, j});
                }
                std::sort(gt.begin(), gt.end());
                for (int a = 0; a < 3; a++) {
                    int hit = 0;
                    for (int x = 0; x < ks[a]; x++)
                        for (int y = 0; y < ks[a]; y++) if (v[x].second == gt[y].second) { ++hit; break; }
                    recall_at[a] += double(hit) / ks[a] / NQUERY;
                }
                for (int x = 0; x < 10; x++) if (v[x].second == gt[0].second) { mrr10 += 1.0 / (x + 1) / NQUERY; break; }
            }
            std::printf("protocol on a 64-cluster Gaussian mixture: Recall@1 %.3f Recall@10 %.3f Recall@100 %.3f MRR@10 %.3f\n",
                        recall_at[0], recall_at[1], recall_at[2], mrr10);
            EXPECT(recall_at[1] >= 0.9 && recall_at[2] >= 0.8 && mrr10 >= 0.9);

            // ---- the same run through the client library and the JSON wire format (include/client/client_lib.h,
            // include/server/wire.h): main() of the reference client, src/client/client.cpp:7-80, with an in-process
            // transport.  Every intermediate must equal what the direct Server calls above produced.
            wire::InProcessTransport link(*srv);
            set_transport(&link);
            ping_server();
            std::vector<std::array<float, PRECISE_VECTOR_DIMENSIONS>> c_cents;
            get_centroids(c_cents);
            EXPECT(c_cents.size() == cents.size() && std::memcmp(c_cents.data(), cents.data(), cents.size() * sizeof cents[0]) == 0);
            std::array<std::vector<DistanceIndexData>, NQUERY> c_near;
            sort_nearest_centroids(q2, c_cents, c_near);
            for (int i = 0; i < NQUERY; i++) {
                EXPECT(c_near[i].size() == static_cast<size_t>(NLIST));
                for (int j = 0; j < NPROBE; j++) EXPECT(c_near[i][j].idx == probe[i][j]);
                float distance = 0.0;                                            // the reference's host loop, bit for bit
                for (int k = 0; k < 128; k++) distance += std::pow(q2[i][k] - cents[c_near[i][0].idx][k], 2);
                EXPECT(std::memcmp(&distance, &c_near[i][0].distance, 4) == 0);
            }
            std::vector<float> c_cs; std::vector<faiss_idx_t> c_ci; std::array<size_t, NQUERY> c_sz{};
            get_coarse_scores(c_near, q2, c_cs, c_ci, c_sz);
            EXPECT(c_sz == sz && c_ci == ci && c_cs.size() == cs.size() && std::memcmp(c_cs.data(), cs.data(), cs.size() * 4) == 0);
            std::array<std::vector<DistanceIndexData>, NQUERY> c_coarse;
            compute_nearest_coarse_vectors(c_cs, c_ci, c_sz, c_coarse);
            std::array<std::array<float, COARSE_PROBE>, NQUERY> c_pd;
            get_precise_scores(c_coarse, q2, c_pd);
            bool same_coarse = true;
            for (int i = 0; i < NQUERY; i++)
                for (int j = 0; j < COARSE_PROBE; j++) same_coarse = same_coarse && c_coarse[i][j].idx == coarse_ids[i][j];
            EXPECT(same_coarse);
            if (same_coarse) EXPECT(std::memcmp(c_pd.data(), pd.data(), sizeof pd) == 0);
            auto c_best = std::make_unique<std::array<std::array<DistanceIndexData, COARSE_PROBE>, NQUERY>>();
            compute_nearest_precise_vectors(c_pd, c_coarse, *c_best);
            auto c_rows = std::make_unique<std::array<std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, K>, NQUERY>>();
            std::array<std::array<faiss_idx_t, K>, NQUERY> c_ids;
            get_precise_vectors_pir(*c_best, *c_rows, c_ids);
            std::vector<int> gt_ids(static_cast<size_t>(NQUERY) * K);             // exact ground truth, the .ivecs layout
            for (int i = 0; i < NQUERY; i++) {
                std::vector<std::pair<double, int64_t>> gt;
                for (int64_t j = 0; j < NBASE; j++) {
                    double dd = 0;
                    for (int t = 0; t < 128; t++) { const double df = double(b2[j * 128 + t]) - double(q2[i][t]); dd += df * df; }
                    gt.push_back({dd, j});
                }
                std::sort(gt.begin(), gt.end());
                for (int j = 0; j < K; j++) gt_ids[i * K + j] = static_cast<int>(gt[j].second);
                for (int j = 0; j < K; j++)                                        // the returned vectors are the base rows
                    EXPECT(std::memcmp((*c_rows)[i][j].data(), b2.data() + c_ids[i][j] * 128, 512) == 0);
                for (int j = 1; j < K; j++) EXPECT((*c_best)[i][j - 1].distance <= (*c_best)[i][j].distance);
            }
            const RecallStats st = compute_recall_stats(c_ids, gt_ids, K);
            std::printf("client library over the wire format: Recall@1 %.3f Recall@10 %.3f Recall@100 %.3f MRR@1 %.3f MRR@10 %.3f MRR@100 %.3f; "
                        "%zu request bytes, %zu response bytes\n", st.recall_1, st.recall_10, st.recall_100, st.mrr_1, st.mrr_10, st.mrr_100,
                        link.bytes_sent, link.bytes_received);
            EXPECT(st.recall_10 >= 0.9 && st.recall_100 >= 0.8 && st.mrr_10 >= 0.9 && link.bytes_sent > 0 && link.bytes_received > 0);
            // a malformed body surfaces as an exception, as it does under Drogon
            bool threw = false;
            try { link.post("coarsesearch", "{\"preciseQuery\": [[1,2]]}"); } catch (const std::out_of_range &) { threw = true; }
            EXPECT(threw);
            set_transport(nullptr);

            // ---- and over HTTP: the POSIX-socket listener (include/server/http.h) on an ephemeral port in a thread, the
            // client library through HttpTransport -- what an unmodified reference client does against Drogon.  Results
            // must equal the in-process run bit for bit.
            {
                wire::HttpListener listener(*srv, "127.0.0.1", 0);
                std::thread th([&] { listener.serve(); });
                {
                    wire::HttpTransport web("127.0.0.1", listener.port());
                    set_transport(&web);
                    ping_server();
                    std::vector<std::array<float, PRECISE_VECTOR_DIMENSIONS>> h_cents;
                    get_centroids(h_cents);
                    EXPECT(h_cents.size() == cents.size() && std::memcmp(h_cents.data(), cents.data(), cents.size() * sizeof cents[0]) == 0);
                    std::vector<float> h_cs; std::vector<faiss_idx_t> h_ci; std::array<size_t, NQUERY> h_sz{};
                    get_coarse_scores(c_near, q2, h_cs, h_ci, h_sz);
                    EXPECT(h_sz == sz && h_ci == ci && h_cs.size() == cs.size() && std::memcmp(h_cs.data(), cs.data(), cs.size() * 4) == 0);
                    std::array<std::array<float, COARSE_PROBE>, NQUERY> h_pd;
                    get_precise_scores(c_coarse, q2, h_pd);
                    EXPECT(std::memcmp(h_pd.data(), c_pd.data(), sizeof c_pd) == 0);
                    auto h_rows = std::make_unique<std::array<std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, K>, NQUERY>>();
                    std::array<std::array<faiss_idx_t, K>, NQUERY> h_ids;
                    get_precise_vectors_pir(*c_best, *h_rows, h_ids);
                    EXPECT(h_ids == c_ids && std::memcmp(h_rows->data(), c_rows->data(), sizeof *c_rows) == 0);
                    bool refused = false;                                       // a malformed body: 500, as under Drogon
                    try { web.post("coarsesearch", "{\"preciseQuery\": [[1,2]]}"); } catch (const std::runtime_error &) { refused = web.last_status == 500; }
                    EXPECT(refused);
                    std::printf("client library over HTTP/1.1 (127.0.0.1:%u): identical results; %zu request bytes, %zu response bytes\n",
                                (unsigned)listener.port(), web.bytes_sent, web.bytes_received);
                    set_transport(nullptr);
                }
                listener.stop();
                th.join();
            }
        }
    }

    Timer t; long long us = -1, ms = -1;
    t.StartTimer(); t.StopTimer(); t.getDuration(us, ms);
    EXPECT(us >= 0 && ms >= 0);
    if (fails) std::printf("test_server: %d FAILURES\n", fails);
    else std::printf("test_server: OK (%s data)\n", gaussian ? "gaussian" : "integer");
    return fails ? 1 : 0;
}
