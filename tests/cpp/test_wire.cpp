// test_wire.cpp -- CPU-only checks of the JSON wire format (include/server/wire.h) and of the recall / MRR
// bookkeeping of the client library (include/client/client_lib.h).  No device is touched.
//   test_wire selftest            JSON reader / writer unit checks; exit code 0 when all pass
//   test_wire recall <file>       file: int64 observed[NQUERY*K], int32 gt_nn, int32 gt[NQUERY*gt_nn]; prints the six figures
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <string>

#include "client_lib.h"
#include "http.h"
#include "pir.h"
#include "wire.h"

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

template <class E, class F>
static bool throws(F &&f) {
    try { f(); } catch (const E &) { return true; } catch (...) { return false; }
    return false;
}

static int selftest() {
    using namespace wire;
    // documents of the protocol's shape
    const Json j = parse(" {\"preciseQuery\": [[1, 2.5, -3e2], [0.1, 1E-3, 7]], \"ids\": [[9007199254740993, -1]], \"s\": \"a\\n\\u0041\", \"t\": true, \"n\": null} ");
    EXPECT(j.kind == Json::Object && j.obj.size() == 5);
    EXPECT(j.at("preciseQuery").at(0).at(1).as_float() == 2.5f);
    EXPECT(j.at("preciseQuery").at(0).at(2).as_float() == -300.0f);
    EXPECT(j.at("preciseQuery").at(1).at(0).as_float() == 0.1f);
    EXPECT(j.at("ids").at(0).at(0).as_int() == 9007199254740993ll);          // beyond 2^53: integers stay exact
    EXPECT(j.at("ids").at(0).at(1).as_int() == -1);
    EXPECT(j.at("s").s == "a\nA" && j.at("t").b && j.at("n").kind == Json::Null);
    EXPECT(j.at("preciseQuery").at(1).at(2).as_int() == 7);
    // error behaviour the handlers rely on
    EXPECT(throws<std::out_of_range>([&] { j.at("missing"); }));
    EXPECT(throws<std::out_of_range>([&] { j.at("ids").at(1); }));
    EXPECT(throws<TypeError>([&] { j.at("n").as_float(); }));
    EXPECT(throws<TypeError>([&] { j.at("s").as_int(); }));
    EXPECT(throws<TypeError>([&] { j.at("t").at(0); }));
    EXPECT(throws<TypeError>([&] { parse("[1.5]").at(0).as_int(); }));
    for (const char *bad : {"", "{", "[1,]", "{\"a\" 1}", "[1 2]", "nul", "[01]", "[1.]", "[1e]", "\"abc", "[1] x", "{\"a\":1,}", "[\"\\q\"]", "-"})
        EXPECT(throws<ParseError>([&] { parse(bad); }));
    EXPECT(parse("[]").arr.empty() && parse("{}").obj.empty() && parse("  3 ").as_int() == 3);
    // writer: every float survives a round trip; integral floats stay float tokens; non-finite -> null
    const float samples[] = {0.0f, -0.0f, 1.0f, 255.0f, 0.1f, 1e-30f, 3.4028235e38f, 1.17549435e-38f, 1e-45f, 16777217.0f, 123456.789f, -2.5e-7f};
    for (float v : samples) {
        std::string s;
        append_float(s, v);
        const Json r = parse(s);
        EXPECT(r.kind == Json::Float);
        EXPECT(r.as_float() == v);
    }
    unsigned x = 12345u;                                                       // a few thousand bit patterns
    for (int i = 0; i < 20000; ++i) {
        x = x * 1664525u + 1013904223u;
        float v;
        std::memcpy(&v, &x, 4);
        if (!std::isfinite(v)) continue;
        std::string s;
        append_float(s, v);
        const float back = parse(s).as_float();
        EXPECT(std::memcmp(&back, &v, 4) == 0 || (back == 0.0f && v == 0.0f));
    }
    std::string s;
    append_float(s, std::numeric_limits<float>::infinity());
    s += ",";
    append_float(s, std::nanf(""));
    EXPECT(s == "null,null");
    s.clear();
    append_int(s, std::numeric_limits<int64_t>::min());
    EXPECT(parse(s).as_int() == std::numeric_limits<int64_t>::min());
    // base64 (ciphertext payloads of the encrypted route): RFC 4648 vectors, every tail length, rejection of bad input
    {
        const char *plain[] = {"", "f", "fo", "foo", "foob", "fooba", "foobar"};
        const char *enc[] = {"", "Zg==", "Zm8=", "Zm9v", "Zm9vYg==", "Zm9vYmE=", "Zm9vYmFy"};
        for (int i = 0; i < 7; ++i) {
            EXPECT(base64_encode(plain[i], std::strlen(plain[i])) == enc[i]);
            const std::vector<uint8_t> back = base64_decode(enc[i]);
            EXPECT(back.size() == std::strlen(plain[i]) && std::memcmp(back.data(), plain[i], back.size()) == 0);
        }
        std::vector<uint8_t> bytes(1000);
        unsigned y = 99u;
        for (auto &b : bytes) { y = y * 1664525u + 1013904223u; b = (uint8_t)(y >> 24); }
        EXPECT(base64_decode(base64_encode(bytes.data(), bytes.size())) == bytes);
        for (const char *bad : {"Zg=", "Z===", "Zm9v!A==", "=AAA", "Zg==Zg=="})
            EXPECT(throws<ParseError>([&] { base64_decode(bad); }));
    }
    // routes: an unknown route is refused before the server is touched
    { Server idle; EXPECT(throws<std::out_of_range>([&] { handle(idle, "nope", ""); })); }
    // client without a transport
    set_transport(nullptr);
    EXPECT(throws<std::runtime_error>([&] { ping_server(); }));
    // client-side bookkeeping that needs no server
    {
        std::vector<float> scores;
        std::vector<faiss_idx_t> ids;
        std::array<size_t, NQUERY> sizes;
        for (size_t q = 0; q < (size_t)NQUERY; ++q) {
            sizes[q] = (size_t)COARSE_PROBE + q;
            for (size_t i = 0; i < sizes[q]; ++i) { scores.push_back((float)((i * 7919 + q) % 251)); ids.push_back((faiss_idx_t)(1000 * q + i)); }
        }
        std::array<std::vector<DistanceIndexData>, NQUERY> nearest;
        compute_nearest_coarse_vectors(scores, ids, sizes, nearest);
        for (size_t q = 0; q < (size_t)NQUERY; ++q) {
            EXPECT(nearest[q].size() == sizes[q]);
            for (size_t i = 1; i < nearest[q].size(); ++i)                     // ascending, ties in arrival order
                EXPECT(nearest[q][i - 1].distance < nearest[q][i].distance ||
                       (nearest[q][i - 1].distance == nearest[q][i].distance && nearest[q][i - 1].idx < nearest[q][i].idx));
        }
        sizes[2] = (size_t)COARSE_PROBE - 1;
        EXPECT(throws<std::runtime_error>([&] { compute_nearest_coarse_vectors(scores, ids, sizes, nearest); }));
        std::array<std::vector<DistanceIndexData>, NQUERY> few;
        std::vector<float> cs; std::vector<faiss_idx_t> ci; std::array<size_t, NQUERY> ls;
        std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> query{};
        EXPECT(throws<std::runtime_error>([&] { get_coarse_scores(few, query, cs, ci, ls); }));   // fewer than NPROBE centroids
    }
    // private retrieval (include/client/pir.h): the host-side arithmetic -- layout, query encoding, row decoding -- needs no device
    {
        const pir::Layout lay = pir::Layout::make(8192, 128, 10000);
        EXPECT(lay.rows_per_poly == 32 && lay.n_polys == 313 && lay.levels == 9 && lay.poly_of(9999) == 312 && lay.slot_of(9999) == 15);
        EXPECT(pir::Layout::make(8192, 128, 1).levels == 0 && pir::Layout::make(8192, 128, 33).levels == 1 && pir::Layout::make(8192, 64, 8192u * 64).levels == 13);
        {                                                                                              // more than N polynomials: columns
            const pir::Layout big = pir::Layout::make(8192, 128, 8192u * 32 + 1);
            EXPECT(big.n_polys == 8193 && big.n_sel == 8192 && big.n_cols == 2 && big.levels == 13);
            EXPECT(big.sel_of(8192u * 32) == 0 && big.col_of(8192u * 32) == 1 && big.col_of(8192u * 32 - 1) == 0 && big.sel_of(8192u * 32 - 1) == 8191);
            const pir::Layout sift = pir::Layout::make(8192, 128, 1000000);
            EXPECT(sift.n_polys == 31250 && sift.n_cols == 4 && sift.levels == 13);
            const pir::Layout forced = pir::Layout::make(8192, 128, 1000, 8);
            EXPECT(forced.n_polys == 32 && forced.n_sel == 8 && forced.n_cols == 4 && forced.levels == 3);
        }
        EXPECT(throws<std::invalid_argument>([] { pir::Layout::make(1024, 1024, 4); }));              // a row does not fit a polynomial
        const std::vector<uint32_t> elts = pir::galois_elements(8192, 9);
        EXPECT(elts.size() == 9 && elts[0] == 8193 && elts[1] == 4097 && elts[8] == 33);
        std::vector<uint64_t> plain(8192);
        pir::encode_query(lay, 65537, 9999, plain.data());
        uint64_t nonzero = 0, at = 0;
        for (size_t i = 0; i < plain.size(); ++i) if (plain[i]) { ++nonzero; at = i; }
        EXPECT(nonzero == 1 && at == 312 && plain[at] * 512 % 65537 == 1);                              // 2^-9 mod t at the row's polynomial
        EXPECT(throws<std::out_of_range>([&] { pir::encode_query(lay, 65537, 10000, plain.data()); }));
        EXPECT(throws<std::invalid_argument>([&] { pir::encode_query(lay, 65536, 0, plain.data()); }));   // even modulus: 2 is not invertible
        // a row as the server packs it (two 16-bit halves per float) comes back bit for bit, -0.0f and a NaN pattern included
        float row[128], back[128];
        for (int i = 0; i < 128; ++i) row[i] = (float)(i * 37 % 101) / 7.0f - 5.0f;
        row[3] = -0.0f;
        const uint32_t nan_bits = 0x7FC12345u;
        std::memcpy(&row[4], &nan_bits, 4);
        std::fill(plain.begin(), plain.end(), 0);
        const size_t c0 = (size_t)lay.slot_of(9999) * 256;
        for (int i = 0; i < 128; ++i) { uint32_t b; std::memcpy(&b, &row[i], 4); plain[c0 + 2 * i] = b & 0xFFFF; plain[c0 + 2 * i + 1] = b >> 16; }
        pir::decode_row(lay, plain.data(), 9999, back);
        EXPECT(std::memcmp(row, back, sizeof row) == 0);
    }
    std::printf(failures ? "selftest: %d failure(s)\n" : "selftest: ok\n", failures);
    return failures ? 1 : 0;
}

static int recall(const char *path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) { std::printf("cannot open %s\n", path); return 2; }
    std::array<std::array<faiss_idx_t, K>, NQUERY> observed;
    in.read(reinterpret_cast<char *>(observed.data()), sizeof observed);
    int32_t gt_nn = 0;
    in.read(reinterpret_cast<char *>(&gt_nn), 4);
    std::vector<int> gt((size_t)NQUERY * (size_t)(gt_nn > 0 ? gt_nn : 0));
    in.read(reinterpret_cast<char *>(gt.data()), (std::streamsize)(gt.size() * 4));
    if (!in) { std::printf("short file\n"); return 2; }
    try {
        const RecallStats s = compute_recall_stats(observed, gt, (size_t)gt_nn);
        std::printf("%.9g %.9g %.9g %.9g %.9g %.9g\n", s.recall_1, s.recall_10, s.recall_100, s.mrr_1, s.mrr_10, s.mrr_100);
    } catch (const std::exception &e) {
        std::printf("error: %s\n", e.what());
        return 3;
    }
    return 0;
}

// HTTP listener without a Server behind it: prints "PORT <n>", answers `n_requests` requests, exits.  Routes: GET /query
// -> a tiny centroid array; POST /echo -> the request body; POST /boom -> the handler throws (500); POST /badbody -> a
// std::out_of_range that is not a routing failure (500); anything else -> "no such route" (404).
static int http(size_t n_requests, int request_timeout_ms = 0) {
    wire::HttpListener listener([](const std::string &method, const std::string &route, const std::string &body) -> std::string {
        if (route == "query" && method == "GET") return "[[1.5,2.0]]";
        if (route == "echo") return body;
        if (route == "boom") throw std::runtime_error("boom");
        if (route == "badbody") throw std::out_of_range("key 'preciseQuery' not found");
        throw std::out_of_range("no such route: " + route);
    }, "127.0.0.1", 0, 8u << 20);
    if (request_timeout_ms > 0) listener.set_request_timeout_ms(request_timeout_ms);
    std::printf("PORT %u\n", (unsigned)listener.port());
    std::fflush(stdout);
    const size_t served = listener.serve(n_requests);
    std::printf("served %zu\n", served);
    return 0;
}

// the C++ client against the same listener (in-process thread): keep-alive reuse, statuses, a 3 MB body
#include <thread>
static int http_loopback() {
    wire::HttpListener listener([](const std::string &, const std::string &route, const std::string &body) -> std::string {
        if (route == "echo") return body;
        if (route == "query") return "[]";
        throw std::out_of_range("no such route: " + route);
    }, "127.0.0.1", 0);
    std::thread th([&] { listener.serve(); });
    {
        wire::HttpTransport t("127.0.0.1", listener.port());
        EXPECT(t.get("query") == "[]" && t.last_status == 200);
        std::string big(3u << 20, 'x');
        for (size_t i = 0; i < big.size(); i += 4097) big[i] = (char)('a' + i % 26);
        EXPECT(t.post("echo", big) == big);
        EXPECT(t.post("echo", "") == "");
        EXPECT(throws<std::runtime_error>([&] { t.post("nowhere", "{}"); }) && t.last_status == 404);
        EXPECT(t.post("echo", "{\"a\":1}") == "{\"a\":1}");          // the connection survived the 404
        EXPECT(t.bytes_sent == big.size() + 9 && t.bytes_received >= big.size());
    }
    listener.stop();
    th.join();
    std::printf(failures ? "http loopback: %d FAILURES\n" : "http loopback: ok\n", failures);
    return failures ? 1 : 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && std::strcmp(argv[1], "selftest") == 0) return selftest();
    if (argc >= 3 && std::strcmp(argv[1], "http") == 0) return http((size_t)std::atoll(argv[2]), argc >= 4 ? std::atoi(argv[3]) : 0);
    if (argc >= 2 && std::strcmp(argv[1], "http-loopback") == 0) return http_loopback();
    if (argc >= 3 && std::strcmp(argv[1], "recall") == 0) return recall(argv[2]);
    std::printf("usage: test_wire selftest | recall <file> | http <n requests> [request timeout ms] | http-loopback\n");
    return 2;
}
