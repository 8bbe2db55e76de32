// wire.h -- the JSON bodies of the four PreFHEtch routes and the handlers that serve them, without an HTTP stack.
//
// The reference's controllers (/root/reference/src/server/controllers/Query.cc:9-127) parse the request body with
// nlohmann::json, call one Server method and dump the result.  This header restates that layer over `class Server`
// with a small JSON reader/writer of its own (nlohmann is not a dependency of this build):
//
//   GET  /query               -> [[c_0 .. c_127] x NLIST]                                   (Query.cc:15-23)
//   POST /coarsesearch        {"preciseQuery": [NQUERY][128], "nearestCentroidIndexes": [NQUERY][NPROBE]}
//                             -> {"coarseDistanceScores": [...], "coarseVectorIndexes": [...],
//                                 "listSizesPerQuery": [NQUERY]}                            (Query.cc:31-63)
//   POST /precisesearch       {"preciseQuery": [NQUERY][128], "nearestCoarseVectorIndexes": [NQUERY][COARSE_PROBE]}
//                             -> {"preciseDistanceScores": [NQUERY][COARSE_PROBE]}          (Query.cc:66-99)
//   POST /precise-vector-pir  {"nearestPreciseVectorIndexes": [NQUERY][K]}
//                             -> {"queryResults": [NQUERY][K][128]}                         (Query.cc:102-127)
//
// Added route (the encrypted form of /precisesearch, SURVEY.md 8(f-2): "then replace float arrays by serialized
// ciphertexts"; the framing is this build's own, SEAL's save/load being unavailable):
//   POST /precisesearch-encrypted  {"nearestCoarseVectorIndexes": [NQUERY][COARSE_PROBE],
//                                   "queryCiphertexts": base64 of NQUERY x [2][4][8192] little-endian uint64}
//                                  -> {"resultCiphertexts": base64 of NQUERY x ENC_POLYS_PER_QUERY x [2][4][8192] uint64,
//                                      "rowNorms": [NQUERY][COARSE_PROBE]}
//
// A transport (Drogon in the reference, anything that moves a body) calls handle(); the client side
// (include/client/client_lib.h) talks to a `Transport`, of which InProcessTransport is the one shipped here.
// Error behaviour follows nlohmann's as the reference relies on it: malformed JSON -> wire::ParseError, a missing
// key or a too-short array -> std::out_of_range, a value of the wrong type -> wire::TypeError; nothing is caught
// inside the handlers (Drogon turns an escaping exception into a 500).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "server_lib.h"

namespace wire {

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };
struct TypeError : std::runtime_error { using std::runtime_error::runtime_error; };

// Minimal JSON document: null, booleans, numbers (integers kept exact in 64 bits), strings, arrays, objects.
struct Json {
    enum Kind { Null, Bool, Int, Float, String, Array, Object };
    Kind kind = Null;
    bool b = false;
    int64_t i = 0;
    double f = 0.0;
    std::string s;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    const Json &at(const std::string &key) const;     // object member; std::out_of_range when absent
    const Json &at(size_t index) const;               // array element; std::out_of_range past the end
    float as_float() const;                           // Int or Float; TypeError otherwise (null included, as in nlohmann)
    int64_t as_int() const;                           // Int, or a Float with an integral value
};

Json parse(const std::string &text);

// Number formatting of the writer: fp32 values with 9 significant digits (round-trips every float), non-finite
// values as null (what nlohmann::json::dump emits for them).
void append_float(std::string &out, float v);
void append_int(std::string &out, int64_t v);

// The four routes.  `route` is the path without the leading slash: "query", "coarsesearch", "precisesearch",
// "precise-vector-pir", "precisesearch-encrypted".  handle() dispatches; an unknown route throws std::out_of_range.
std::string handle_query(const Server &server);
std::string handle_coarse_search(const Server &server, const std::string &body);
std::string handle_precise_search(const Server &server, const std::string &body);
std::string handle_precise_vector_pir(Server &server, const std::string &body);
std::string handle_precise_search_encrypted(const Server &server, const std::string &body);
// private row retrieval (include/client/pir.h): {"rows", "levels", "ringDegree", "plainModulus"}; and
// {"count": n, "queryCiphertexts": base64 [n][2][4][8192], "galoisKeys": base64 [levels][4][2][5][8192] (first request; kept for
// the following ones)} -> {"replyCiphertexts": base64 [n][2][4][8192]}
std::string handle_pir_layout(const Server &server);
std::string handle_precise_vector_pir_private(const Server &server, const std::string &body);
// RFC 4648 base64 of raw bytes (ciphertext payloads)
std::string base64_encode(const void *data, size_t bytes);
std::vector<uint8_t> base64_decode(const std::string &text);      // ParseError on malformed input
std::string handle(Server &server, const std::string &route, const std::string &body);

// What the client needs from a connection: the body of GET <route> / POST <route>.
struct Transport {
    virtual ~Transport() = default;
    virtual std::string get(const std::string &route) = 0;
    virtual std::string post(const std::string &route, const std::string &body) = 0;
};

// Client and server in one process: every request is serialised, handled and parsed exactly as over HTTP.
class InProcessTransport : public Transport {
  public:
    explicit InProcessTransport(Server &server) : m_Server(server) {}
    std::string get(const std::string &route) override { return count(handle(m_Server, route, std::string())); }
    std::string post(const std::string &route, const std::string &body) override {
        bytes_sent += body.size();
        return count(handle(m_Server, route, body));
    }
    size_t bytes_sent = 0, bytes_received = 0;        // request / response body bytes so far

  private:
    std::string count(std::string r) { bytes_received += r.size(); return r; }
    Server &m_Server;
};

}  // namespace wire
