// wire.cpp -- JSON bodies and route handlers of the PreFHEtch server (include/server/wire.h), restating
// /root/reference/src/server/controllers/Query.cc:9-127 without Drogon or nlohmann.
#include "../../include/server/wire.h"

#include <algorithm>
#include <array>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <memory>

namespace wire {

// ---- reader ------------------------------------------------------------------------------------------
namespace {

struct Reader {
    const char *p, *end, *begin;
    [[noreturn]] void fail(const char *what) const {
        throw ParseError(std::string("JSON parse error at byte ") + std::to_string(p - begin) + ": " + what);
    }
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool lit(const char *word) {
        const size_t n = std::strlen(word);
        if ((size_t)(end - p) < n || std::memcmp(p, word, n) != 0) return false;
        p += n;
        return true;
    }
    std::string string() {
        std::string out;
        ++p;                                              // opening quote
        for (;;) {
            if (p >= end) fail("unterminated string");
            const char c = *p++;
            if (c == '"') return out;
            if ((unsigned char)c < 0x20) fail("control character in string");
            if (c != '\\') { out.push_back(c); continue; }
            if (p >= end) fail("unterminated escape");
            const char e = *p++;
            switch (e) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    if (end - p < 4) fail("short \\u escape");
                    unsigned cp = 0;
                    for (int k = 0; k < 4; ++k) {
                        const char h = *p++;
                        cp <<= 4;
                        if (h >= '0' && h <= '9') cp |= (unsigned)(h - '0');
                        else if (h >= 'a' && h <= 'f') cp |= (unsigned)(h - 'a' + 10);
                        else if (h >= 'A' && h <= 'F') cp |= (unsigned)(h - 'A' + 10);
                        else fail("bad \\u escape");
                    }
                    // keys of this protocol are ASCII; other code points are kept as UTF-8 (BMP only, no surrogate pairing)
                    if (cp < 0x80) out.push_back((char)cp);
                    else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                    else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                    break;
                }
                default: fail("unknown escape");
            }
        }
    }
    Json number() {
        const char *s = p;
        bool integral = true;
        if (p < end && *p == '-') ++p;
        if (p >= end || *p < '0' || *p > '9') fail("digit expected");
        if (*p == '0') ++p; else while (p < end && *p >= '0' && *p <= '9') ++p;
        if (p < end && *p == '.') {
            integral = false;
            ++p;
            if (p >= end || *p < '0' || *p > '9') fail("digit expected after the decimal point");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            integral = false;
            ++p;
            if (p < end && (*p == '+' || *p == '-')) ++p;
            if (p >= end || *p < '0' || *p > '9') fail("digit expected in the exponent");
            while (p < end && *p >= '0' && *p <= '9') ++p;
        }
        const std::string tok(s, p);
        Json j;
        if (integral) {
            errno = 0;
            char *e = nullptr;
            const long long v = std::strtoll(tok.c_str(), &e, 10);
            if (errno == 0 && e && *e == 0) { j.kind = Json::Int; j.i = v; j.f = (double)v; return j; }
        }
        j.kind = Json::Float;
        j.f = std::strtod(tok.c_str(), nullptr);
        return j;
    }
    Json value(int depth) {
        if (depth > 64) fail("nesting too deep");
        ws();
        if (p >= end) fail("value expected");
        Json j;
        switch (*p) {
            case '{': {
                ++p;
                j.kind = Json::Object;
                ws();
                if (p < end && *p == '}') { ++p; return j; }
                for (;;) {
                    ws();
                    if (p >= end || *p != '"') fail("object key expected");
                    std::string key = string();
                    ws();
                    if (p >= end || *p != ':') fail("':' expected");
                    ++p;
                    j.obj.emplace_back(std::move(key), value(depth + 1));
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == '}') { ++p; return j; }
                    fail("',' or '}' expected");
                }
            }
            case '[': {
                ++p;
                j.kind = Json::Array;
                ws();
                if (p < end && *p == ']') { ++p; return j; }
                for (;;) {
                    j.arr.push_back(value(depth + 1));
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == ']') { ++p; return j; }
                    fail("',' or ']' expected");
                }
            }
            case '"': j.kind = Json::String; j.s = string(); return j;
            case 't': if (lit("true")) { j.kind = Json::Bool; j.b = true; return j; } fail("bad literal");
            case 'f': if (lit("false")) { j.kind = Json::Bool; j.b = false; return j; } fail("bad literal");
            case 'n': if (lit("null")) return j; fail("bad literal");
            default: return number();
        }
    }
};

const char *kind_name(Json::Kind k) {
    static const char *const names[] = {"null", "boolean", "integer", "number", "string", "array", "object"};
    return names[k];
}

}  // namespace

Json parse(const std::string &text) {
    Reader r{text.data(), text.data() + text.size(), text.data()};
    Json j = r.value(0);
    r.ws();
    if (r.p != r.end) r.fail("trailing characters");
    return j;
}

const Json &Json::at(const std::string &key) const {
    if (kind != Object) throw TypeError(std::string("cannot use at(key) with ") + kind_name(kind));
    for (const auto &kv : obj)
        if (kv.first == key) return kv.second;
    throw std::out_of_range("key '" + key + "' not found");
}

const Json &Json::at(size_t index) const {
    if (kind != Array) throw TypeError(std::string("cannot use at(index) with ") + kind_name(kind));
    if (index >= arr.size()) throw std::out_of_range("array index " + std::to_string(index) + " is out of range");
    return arr[index];
}

float Json::as_float() const {
    if (kind == Int) return (float)i;
    if (kind == Float) return (float)f;
    throw TypeError(std::string("type must be number, but is ") + kind_name(kind));
}

int64_t Json::as_int() const {
    if (kind == Int) return i;
    if (kind == Float && std::nearbyint(f) == f && std::fabs(f) < 9.2e18) return (int64_t)f;
    throw TypeError(std::string("type must be integer, but is ") + kind_name(kind));
}

// ---- writer ------------------------------------------------------------------------------------------
void append_float(std::string &out, float v) {
    if (!std::isfinite(v)) { out += "null"; return; }
    char buf[32];
    const int n = std::snprintf(buf, sizeof buf, "%.9g", (double)v);
    out.append(buf, (size_t)n);
    // keep it a JSON *float* token, as nlohmann does for floating values ("3.0", not "3")
    if (std::strpbrk(buf, ".eE") == nullptr) out += ".0";
}

void append_int(std::string &out, int64_t v) {
    char buf[24];
    const int n = std::snprintf(buf, sizeof buf, "%lld", (long long)v);
    out.append(buf, (size_t)n);
}

namespace {

template <class Row>
void write_float_row(std::string &out, const Row &row) {
    out.push_back('[');
    bool first = true;
    for (const float v : row) {
        if (!first) out.push_back(',');
        first = false;
        append_float(out, v);
    }
    out.push_back(']');
}

template <class It>
void write_int_list(std::string &out, It b, It e) {
    out.push_back('[');
    for (It it = b; it != e; ++it) {
        if (it != b) out.push_back(',');
        append_int(out, (int64_t)*it);
    }
    out.push_back(']');
}

// std::array<std::array<T, COLS>, ROWS> from a JSON array of arrays: shorter input throws (at()), longer input is
// read up to the array's extent -- nlohmann's from_json for std::array behaves the same way
template <size_t ROWS, size_t COLS>
void read_floats(const Json &j, std::array<std::array<float, COLS>, ROWS> &out) {
    for (size_t r = 0; r < ROWS; ++r)
        for (size_t c = 0; c < COLS; ++c) out[r][c] = j.at(r).at(c).as_float();
}

template <size_t ROWS, size_t COLS>
void read_ids(const Json &j, std::array<std::array<faiss_idx_t, COLS>, ROWS> &out) {
    for (size_t r = 0; r < ROWS; ++r)
        for (size_t c = 0; c < COLS; ++c) out[r][c] = j.at(r).at(c).as_int();
}

}  // namespace

// ---- handlers ----------------------------------------------------------------------------------------
std::string handle_query(const Server &server) {
    std::vector<std::array<float, PRECISE_VECTOR_DIMENSIONS>> centroids;
    server.retrieve_centroids(centroids);
    std::string out;
    out.reserve(centroids.size() * PRECISE_VECTOR_DIMENSIONS * 12);
    out.push_back('[');
    for (size_t i = 0; i < centroids.size(); ++i) {
        if (i) out.push_back(',');
        write_float_row(out, centroids[i]);
    }
    out.push_back(']');
    return out;
}

std::string handle_coarse_search(const Server &server, const std::string &body) {
    const Json req = parse(body);
    std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> precise_query;
    std::array<std::array<faiss_idx_t, NPROBE>, NQUERY> nearest_centroids;
    read_floats(req.at("preciseQuery"), precise_query);
    read_ids(req.at("nearestCentroidIndexes"), nearest_centroids);

    std::vector<float> coarse_distance_scores;
    std::vector<faiss::idx_t> coarse_vector_indexes;
    std::array<size_t, NQUERY> list_sizes_per_query;
    server.coarseSearch(precise_query, nearest_centroids, coarse_distance_scores, coarse_vector_indexes, list_sizes_per_query);

    std::string out;
    out.reserve(coarse_distance_scores.size() * 20 + 128);
    out += "{\"coarseDistanceScores\":";
    write_float_row(out, coarse_distance_scores);
    out += ",\"coarseVectorIndexes\":";
    write_int_list(out, coarse_vector_indexes.begin(), coarse_vector_indexes.end());
    out += ",\"listSizesPerQuery\":";
    write_int_list(out, list_sizes_per_query.begin(), list_sizes_per_query.end());
    out.push_back('}');
    return out;
}

std::string handle_precise_search(const Server &server, const std::string &body) {
    const Json req = parse(body);
    std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> precise_query;
    std::array<std::array<faiss_idx_t, COARSE_PROBE>, NQUERY> nearest_coarse_vectors_id;
    read_floats(req.at("preciseQuery"), precise_query);
    read_ids(req.at("nearestCoarseVectorIndexes"), nearest_coarse_vectors_id);

    std::array<std::array<float, COARSE_PROBE>, NQUERY> precise_distance_scores;
    server.preciseSearch(precise_query, nearest_coarse_vectors_id, precise_distance_scores);

    std::string out = "{\"preciseDistanceScores\":[";
    for (size_t q = 0; q < (size_t)NQUERY; ++q) {
        if (q) out.push_back(',');
        write_float_row(out, precise_distance_scores[q]);
    }
    out += "]}";
    return out;
}

std::string handle_precise_vector_pir(Server &server, const std::string &body) {
    const Json req = parse(body);
    std::array<std::array<faiss_idx_t, K>, NQUERY> ids;
    read_ids(req.at("nearestPreciseVectorIndexes"), ids);

    // NQUERY * K * 128 floats: on the heap (the reference keeps it on the handler's stack)
    auto results = std::make_unique<std::array<std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, K>, NQUERY>>();
    server.preciseVectorPIR(ids, *results);

    std::string out;
    out.reserve((size_t)NQUERY * K * PRECISE_VECTOR_DIMENSIONS * 6);
    out += "{\"queryResults\":[";
    for (size_t q = 0; q < (size_t)NQUERY; ++q) {
        if (q) out.push_back(',');
        out.push_back('[');
        for (size_t r = 0; r < (size_t)K; ++r) {
            if (r) out.push_back(',');
            write_float_row(out, (*results)[q][r]);
        }
        out.push_back(']');
    }
    out += "]}";
    return out;
}

// ---- base64 ------------------------------------------------------------------------------------------
std::string base64_encode(const void *data, size_t bytes) {
    static const char tab[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    const uint8_t *p = static_cast<const uint8_t *>(data);
    std::string out;
    out.reserve((bytes + 2) / 3 * 4);
    size_t i = 0;
    for (; i + 3 <= bytes; i += 3) {
        const uint32_t v = (uint32_t)p[i] << 16 | (uint32_t)p[i + 1] << 8 | p[i + 2];
        out.push_back(tab[v >> 18]); out.push_back(tab[(v >> 12) & 63]); out.push_back(tab[(v >> 6) & 63]); out.push_back(tab[v & 63]);
    }
    if (i + 1 == bytes) {
        const uint32_t v = (uint32_t)p[i] << 16;
        out.push_back(tab[v >> 18]); out.push_back(tab[(v >> 12) & 63]); out += "==";
    } else if (i + 2 == bytes) {
        const uint32_t v = (uint32_t)p[i] << 16 | (uint32_t)p[i + 1] << 8;
        out.push_back(tab[v >> 18]); out.push_back(tab[(v >> 12) & 63]); out.push_back(tab[(v >> 6) & 63]); out.push_back('=');
    }
    return out;
}

std::vector<uint8_t> base64_decode(const std::string &text) {
    static int8_t rev[256];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < 256; ++i) rev[i] = -1;
        const char *tab = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) rev[(unsigned char)tab[i]] = (int8_t)i;
        init = true;
    }
    if (text.size() % 4) throw ParseError("base64: length is not a multiple of 4");
    std::vector<uint8_t> out;
    out.reserve(text.size() / 4 * 3);
    for (size_t i = 0; i < text.size(); i += 4) {
        int v[4];
        int pad = 0;
        for (int k = 0; k < 4; ++k) {
            const unsigned char c = (unsigned char)text[i + k];
            if (c == '=' && i + 4 == text.size() && k >= 2) { v[k] = 0; ++pad; continue; }
            if (pad || rev[c] < 0) throw ParseError("base64: bad character at byte " + std::to_string(i + k));
            v[k] = rev[c];
        }
        const uint32_t w = (uint32_t)v[0] << 18 | (uint32_t)v[1] << 12 | (uint32_t)v[2] << 6 | (uint32_t)v[3];
        out.push_back((uint8_t)(w >> 16));
        if (pad < 2) out.push_back((uint8_t)(w >> 8));
        if (pad < 1) out.push_back((uint8_t)w);
    }
    return out;
}

std::string handle_precise_search_encrypted(const Server &server, const std::string &body) {
    const Json req = parse(body);
    std::array<std::array<faiss_idx_t, COARSE_PROBE>, NQUERY> ids;
    read_ids(req.at("nearestCoarseVectorIndexes"), ids);
    const Json &blob = req.at("queryCiphertexts");
    if (blob.kind != Json::String) throw TypeError("queryCiphertexts must be a base64 string");
    const std::vector<uint8_t> raw = base64_decode(blob.s);
    constexpr size_t ct_words = static_cast<size_t>(NQUERY) * 2 * Server::ENC_LIMBS * Server::ENC_RING_DEGREE;
    if (raw.size() != ct_words * 8) throw std::out_of_range("queryCiphertexts: expected " + std::to_string(ct_words * 8) + " bytes, got " + std::to_string(raw.size()));
    std::vector<uint64_t> in(ct_words), out(ct_words * Server::ENC_POLYS_PER_QUERY);
    std::memcpy(in.data(), raw.data(), raw.size());
    std::array<std::array<float, COARSE_PROBE>, NQUERY> norms;
    server.preciseSearchEncryptedHost(in.data(), ids, out.data(), norms);
    std::string resp = "{\"resultCiphertexts\":\"";
    resp += base64_encode(out.data(), out.size() * 8);
    resp += "\",\"rowNorms\":[";
    for (size_t q = 0; q < (size_t)NQUERY; ++q) {
        if (q) resp.push_back(',');
        write_float_row(resp, norms[q]);
    }
    resp += "]}";
    return resp;
}

// ---- private row retrieval (include/client/pir.h) -------------------------------------------------------------
std::string handle_pir_layout(const Server &server) {
    return "{\"rows\":" + std::to_string(server.pirRows()) + ",\"levels\":" + std::to_string(server.pirLevels()) + ",\"cols\":" + std::to_string(server.pirCols()) +
           ",\"ringDegree\":" + std::to_string(Server::ENC_RING_DEGREE) + ",\"plainModulus\":" + std::to_string(Server::PIR_PLAIN_MODULUS) + "}";
}

// residues of every limb-polynomial must be canonical (< q_limb): what the device kernels assume of their inputs
static void check_residues(const uint64_t *w, size_t n_limb_polys, uint32_t limbs_cycle, const char *what) {
    constexpr size_t N = Server::ENC_RING_DEGREE;
    // key layout [..][K = L + 1][N] cycles over L data primes + the special prime; ciphertext layout cycles over the L data primes
    for (size_t p = 0; p < n_limb_polys; ++p) {
        const uint32_t l = (uint32_t)(p % limbs_cycle);
        const uint64_t q = l < Server::ENC_LIMBS ? Server::ENC_MODULI[l] : Server::ENC_SPECIAL_PRIME;
        for (size_t i = 0; i < N; ++i)
            if (w[p * N + i] >= q) throw std::out_of_range(std::string(what) + ": residue out of range (limb " + std::to_string(l) + ")");
    }
}

std::string handle_precise_vector_pir_private(const Server &server, const std::string &body) {
    // The Galois keys of the expansion (tens of megabytes) are sent once per SESSION and kept for the requests that follow without
    // "galoisKeys".  A client names its session with the optional string "session" (the reference's single-client demo sends
    // none: session ""), so one client's keys never replace another's; at most MAX_SESSIONS key sets are kept, the least recently USED one
    // dropped.  Session ids are not authenticated: whoever names a session may replace its keys (as anyone may call any route of the reference).
    static std::mutex key_lock;
    static std::vector<std::pair<std::string, std::vector<uint64_t>>> sessions;
    constexpr size_t MAX_SESSIONS = 8;
    const Json req = parse(body);
    const int64_t count_in = req.at("count").as_int();
    if (count_in <= 0 || (size_t)count_in > (size_t)NQUERY * K) throw std::out_of_range("count must be in [1, NQUERY * K]");
    const size_t count = (size_t)count_in;
    constexpr size_t per = 2 * (size_t)Server::ENC_LIMBS * Server::ENC_RING_DEGREE;
    const size_t key_words = (size_t)server.pirLevels() * Server::ENC_LIMBS * 2 * (Server::ENC_LIMBS + 1) * Server::ENC_RING_DEGREE;
    std::vector<uint64_t> keys;
    {
        const Json *k = nullptr, *sid = nullptr;
        if (req.kind == Json::Object)
            for (const auto &kv : req.obj) {
                if (kv.first == "galoisKeys") k = &kv.second;
                if (kv.first == "session") sid = &kv.second;
            }
        if (sid && sid->kind != Json::String) throw TypeError("session must be a string");
        const std::string session = sid ? sid->s : std::string();
        std::vector<uint64_t> fresh;
        if (k) {
            if (k->kind != Json::String) throw TypeError("galoisKeys must be a base64 string");
            const std::vector<uint8_t> raw = base64_decode(k->s);
            if (raw.size() != key_words * 8) throw std::out_of_range("galoisKeys: expected " + std::to_string(key_words * 8) + " bytes, got " + std::to_string(raw.size()));
            fresh.resize(key_words);
            std::memcpy(fresh.data(), raw.data(), raw.size());
            check_residues(fresh.data(), key_words / Server::ENC_RING_DEGREE, Server::ENC_LIMBS + 1, "galoisKeys");
        }
        std::lock_guard<std::mutex> g(key_lock);
        auto it = std::find_if(sessions.begin(), sessions.end(), [&](const auto &e) { return e.first == session; });
        if (k) {
            if (it != sessions.end()) sessions.erase(it);
            if (sessions.size() >= MAX_SESSIONS) sessions.erase(sessions.begin());
            sessions.emplace_back(session, std::move(fresh));
            it = sessions.end() - 1;
        }
        if (it == sessions.end() || it->second.size() != key_words)
            throw std::out_of_range("precise-vector-pir-private: no Galois keys in this request and none kept for this session");
        if (it + 1 != sessions.end()) { std::rotate(it, it + 1, sessions.end()); it = sessions.end() - 1; }      // most recently used last
        keys = it->second;
    }
    const Json &blob = req.at("queryCiphertexts");
    if (blob.kind != Json::String) throw TypeError("queryCiphertexts must be a base64 string");
    const std::vector<uint8_t> raw = base64_decode(blob.s);
    if (raw.size() != count * per * 8) throw std::out_of_range("queryCiphertexts: expected " + std::to_string(count * per * 8) + " bytes, got " + std::to_string(raw.size()));
    std::vector<uint64_t> in(count * per), out(count * server.pirCols() * per);
    std::memcpy(in.data(), raw.data(), raw.size());
    check_residues(in.data(), count * 2 * Server::ENC_LIMBS, Server::ENC_LIMBS, "queryCiphertexts");
    server.preciseVectorPIRPrivateHost(in.data(), count, keys.data(), out.data());
    return "{\"replyCiphertexts\":\"" + base64_encode(out.data(), out.size() * 8) + "\"}";
}

std::string handle(Server &server, const std::string &route, const std::string &body) {
    if (route == "query") return handle_query(server);
    if (route == "coarsesearch") return handle_coarse_search(server, body);
    if (route == "precisesearch") return handle_precise_search(server, body);
    if (route == "precise-vector-pir") return handle_precise_vector_pir(server, body);
    if (route == "precisesearch-encrypted") return handle_precise_search_encrypted(server, body);
    if (route == "pir-layout") return handle_pir_layout(server);
    if (route == "precise-vector-pir-private") return handle_precise_vector_pir_private(server, body);
    throw std::out_of_range("no such route: " + route);
}

}  // namespace wire
