// http.cpp -- POSIX-socket HTTP/1.1 listener and client for the PreFHEtch routes (include/server/http.h).
#include "http.h"

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <stdexcept>

namespace wire {

namespace {

bool send_all(int fd, const char *p, size_t n) {
    while (n) {
        const ssize_t w = ::send(fd, p, n, MSG_NOSIGNAL);
        if (w < 0) { if (errno == EINTR) continue; return false; }        // EAGAIN here = SO_SNDTIMEO expired: the peer is not reading
        p += w; n -= (size_t)w;
    }
    return true;
}

std::string lower(std::string s) { std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)std::tolower(c); }); return s; }

// One HTTP message off a socket: start line, headers (lower-cased names), body by Content-Length.  `buf` carries bytes
// read past the previous message (keep-alive).  Returns false on EOF / error before a complete head.
struct Message { std::string start; std::vector<std::pair<std::string, std::string>> headers; std::string body; };
enum class Recv { Ok, Closed, Bad, TooLarge };

const std::string *header(const Message &m, const char *name) {
    for (const auto &h : m.headers) if (h.first == name) return &h.second;
    return nullptr;
}

bool fill(int fd, std::string &buf) {
    char tmp[65536];
    for (;;) {
        const ssize_t r = ::recv(fd, tmp, sizeof tmp, 0);
        if (r > 0) { buf.append(tmp, (size_t)r); return true; }
        if (r < 0 && errno == EINTR) continue;
        return false;
    }
}

// on_head: called once the head is parsed and before the body is read (the server answers "Expect: 100-continue" there)
Recv read_message(int fd, std::string &buf, Message &m, size_t max_body, const std::function<void(const Message &)> &on_head) {
    size_t end;
    while ((end = buf.find("\r\n\r\n")) == std::string::npos) {
        if (buf.size() > (1u << 20)) return Recv::Bad;
        if (!fill(fd, buf)) return buf.empty() ? Recv::Closed : Recv::Bad;
    }
    const std::string head = buf.substr(0, end);
    buf.erase(0, end + 4);
    size_t pos = head.find("\r\n");
    m.start = head.substr(0, pos);
    m.headers.clear();
    while (pos != std::string::npos) {
        const size_t next = head.find("\r\n", pos + 2);
        const std::string line = head.substr(pos + 2, next == std::string::npos ? std::string::npos : next - pos - 2);
        const size_t colon = line.find(':');
        if (colon == std::string::npos) { if (!line.empty()) return Recv::Bad; }
        else {
            size_t v = colon + 1;
            while (v < line.size() && (line[v] == ' ' || line[v] == '\t')) ++v;
            m.headers.emplace_back(lower(line.substr(0, colon)), line.substr(v));
        }
        pos = next;
    }
    size_t len = 0;
    if (const std::string *cl = header(m, "content-length")) {
        if (cl->empty() || cl->find_first_not_of("0123456789") != std::string::npos || cl->size() > 18) return Recv::Bad;
        len = (size_t)std::stoull(*cl);
    } else if (header(m, "transfer-encoding")) return Recv::Bad;             // chunked bodies: not spoken here
    if (len > max_body) return Recv::TooLarge;
    if (on_head) on_head(m);
    while (buf.size() < len) if (!fill(fd, buf)) return Recv::Bad;
    m.body = buf.substr(0, len);
    buf.erase(0, len);
    return Recv::Ok;
}

void respond(int fd, int status, const char *reason, const std::string &body, bool keep_alive, const char *type = "application/json") {
    std::string head = "HTTP/1.1 " + std::to_string(status) + " " + reason + "\r\nContent-Type: " + type + "\r\nContent-Length: " +
                       std::to_string(body.size()) + "\r\nConnection: " + (keep_alive ? "keep-alive" : "close") + "\r\n\r\n";
    if (send_all(fd, head.data(), head.size())) send_all(fd, body.data(), body.size());
}

}  // namespace

HttpListener::HttpListener(HttpHandler handler, const std::string &address, uint16_t port, size_t max_body)
    : m_Handler(std::move(handler)), m_MaxBody(max_body) {
    m_Fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (m_Fd < 0) throw std::runtime_error(std::string("HttpListener: socket: ") + std::strerror(errno));
    int one = 1;
    ::setsockopt(m_Fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_port = htons(port);
    if (::inet_pton(AF_INET, address.c_str(), &a.sin_addr) != 1) { ::close(m_Fd); throw std::runtime_error("HttpListener: bad IPv4 address " + address); }
    if (::bind(m_Fd, reinterpret_cast<sockaddr *>(&a), sizeof a) < 0 || ::listen(m_Fd, 64) < 0) {
        const std::string why = std::strerror(errno);
        ::close(m_Fd);
        throw std::runtime_error("HttpListener: cannot listen on " + address + ":" + std::to_string(port) + ": " + why);
    }
    socklen_t sl = sizeof a;
    ::getsockname(m_Fd, reinterpret_cast<sockaddr *>(&a), &sl);
    m_Port = ntohs(a.sin_port);
}

HttpListener::HttpListener(Server &server, const std::string &address, uint16_t port)
    : HttpListener([&server](const std::string &, const std::string &route, const std::string &body) { return handle(server, route, body); },
                   address, port) {}

HttpListener::~HttpListener() { if (m_Fd >= 0) ::close(m_Fd); }

void HttpListener::stop() { m_Stop = true; }

namespace {

// Parses ONE complete message out of the front of `buf` without touching the socket.  NeedMore: the head or the body is not all
// there yet (`head_done` then says whether the head is: the caller answers "Expect: 100-continue" at that point).
enum class Parse { Ok, NeedMore, Bad, TooLarge };
Parse try_parse(std::string &buf, Message &m, size_t max_body, bool &head_done) {
    head_done = false;
    const size_t end = buf.find("\r\n\r\n");
    if (end == std::string::npos) return buf.size() > (1u << 20) ? Parse::Bad : Parse::NeedMore;
    const std::string head = buf.substr(0, end);
    size_t pos = head.find("\r\n");
    m.start = head.substr(0, pos);
    m.headers.clear();
    while (pos != std::string::npos) {
        const size_t next = head.find("\r\n", pos + 2);
        const std::string line = head.substr(pos + 2, next == std::string::npos ? std::string::npos : next - pos - 2);
        const size_t colon = line.find(':');
        if (colon == std::string::npos) { if (!line.empty()) return Parse::Bad; }
        else {
            size_t v = colon + 1;
            while (v < line.size() && (line[v] == ' ' || line[v] == '\t')) ++v;
            m.headers.emplace_back(lower(line.substr(0, colon)), line.substr(v));
        }
        pos = next;
    }
    size_t len = 0;
    if (const std::string *cl = header(m, "content-length")) {
        if (cl->empty() || cl->find_first_not_of("0123456789") != std::string::npos || cl->size() > 18) return Parse::Bad;
        len = (size_t)std::stoull(*cl);
    } else if (header(m, "transfer-encoding")) return Parse::Bad;            // chunked bodies: not spoken here
    if (len > max_body) return Parse::TooLarge;
    head_done = true;
    if (buf.size() < end + 4 + len) return Parse::NeedMore;
    m.body = buf.substr(end + 4, len);
    buf.erase(0, end + 4 + len);
    return Parse::Ok;
}

}  // namespace

// One thread, many connections: the listening socket and every open client socket sit in ONE poll set, bytes are taken off a
// socket only when poll says they are there (never a blocking recv), and a request is handled -- one at a time, like the
// reference, which never calls setThreadNum -- as soon as all of it is buffered.  A client that stalls in the middle of a request is
// answered 408 and closed once its request is older than the request timeout (10 s unless set_request_timeout_ms says otherwise), an idle keep-alive connection is closed after IDLE_TIMEOUT_MS; neither keeps
// any other client waiting, and stop() is seen within one poll interval whatever the clients do.
size_t HttpListener::serve(size_t max_requests) {
    using Clock = std::chrono::steady_clock;
    constexpr int POLL_MS = 100, IDLE_TIMEOUT_MS = 30000, SEND_TIMEOUT_S = 10;
    constexpr size_t MAX_CONNECTIONS = 256;
    // last: the latest byte received or response sent (the keep-alive idle timer).  req_start: when the request now in the buffer began -- its first
    // byte, or the answer to the request before it on the same connection: a per-request DEADLINE, so a client trickling a byte every few seconds
    // gets its 408 like one that stalls outright.
    struct Conn { int fd; std::string buf; Clock::time_point last, req_start; bool continued = false; bool eof = false; };
    std::vector<Conn> conns;
    size_t served = 0;
    auto close_at = [&](size_t i) { ::close(conns[i].fd); conns.erase(conns.begin() + (long)i); };
    // answers every complete request at the front of the connection's buffer; false = close the connection
    auto drain = [&](Conn &c) -> bool {
        while (!m_Stop && (max_requests == 0 || served < max_requests)) {
            Message m;
            bool head_done = false;
            const Parse rc = try_parse(c.buf, m, m_MaxBody, head_done);
            if (rc == Parse::NeedMore) {
                if (head_done && !c.continued) {
                    const std::string *e = header(m, "expect");
                    if (e && lower(*e) == "100-continue") send_all(c.fd, "HTTP/1.1 100 Continue\r\n\r\n", 25);
                    c.continued = true;
                }
                return true;
            }
            c.continued = false;
            if (rc == Parse::TooLarge) { respond(c.fd, 413, "Payload Too Large", "{\"error\":\"body too large\"}", false); return false; }
            if (rc == Parse::Bad) { respond(c.fd, 400, "Bad Request", "{\"error\":\"malformed request\"}", false); return false; }
            // request line: METHOD SP target SP HTTP/1.x
            const size_t s1 = m.start.find(' '), s2 = m.start.rfind(' ');
            if (s1 == std::string::npos || s2 == s1 || m.start.compare(s2 + 1, 7, "HTTP/1.") != 0) {
                respond(c.fd, 400, "Bad Request", "{\"error\":\"malformed request line\"}", false);
                return false;
            }
            const std::string method = m.start.substr(0, s1);
            std::string target = m.start.substr(s1 + 1, s2 - s1 - 1);
            const size_t q = target.find('?');
            if (q != std::string::npos) target.erase(q);
            const std::string *conn = header(m, "connection");
            const bool keep = !(conn && lower(*conn) == "close") && m.start.compare(s2 + 1, 8, "HTTP/1.0") != 0;
            ++served;
            if (method != "GET" && method != "POST") { respond(c.fd, 405, "Method Not Allowed", "{\"error\":\"GET or POST\"}", keep); if (!keep) return false; continue; }
            const std::string route = target.empty() || target[0] != '/' ? target : target.substr(1);
            try {
                respond(c.fd, 200, "OK", m_Handler(method, route, m.body), keep);
            } catch (const std::out_of_range &e) {
                // wire::handle signals an unknown route this way; a missing JSON key inside a known route is the
                // client's malformed body -- Drogon would answer 500 for the escaping exception either way
                const bool unknown = std::string(e.what()).rfind("no such route", 0) == 0;
                respond(c.fd, unknown ? 404 : 500, unknown ? "Not Found" : "Internal Server Error", std::string("{\"error\":\"") + (unknown ? "unknown route" : "bad request body") + "\"}", keep);
            } catch (const std::exception &) {
                respond(c.fd, 500, "Internal Server Error", "{\"error\":\"handler failed\"}", keep);
            }
            c.last = c.req_start = Clock::now();
            if (!keep) return false;
        }
        return true;
    };
    while (!m_Stop && (max_requests == 0 || served < max_requests)) {
        std::vector<pollfd> set(1 + conns.size());
        set[0] = pollfd{m_Fd, (short)(conns.size() < MAX_CONNECTIONS ? POLLIN : 0), 0};
        for (size_t i = 0; i < conns.size(); ++i) set[1 + i] = pollfd{conns[i].fd, POLLIN, 0};
        const int pr = ::poll(set.data(), (nfds_t)set.size(), POLL_MS);     // wakes up to look at the stop flag and the timeouts
        if (pr < 0 && errno != EINTR) break;
        const auto now = Clock::now();
        for (size_t i = conns.size(); i-- > 0;) {                            // backwards: close_at() erases
            Conn &c = conns[i];
            const short ev = pr > 0 ? set[1 + i].revents : 0;
            bool alive = true;
            if (ev & POLLIN) {
                char tmp[65536];
                const ssize_t r = ::recv(c.fd, tmp, sizeof tmp, MSG_DONTWAIT);
                if (r > 0) { if (c.buf.empty()) c.req_start = now; c.buf.append(tmp, (size_t)r); c.last = now; alive = drain(c); }
                else if (r == 0) alive = false;                              // the peer closed; a partial request dies with it
                else if (errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR) alive = false;
            } else if (ev & (POLLERR | POLLHUP | POLLNVAL)) {
                alive = false;
            }
            if (alive) {
                const auto quiet = std::chrono::duration_cast<std::chrono::milliseconds>(now - c.last).count();
                const auto pending = std::chrono::duration_cast<std::chrono::milliseconds>(now - c.req_start).count();
                if (!c.buf.empty() && pending > m_RequestTimeoutMs) { respond(c.fd, 408, "Request Timeout", "{\"error\":\"request not completed in time\"}", false); alive = false; }
                else if (c.buf.empty() && quiet > IDLE_TIMEOUT_MS) alive = false;
            }
            if (!alive) close_at(i);
            if (m_Stop || (max_requests && served >= max_requests)) break;
        }
        if (pr > 0 && (set[0].revents & POLLIN) && !m_Stop) {
            const int c = ::accept(m_Fd, nullptr, nullptr);
            if (c >= 0) {
                int one = 1;
                ::setsockopt(c, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
                // a client that stops READING its (possibly multi-megabyte) response must not hold the one serving thread in send():
                // the kernel gives up on a blocked send after SEND_TIMEOUT_S, send_all() fails, the connection is closed
                timeval tv{SEND_TIMEOUT_S, 0};
                ::setsockopt(c, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
                conns.push_back(Conn{c, std::string(), Clock::now(), Clock::now()});
            }
        }
    }
    for (const Conn &c : conns) ::close(c.fd);
    return served;
}

HttpTransport::HttpTransport(const std::string &host, uint16_t port) : m_Host(host), m_Port(port) {}
HttpTransport::~HttpTransport() { if (m_Fd >= 0) ::close(m_Fd); }

std::string HttpTransport::get(const std::string &route) { return request("GET", route, std::string()); }
std::string HttpTransport::post(const std::string &route, const std::string &body) { return request("POST", route, body); }

std::string HttpTransport::request(const char *method, const std::string &route, const std::string &body) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (m_Fd < 0) {
            addrinfo hints{}, *res = nullptr;
            hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
            if (::getaddrinfo(m_Host.c_str(), std::to_string(m_Port).c_str(), &hints, &res) != 0 || !res) throw std::runtime_error("HttpTransport: cannot resolve " + m_Host);
            m_Fd = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);
            const bool ok = m_Fd >= 0 && ::connect(m_Fd, res->ai_addr, res->ai_addrlen) == 0;
            ::freeaddrinfo(res);
            if (!ok) { if (m_Fd >= 0) ::close(m_Fd); m_Fd = -1; throw std::runtime_error("HttpTransport: cannot connect to " + m_Host + ":" + std::to_string(m_Port)); }
            int one = 1;
            ::setsockopt(m_Fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        }
        std::string head = std::string(method) + " /" + route + " HTTP/1.1\r\nHost: " + m_Host + ":" + std::to_string(m_Port) +
                           "\r\nAccept: */*\r\nConnection: keep-alive\r\n";
        if (std::strcmp(method, "POST") == 0) head += "Content-Type: application/json\r\nContent-Length: " + std::to_string(body.size()) + "\r\n";
        head += "\r\n";
        Message m;
        std::string buf;
        if (send_all(m_Fd, head.data(), head.size()) && send_all(m_Fd, body.data(), body.size()) &&
            read_message(m_Fd, buf, m, (size_t)1 << 34, nullptr) == Recv::Ok) {
            last_status = m.start.size() >= 12 ? std::atoi(m.start.c_str() + 9) : 0;
            bytes_sent += body.size(); bytes_received += m.body.size();
            const std::string *conn = header(m, "connection");
            if (conn && lower(*conn) == "close") { ::close(m_Fd); m_Fd = -1; }
            if (last_status != 200) throw std::runtime_error("HttpTransport: " + std::string(method) + " /" + route + " answered " + m.start);
            return std::move(m.body);
        }
        ::close(m_Fd); m_Fd = -1;                                    // a keep-alive connection the server has closed: once more on a fresh one
    }
    throw std::runtime_error("HttpTransport: no answer from " + m_Host + ":" + std::to_string(m_Port));
}

}  // namespace wire
