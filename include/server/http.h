// http.h -- plain POSIX-socket HTTP/1.1 transport for the PreFHEtch routes.
//
// The reference serves its four routes through Drogon (/root/reference/src/server/server_lib.cpp:48-53 listens on
// SERVER_ADDRESS:SERVER_PORT; /root/reference/src/server/controllers/Query.h:14,21,26,31 declares GET /query and
// POST /coarsesearch, /precisesearch, /precise-vector-pir) and its client talks to them with cpr (libcurl)
// (/root/reference/src/client/client_lib.cpp:43,109,179,231).  Neither library is available to this build, and neither
// is needed: a listener that speaks enough HTTP/1.1 for libcurl -- request line, headers, Content-Length bodies,
// "Expect: 100-continue", keep-alive -- hands every request to wire::handle() and writes the body back as
// application/json.  Like the reference (which never calls setThreadNum) it handles one request at a time, but like Drogon's
// event loop it keeps MANY connections open at once: one poll set over the listening socket and every client socket, no
// blocking read anywhere -- an idle keep-alive client or one that stalls in the middle of a request (408 after 10 s, closed)
// never keeps another client waiting, and stop() is seen within 100 ms.
//   200  handler returned a body            404  unknown route (wire::handle threw std::out_of_range for the route)
//   405  method other than GET / POST       500  the handler threw (what Drogon answers for an escaping exception)
//   400  malformed request                  413  body larger than max_body      408  request not completed within 10 s
#pragma once

#include <atomic>
#include <cstdint>
#include <functional>
#include <string>

#include "wire.h"

namespace wire {

// (method, route without the leading slash, request body) -> response body; may throw
using HttpHandler = std::function<std::string(const std::string &, const std::string &, const std::string &)>;

class HttpListener {
  public:
    // Binds and listens at once (port 0 = an ephemeral port, see port()); throws std::runtime_error on failure.
    HttpListener(HttpHandler handler, const std::string &address, uint16_t port, size_t max_body = (size_t)1 << 30);
    HttpListener(Server &server, const std::string &address, uint16_t port);       // the routes of wire::handle
    ~HttpListener();
    HttpListener(const HttpListener &) = delete;
    HttpListener &operator=(const HttpListener &) = delete;
    uint16_t port() const { return m_Port; }
    // Accepts connections and serves their requests until stop() is called (from another thread or a handler) or, when
    // max_requests > 0, that many requests have been answered.  Returns the number of requests answered.
    size_t serve(size_t max_requests = 0);
    void stop();
    // A request that is not complete this long after its first byte is answered 408 and its connection closed (default 10 s) -- a deadline per
    // request, not an idle timer between bytes.  Call before serve().
    void set_request_timeout_ms(int ms) { m_RequestTimeoutMs = ms; }

  private:
    HttpHandler m_Handler;
    int m_Fd = -1;
    uint16_t m_Port = 0;
    size_t m_MaxBody;
    int m_RequestTimeoutMs = 10000;
    std::atomic<bool> m_Stop{false};
};

// Client side of the same wire: GET / POST over one keep-alive connection (reconnects when the server closed it).
class HttpTransport : public Transport {
  public:
    HttpTransport(const std::string &host, uint16_t port);
    ~HttpTransport() override;
    std::string get(const std::string &route) override;
    std::string post(const std::string &route, const std::string &body) override;
    int last_status = 0;
    size_t bytes_sent = 0, bytes_received = 0;        // request / response body bytes so far

  private:
    std::string request(const char *method, const std::string &route, const std::string &body);
    std::string m_Host;
    uint16_t m_Port;
    int m_Fd = -1;
};

}  // namespace wire
