// client_server_utils.h -- shared constants, dataset reader and timer of the PreFHEtch server,
// source-compatible with the reference header of the same path
// (/root/reference/include/common/client_server_utils.h:10-67: same names, values, signatures and
// error behaviour) but free of third-party includes: the reference pulls in spdlog only to log the
// fopen failure, which goes to stderr here.
#pragma once

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

// Problem shape of the reference deployment (SIFT10K); reference :10-20.  These are template arguments
// of every Server signature, so they are part of the ABI.
constexpr int64_t PRECISE_VECTOR_DIMENSIONS = 128;
constexpr int64_t NPROBE = 20;          // inverted lists probed per query
constexpr int64_t COARSE_PROBE = 200;   // candidates re-ranked with exact distances
constexpr int64_t K = 100;              // results returned
constexpr int64_t NBASE = 10000;
constexpr int64_t NQUERY = 5;
constexpr int64_t NLIST = 256;          // IVF centroids
constexpr int64_t SUB_QUANTIZERS = 32;
constexpr int64_t SUB_QUANTIZER_SIZE = 8;

using faiss_idx_t = int64_t;            // reference :22

// Reads an .fvecs / .ivecs file: every row is an int32 dimension d followed by d 4-byte values
// (reference :24-56).  On return vecs holds n*d values row-major.  Error behaviour follows the reference:
// a file that cannot be opened aborts the process; a malformed header or size is an assertion-class
// failure (abort as well, also in release builds).
template <typename T>
void vecs_read(const char *fname, size_t &d_out, size_t &n_out, std::vector<T> &vecs) {
    static_assert(sizeof(T) == 4, "fvecs/ivecs hold 4-byte elements");
    std::ifstream in(fname, std::ios::binary | std::ios::ate);
    if (!in) {
        std::fprintf(stderr, "could not open %s\n", fname);
        std::perror("");
        std::abort();
    }
    const std::streamoff bytes = in.tellg();
    in.seekg(0);
    int32_t d = 0;
    in.read(reinterpret_cast<char *>(&d), sizeof d);
    const std::streamoff row_bytes = (static_cast<std::streamoff>(d) + 1) * 4;
    if (!in || d <= 0 || d >= 1000000 || bytes % row_bytes != 0) {
        std::fprintf(stderr, "%s: not a vecs file (d=%d, %lld bytes)\n", fname, d, static_cast<long long>(bytes));
        std::abort();
    }
    const size_t n = static_cast<size_t>(bytes / row_bytes);
    std::vector<char> raw(static_cast<size_t>(bytes));
    in.seekg(0);
    in.read(raw.data(), bytes);
    if (in.gcount() != bytes) {
        std::fprintf(stderr, "%s: short read\n", fname);
        std::abort();
    }
    vecs.resize(n * static_cast<size_t>(d));
    for (size_t i = 0; i < n; ++i)                       // drop the per-row dimension header
        std::memcpy(vecs.data() + i * d, raw.data() + i * row_bytes + 4, static_cast<size_t>(d) * 4);
    d_out = static_cast<size_t>(d);
    n_out = n;
}

// Wall-clock span with the reference's interface (reference :58-67, src/common/client_server_utils.cpp:3-24).
class Timer {
  public:
    void StartTimer() { m_TimerStart = std::chrono::high_resolution_clock::now(); }
    void StopTimer() { m_TimerEnd = std::chrono::high_resolution_clock::now(); }
    void getDuration(long long &time_micro, long long &time_milli) const {
        time_micro = std::chrono::duration_cast<std::chrono::microseconds>(m_TimerEnd - m_TimerStart).count();
        time_milli = static_cast<long long>(time_micro * 0.001);
    }

  private:
    std::chrono::time_point<std::chrono::high_resolution_clock> m_TimerStart;
    std::chrono::time_point<std::chrono::high_resolution_clock> m_TimerEnd;
};
