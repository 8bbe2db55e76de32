// tables.hpp -- host-side construction of the per-modulus NTT tables (product code; shares nothing
// with oracle/).  Follows the published construction of SEAL's NTTTables (util/ntt.cpp) and
// try_minimal_primitive_root (util/numth.cpp), which the reference links un-vendored
// (/root/reference/CMakeLists.txt:33-38): psi = smallest primitive 2N-th root of unity mod q,
// table entry j = psi^bitrev(j) with its Shoup quotient.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "ntt_core.hpp"

namespace pf {

typedef unsigned __int128 u128_t;

inline uint64_t h_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128_t)a * b) % q); }
inline uint64_t h_powmod(uint64_t b, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    b %= q;
    for (; e; e >>= 1) { if (e & 1) r = h_mulmod(r, b, q); b = h_mulmod(b, b, q); }
    return r;
}
inline bool h_is_prime(uint64_t n) {
    if (n < 2) return false;
    const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (uint64_t p : bases) if (n % p == 0) return n == p;
    uint64_t d = n - 1; int s = 0;
    while (!(d & 1)) { d >>= 1; ++s; }
    for (uint64_t a : bases) {
        uint64_t x = h_powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool witness = true;
        for (int r = 1; r < s && witness; ++r) { x = h_mulmod(x, x, n); if (x == n - 1) witness = false; }
        if (witness) return false;
    }
    return true;
}
inline uint32_t h_bitrev(uint32_t x, int bits) { uint32_t r = 0; for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

struct LimbTables {
    uint64_t q = 0, psi = 0, n_inv = 0;
    uint64_t ratio0 = 0, ratio1 = 0;              // floor(2^128/q)
    bool f64_ok = false;                          // q small enough for the exact-FP64 back-end
    std::vector<TwU64> fwd_u, inv_u;              // N entries each, entry 0 of inv_u = N^-1, entry 1 = psi^-bitrev(1)*N^-1
    std::vector<TwF64> fwd_f, inv_f;              // the same twiddles as doubles (8 bytes per entry), then 64 quotients
};

// Bound for ArithF64 (see ntt_core.hpp): R*q <= 2^50 (R = coefficients per thread: an inverse pass doubles R/2.. times
// between re-centrings) and (1+LOGN)*q <= 2^50.
inline bool f64_path_ok(uint64_t q, int logn) { return logn <= 15 && q < (1ull << (logn >= 15 ? 44 : 45)); }

inline bool build_limb_tables(uint32_t N, uint64_t q, LimbTables &t, std::string &err) {
    int logn = 0;
    while ((1u << logn) < N) ++logn;
    if (N < 2 || (1u << logn) != N) { err = "N must be a power of two"; return false; }
    if (q < 2 || (q >> 61)) { err = "modulus must be below 2^61"; return false; }
    if (!h_is_prime(q)) { err = "modulus is not prime"; return false; }
    if ((q - 1) % (2ull * N)) { err = "modulus is not 1 mod 2N"; return false; }
    // any primitive 2N-th root, then the smallest among its odd powers
    uint64_t root = 0;
    const uint64_t cof = (q - 1) / (2ull * N);
    for (uint64_t x = 2; x < 1000000 && !root; ++x) {
        const uint64_t g = h_powmod(x, cof, q);
        if (h_powmod(g, N, q) == q - 1) root = g;
    }
    if (!root) { err = "no primitive 2N-th root found"; return false; }
    {
        const uint64_t sq = h_mulmod(root, root, q);
        uint64_t cur = root, best = root;
        for (uint32_t i = 0; i < N; ++i) { if (cur < best) best = cur; cur = h_mulmod(cur, sq, q); }
        root = best;
    }
    t.q = q; t.psi = root;
    t.n_inv = h_powmod(N % q, q - 2, q);
    const u128_t ratio = (~(u128_t)0) / q;
    t.ratio0 = (uint64_t)ratio; t.ratio1 = (uint64_t)(ratio >> 64);
    t.f64_ok = f64_path_ok(q, logn);
    std::vector<uint64_t> fw(N), iw(N);
    const uint64_t psi_inv = h_powmod(root, q - 2, q);
    uint64_t p = 1, ip = 1;
    for (uint32_t i = 0; i < N; ++i) {
        const uint32_t r = h_bitrev(i, logn);
        fw[r] = p; iw[r] = ip;
        p = h_mulmod(p, root, q); ip = h_mulmod(ip, psi_inv, q);
    }
    iw[0] = t.n_inv;
    if (N > 1) iw[1] = h_mulmod(iw[1], t.n_inv, q);
    auto shoup = [q](uint64_t w) { return (uint64_t)((((u128_t)w) << 64) / q); };
    t.fwd_u.resize(N); t.inv_u.resize(N);
    for (uint32_t j = 0; j < N; ++j) { t.fwd_u[j] = TwU64{fw[j], shoup(fw[j])}; t.inv_u[j] = TwU64{iw[j], shoup(iw[j])}; }
    t.fwd_f.clear(); t.inv_f.clear();
    if (t.f64_ok) {
        t.fwd_f.resize(N); t.inv_f.resize(N);
        for (uint32_t j = 0; j < N; ++j) {
            t.fwd_f[j] = TwF64{(double)fw[j]};
            t.inv_f[j] = TwF64{(double)iw[j]};
        }
        // entries N .. N+63: the quotient estimates fl(w * fl(1/q)) of entries 0 .. 63 (R <= 64 registers per thread), exactly as the device
        // would compute them, for the workgroup-uniform twiddles of pass 0 (ntt_core.hpp, PassTw)
        const double qinv = 1.0 / (double)q;
        for (uint32_t j = 0; j < 64; ++j) {
            t.fwd_f.push_back(TwF64{j < N ? (double)fw[j] * qinv : 0.0});
            t.inv_f.push_back(TwF64{j < N ? (double)iw[j] * qinv : 0.0});
        }
    }
    return true;
}

}  // namespace pf
