// pf_ks_kernels.hpp -- element-wise kernels of key switching (steps 2 and 4 of SEAL's
// Evaluator::switch_key_inplace; step 1, the digit NTTs, is k_ks_ntt in pf_ntt_kernels.hpp and step 3 the plain
// inverse NTT).  The reference links SEAL un-vendored and never calls it (/root/reference/CMakeLists.txt:33-38);
// BASELINE config 5 names the path.
#pragma once
#include "pf_ntt_kernels.hpp"

namespace pf {

struct KsArgs {
    const LimbDev *limbs;
    const uint64_t *x;       // [Bs][D][K][N] digit NTTs (step 1)
    const uint64_t *ksk;     // [D][2][K][N]  key, NTT form
    uint64_t *acc;           // [Bs][2][K][N] products: NTT form after step 2, coefficient form after the inverse NTT
    uint64_t *ct;            // [Bs][2][D][N] ciphertext the switched polynomial is added into (step 4)
    uint32_t D, K, logn;
    uint32_t nb;             // ciphertexts in this round of the workspace (Bs)
};

// step 2: acc[b][c][J] = sum_I x[b][I][J] . ksk[I][c][J]  (mod m_J), accumulated in 128 bits like SEAL's lazy sum
// (D <= 63 summands of < 2^122 each), reduced once.  One block per (b, J, 2048-coefficient chunk), 16 B per lane.
// Block order: the key slice of a (J, chunk) unit is what the nb ciphertexts of the round share (the key is 126 MB at config 5,
// four times the L2s together), so XCD x (blocks x, x + 8, ...) takes the units u = x (mod 8) and runs a unit's nb ciphertexts
// back to back: the slice comes from memory once and from that XCD's L2 for the other nb - 1.  (With b outermost every
// ciphertext re-read the whole key from the Infinity Cache.)
__global__ void __launch_bounds__(256) k_ks_mac(KsArgs p) {
    const uint32_t chunk_log = p.logn < 11 ? p.logn : 11;
    const uint32_t chunks = 1u << (p.logn - chunk_log);
    const uint32_t xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const uint32_t unit = (seq / p.nb) * 8 + xcd;               // J * chunks + chunk
    if (unit >= p.K * chunks) return;
    const size_t b = seq % p.nb;
    const uint32_t ch = unit % chunks, J = unit / chunks;
    const LimbDev &lm = p.limbs[J];
    const ArithU64 ar{lm.q, lm.two_q, lm.ratio0, lm.ratio1};
    const size_t N = (size_t)1 << p.logn;
    const uint32_t per_thread = (1u << chunk_log) / 512;
    for (uint32_t jj = 0; jj < per_thread; ++jj) {
        const size_t n2 = ((size_t)ch << chunk_log) / 2 + jj * 256 + threadIdx.x;      // index of a coefficient PAIR
        uint64_t lo[2][2] = {{0, 0}, {0, 0}}, hi[2][2] = {{0, 0}, {0, 0}};            // [component][element]
#pragma unroll 5                      // five digits' loads (15 x 16 bytes per lane) in flight: the kernel is bound by memory latency, not arithmetic
        for (uint32_t I = 0; I < p.D; ++I) {
            const ulonglong2 xv = reinterpret_cast<const ulonglong2 *>(p.x + ((b * p.D + I) * p.K + J) * N)[n2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const ulonglong2 kv = reinterpret_cast<const ulonglong2 *>(p.ksk + (((size_t)I * 2 + c) * p.K + J) * N)[n2];
                const uint64_t xs[2] = {xv.x, xv.y}, ks[2] = {kv.x, kv.y};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const uint64_t pl = xs[e] * ks[e], ph = mulhi64(xs[e], ks[e]);
                    const uint64_t s = lo[c][e] + pl;
                    hi[c][e] += ph + (s < pl ? 1 : 0);
                    lo[c][e] = s;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            ulonglong2 o;
            const uint64_t r0 = ar.barrett128(lo[c][0], hi[c][0]), r1 = ar.barrett128(lo[c][1], hi[c][1]);
            o.x = r0 >= lm.q ? r0 - lm.q : r0;
            o.y = r1 >= lm.q ? r1 - lm.q : r1;
            reinterpret_cast<ulonglong2 *>(p.acc + ((b * 2 + c) * p.K + J) * N)[n2] = o;
        }
    }
}

// step 4: modulus switching with rounding.  With s_J = acc[b][c][J] and s_P = acc[b][c][K-1] in coefficient form:
//   t = (s_P + floor(P/2)) mod P;   ct[b][c][J] += P^-1 * (s_J - (t mod q_J) + (floor(P/2) mod q_J))   (mod q_J)
__global__ void __launch_bounds__(256) k_ks_moddown(KsArgs p) {
    const uint32_t chunk_log = p.logn < 11 ? p.logn : 11;
    const uint32_t chunks = 1u << (p.logn - chunk_log);
    const uint32_t ch = blockIdx.x % chunks;
    const size_t bcj = blockIdx.x / chunks;
    const uint32_t J = (uint32_t)(bcj % p.D);
    const size_t bc = bcj / p.D;                          // b * 2 + component
    const LimbDev &lm = p.limbs[J];
    const LimbDev &lp = p.limbs[p.K - 1];
    const uint64_t q = lm.q, P = lp.q, half = P >> 1;
    const size_t N = (size_t)1 << p.logn;
    const uint32_t per_thread = (1u << chunk_log) / 512;
    const ulonglong2 *sj = reinterpret_cast<const ulonglong2 *>(p.acc + (bc * p.K + J) * N);
    const ulonglong2 *sp = reinterpret_cast<const ulonglong2 *>(p.acc + (bc * p.K + (p.K - 1)) * N);
    ulonglong2 *ct = reinterpret_cast<ulonglong2 *>(p.ct + (bc * p.D + J) * N);
    auto one = [&](uint64_t s_j, uint64_t s_p, uint64_t c_in) {
        uint64_t t = s_p + half;
        t = t >= P ? t - P : t;
        uint64_t tj = t - mulhi64(t, lm.ratio1) * q;                      // t mod q_J (barrett_reduce_64)
        tj = tj >= q ? tj - q : tj;
        uint64_t v = s_j + (q - tj) + lm.ks_half_mod;                      // < 3q
        v = v >= 2 * q ? v - 2 * q : v;
        v = v >= q ? v - q : v;
        uint64_t r = v * lm.ks_pinv - mulhi64(v, lm.ks_pinv_quot) * q;     // Shoup product, [0, 2q)
        r = r >= q ? r - q : r;
        r += c_in;
        return r >= q ? r - q : r;
    };
    for (uint32_t jj = 0; jj < per_thread; ++jj) {
        const size_t n2 = ((size_t)ch << chunk_log) / 2 + jj * 256 + threadIdx.x;
        const ulonglong2 a = sj[n2], b = sp[n2], c = ct[n2];
        ulonglong2 o;
        o.x = one(a.x, b.x, c.x);
        o.y = one(a.y, b.y, c.y);
        ct[n2] = o;
    }
}

// ---- plaintext packing for the encrypted inner products (pf_pack_rows) ---------------------------------
// One thread per output coefficient: coefficient c of polynomial p, limb l.  Row j > 0 occupies coefficients
// d*j - i (i < d), row 0 occupies coefficient 0 and, negated, the top coefficients N - i (X^-i = -X^(N-i)).
struct PackArgs {
    const LimbDev *limbs;
    const float *xb; const int64_t *ids; uint64_t *out;
    size_t nb; uint32_t d, L, logn, rows_per_poly;
};

__global__ void __launch_bounds__(256) k_pack_rows(PackArgs p) {
    const uint32_t N = 1u << p.logn, per = N / 256;
    const size_t b = blockIdx.x;
    const uint32_t c = (uint32_t)(b % per) * 256 + threadIdx.x;
    const size_t pl = b / per;                                   // polynomial * L + limb
    const uint32_t l = (uint32_t)(pl % p.L);
    const size_t poly = pl / p.L;
    uint32_t j = 0, i = 0;
    bool neg = false, live = true;
    if (c != 0) {
        const uint32_t jj = (c + p.d - 1) / p.d;
        if (jj < p.rows_per_poly) { j = jj; i = jj * p.d - c; }
        else if (N - c < p.d) { j = 0; i = N - c; neg = true; }
        else live = false;
    }
    int64_t v = 0;
    if (live) {
        const int64_t id = p.ids[poly * p.rows_per_poly + j];
        if (id >= 0 && (size_t)id < p.nb) v = (int64_t)rintf(p.xb[(size_t)id * p.d + i]);
    }
    if (neg) v = -v;
    const uint64_t q = p.limbs[l].q;
    p.out[pl * N + c] = v >= 0 ? (uint64_t)v : q - (uint64_t)(-v);
}

// ---- Galois automorphism X -> X^g on coefficient-form polynomials (pf_apply_galois) ---------------------------
// One thread per OUTPUT coefficient c: the source is i = c * g^-1 mod 2N (g^-1 mod 2N from the host), taken negated when
// i >= N (then (i - N) * g = c + N mod 2N).  Writes are contiguous, reads follow the permutation: the round-2 form (thread per
// input, scattered 8-byte writes) ran at 1.7 TB/s on the private retrieval's batches -- strided stores cost several times what
// strided loads do here (tools/ubench_stride.hip).  16 B of traffic per coefficient.
struct GaloisArgs {
    const LimbDev *limbs;
    const uint64_t *in; uint64_t *out;
    uint32_t L, logn, galois_inv;
    uint64_t *target;            // ciphertext form (pf_apply_galois_ct): the permuted second components go here, zeros to `out`
};

__global__ void __launch_bounds__(256) k_apply_galois(GaloisArgs p) {
    const uint32_t N = 1u << p.logn, per = N / 256;
    const size_t poly = blockIdx.x / per;
    const uint32_t c = (uint32_t)(blockIdx.x % per) * 256 + threadIdx.x;
    const uint32_t l = (uint32_t)(poly % p.L);
    const uint64_t q = p.limbs[l].q;
    const uint32_t i = (uint32_t)(((uint64_t)c * p.galois_inv) & (2u * N - 1));
    const uint64_t v = p.in[poly * N + (i & (N - 1))];
    uint64_t *dst = p.out + poly * N;
    if (p.target) {                                            // poly = (b * 2 + component) * L + l
        const size_t bc = poly / p.L;
        if (bc & 1) {
            dst[c] = 0;
            dst = p.target + ((bc >> 1) * p.L + l) * N;
        }
    }
    dst[c] = (i >= N && v) ? q - v : v;
}

// ---- sum and shifted difference in one pass (pf_poly_addsub_monomial): sum = a + b, diff = (a - b) * X^k ------------------
// The butterfly of SealPIR's query expansion.  One thread per coefficient; sum may alias a or b.  32 B of traffic per coefficient.
struct AddSubArgs {
    const LimbDev *limbs;
    const uint64_t *a, *b; uint64_t *sum, *diff;
    uint32_t L, logn, shift;
};

__global__ void __launch_bounds__(256) k_addsub_monomial(AddSubArgs p) {
    const uint32_t N = 1u << p.logn, per = N / 256;
    const size_t poly = blockIdx.x / per;
    const uint32_t c = (uint32_t)(blockIdx.x % per) * 256 + threadIdx.x;
    const uint64_t q = p.limbs[poly % p.L].q;
    const uint64_t x = p.a[poly * N + c], y = p.b[poly * N + c];
    const uint64_t s = x + y, d = x >= y ? x - y : x + q - y;
    p.sum[poly * N + c] = s >= q ? s - q : s;
    const uint32_t j = (c + p.shift) & (2u * N - 1);
    p.diff[poly * N + (j & (N - 1))] = (j >= N && d) ? q - d : d;
}

// ---- product with a monomial X^k, k in [0, 2N) (pf_poly_mul_monomial; SEAL util::negacyclic_shift_poly_coeffmod) ----------
// One thread per OUTPUT coefficient: out[c] = in[c - k] for c >= k (mod 2N bookkeeping: an index that wraps past N once picks
// up a minus sign).  Reads and writes are both contiguous.  16 B of traffic per coefficient.
struct MonoArgs {
    const LimbDev *limbs;
    const uint64_t *in; uint64_t *out;
    uint32_t L, logn, shift;
};

__global__ void __launch_bounds__(256) k_mul_monomial(MonoArgs p) {
    const uint32_t N = 1u << p.logn, per = N / 256;
    const size_t poly = blockIdx.x / per;
    const uint32_t c = (uint32_t)(blockIdx.x % per) * 256 + threadIdx.x;
    const uint64_t q = p.limbs[poly % p.L].q;
    const uint32_t i = (c - p.shift) & (2u * N - 1);           // source index in [0, 2N): X^i * X^k = X^c up to the sign of X^N = -1
    const uint64_t v = p.in[poly * N + (i & (N - 1))];
    p.out[poly * N + c] = (i >= N && v) ? q - v : v;
}

// ---- sum of ciphertext x plaintext products, everything in NTT form (pf_ct_pt_dot) --------------------------------------
// out[g] = sum over the plaintexts p of chunk g (p in [g * chunk, min((g + 1) * chunk, n_pt))) of ct[p % n_ct] . pt[p]; chunk divides
// n_ct, so a chunk never wraps.  The private retrieval's answer: sum_k selection_k x database_k per column, split 16 ways along k
// for parallelism.  One block per (g, limb, 512 coefficients), both components per thread (they share the plaintext value), 128-bit
// lazy sums reduced once like k_ks_mac (the host checks chunk * q^2 < 2^128).  Streams ct and pt once: 40 B per coefficient and term.
struct DotArgs {
    const LimbDev *limbs;
    const uint64_t *ct, *pt; uint64_t *out;
    uint32_t L, logn;
    uint64_t n_ct, n_pt, chunk;
};

__global__ void __launch_bounds__(256) k_ct_pt_dot(DotArgs p) {
    const size_t N = (size_t)1 << p.logn;
    const uint32_t cpp = (uint32_t)(N / 512);
    const uint32_t ch = blockIdx.x % cpp, limb = (blockIdx.x / cpp) % p.L;
    const size_t g = blockIdx.x / (cpp * p.L);
    const LimbDev &lm = p.limbs[limb];
    const ArithU64 ar{lm.q, lm.two_q, lm.ratio0, lm.ratio1};
    const size_t n2 = (size_t)ch * 256 + threadIdx.x, half = N / 2;                       // index of a coefficient PAIR
    const size_t p0 = g * p.chunk, p1 = p0 + p.chunk < p.n_pt ? p0 + p.chunk : p.n_pt, k0 = p0 % p.n_ct;
    const ulonglong2 *pt2 = reinterpret_cast<const ulonglong2 *>(p.pt) + ((p0 * p.L + limb) * half + n2);
    const ulonglong2 *c2 = reinterpret_cast<const ulonglong2 *>(p.ct) + ((k0 * 2 * p.L + limb) * half + n2);
    const size_t pt_step = (size_t)p.L * half, ct_step = 2 * (size_t)p.L * half, comp = (size_t)p.L * half;
    uint64_t lo[2][2] = {{0, 0}, {0, 0}}, hi[2][2] = {{0, 0}, {0, 0}};                    // [component][element]
#pragma unroll 4
    for (size_t j = 0; j < p1 - p0; ++j) {
        const ulonglong2 pv = pt2[j * pt_step], a = c2[j * ct_step], b = c2[j * ct_step + comp];
        const uint64_t ps[2] = {pv.x, pv.y}, xs[2][2] = {{a.x, a.y}, {b.x, b.y}};
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint64_t pl = xs[c][e] * ps[e], ph = mulhi64(xs[c][e], ps[e]);
                const uint64_t s = lo[c][e] + pl;
                hi[c][e] += ph + (s < pl ? 1 : 0);
                lo[c][e] = s;
            }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        ulonglong2 o;
        const uint64_t r0 = ar.barrett128(lo[c][0], hi[c][0]), r1 = ar.barrett128(lo[c][1], hi[c][1]);
        o.x = r0 >= lm.q ? r0 - lm.q : r0;
        o.y = r1 >= lm.q ? r1 - lm.q : r1;
        reinterpret_cast<ulonglong2 *>(p.out)[((g * 2 + c) * p.L + limb) * half + n2] = o;
    }
}

}  // namespace pf
