/*
 * pf_oracle.c -- CPU restatement ("Oracle A") of the PreFHEtch server-side hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the shipped library (libprefhetch_hip.so) never
 * links, loads or calls anything in oracle/.
 *
 * PARITY UNPINNED.  The reference (PES-Innovation-Lab/PreFHEtch @ 2025-08-01) links Microsoft
 * SEAL @ 7a931d55ba84a40b85938f6ca3ac206f18654093 (CMakeLists.txt:33-38,66) and the fork
 * PreFHEtch-faiss @ 49c5b57c759e06c69447fea2342116fa521f4083 (CMakeLists.txt:22-26), but neither
 * is vendored, neither is in this container, SEAL has zero call sites in the reference, and the
 * reference holds no tests, golden vectors or fixtures.  What follows
 * restates the PUBLISHED algorithms of those libraries:
 *   - negacyclic NTT with Harvey lazy butterflies and Shoup ("MultiplyUIntModOperand") twiddles,
 *     Cooley-Tukey natural->bit-reversed forward, Gentleman-Sande bit-reversed->natural inverse
 *     with N^-1 merged into the last layer (SEAL util/ntt.cpp, util/dwthandler.h;
 *     Longa-Naehrig 2016 Alg. 1/2; Harvey 2014);
 *   - minimal primitive 2N-th root of unity (SEAL util/numth.cpp try_minimal_primitive_root);
 *   - 128->64 bit Barrett dyadic product with const_ratio = floor(2^128/q)
 *     (SEAL util/polyarithsmallmod.cpp dyadic_product_coeffmod);
 *   - add/sub/negate with one conditional subtract (SEAL add_poly_coeffmod etc.);
 *   - Evaluator::multiply_plain on a coefficient-form ciphertext with an NTT-form plaintext:
 *     per (poly, limb): forward NTT, dyadic product, inverse NTT;
 *   - IndexFlatL2::search semantics (squared L2, ascending, int64 labels); tie order is
 *     undefined in faiss, this build fixes (distance, then smaller id).
 * First-party reference code that IS readable and is followed literally:
 *   - Server::preciseSearch      /root/reference/src/server/server_lib.cpp:140-167
 *   - sort_nearest_centroids     /root/reference/src/client/client_lib.cpp:50-81
 *   - Server::preciseVectorPIR   /root/reference/src/server/server_lib.cpp:169-196
 * The oracle is pinned instead by (1) an independent big-integer Python restatement
 * (oracle/bigint_ref.py, O(N^2) evaluation of the mathematical definition), (2) the prime /
 * minimal-root table of SURVEY.md section 8c, (3) the committed fixtures under tests/golden/.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ modular helpers */
static inline uint64_t mulmod_u128(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }

static uint64_t powmod(uint64_t b, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    b %= q;
    while (e) {
        if (e & 1) r = mulmod_u128(r, b, q);
        b = mulmod_u128(b, b, q);
        e >>= 1;
    }
    return r;
}

static uint64_t invmod(uint64_t a, uint64_t q) { return powmod(a, q - 2, q); } /* q prime */

static uint32_t bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

int pfo_is_prime(uint64_t n) {
    if (n < 2) return 0;
    static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (unsigned i = 0; i < 12; i++) { if (n % small[i] == 0) return n == small[i]; }
    uint64_t d = n - 1; int s = 0;
    while (!(d & 1)) { d >>= 1; s++; }
    for (unsigned i = 0; i < 12; i++) {            /* deterministic for n < 3.3e24 */
        uint64_t x = powmod(small[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int r = 1; r < s; r++) { x = mulmod_u128(x, x, n); if (x == n - 1) { comp = 0; break; } }
        if (comp) return 0;
    }
    return 1;
}

/* SEAL numth.cpp try_minimal_primitive_root: smallest primitive (2N)-th root of unity mod q.
 * SEAL finds any primitive root by sampling and then walks the odd powers; the minimum over
 * all odd powers is independent of the starting root, so a deterministic search is equivalent. */
int pfo_min_primitive_root(uint64_t two_n, uint64_t q, uint64_t *out) {
    if ((q - 1) % two_n) return -1;
    uint64_t e = (q - 1) / two_n, root = 0;
    for (uint64_t x = 2; x < q && x < 100000; x++) {
        uint64_t g = powmod(x, e, q);
        if (powmod(g, two_n / 2, q) == q - 1) { root = g; break; }
    }
    if (!root) return -2;
    uint64_t gen_sq = mulmod_u128(root, root, q), cur = root, best = root;
    for (uint64_t i = 0; i < two_n / 2; i++) {
        if (cur < best) best = cur;
        cur = mulmod_u128(cur, gen_sq, q);
    }
    *out = best;
    return 0;
}

/* ------------------------------------------------------------------ context / tables */
typedef struct {
    uint64_t q, two_q;
    uint64_t ratio0, ratio1;       /* floor(2^128/q) low/high words (SEAL Modulus::const_ratio) */
    uint64_t psi, psi_inv, n_inv, n_inv_quot;
    /* SEAL NTTTables: root_powers[i] = psi^bitrev(i) with Shoup quotient floor(op*2^64/q) */
    uint64_t *w, *wq;              /* forward, index 1..N-1 */
    uint64_t *iw, *iwq;            /* inverse of the matching forward twiddle, same indexing */
} pfo_limb;

typedef struct {
    uint32_t N, logn, L;
    pfo_limb *limb;
} pfo_ctx;

static uint64_t shoup_quot(uint64_t op, uint64_t q) { return (uint64_t)((((u128)op) << 64) / q); }

void pfo_ctx_destroy(pfo_ctx *c) {
    if (!c) return;
    for (uint32_t l = 0; c->limb && l < c->L; l++) { free(c->limb[l].w); free(c->limb[l].wq); free(c->limb[l].iw); free(c->limb[l].iwq); }
    free(c->limb); free(c);
}

pfo_ctx *pfo_ctx_create(uint32_t N, uint32_t L, const uint64_t *moduli) {
    if (N < 2 || (N & (N - 1)) || L == 0) return NULL;
    pfo_ctx *c = (pfo_ctx *)calloc(1, sizeof(pfo_ctx));
    c->N = N; c->L = L; c->logn = 0;
    while ((1u << c->logn) < N) c->logn++;
    c->limb = (pfo_limb *)calloc(L, sizeof(pfo_limb));
    for (uint32_t l = 0; l < L; l++) {
        pfo_limb *m = &c->limb[l];
        uint64_t q = moduli[l];
        if (q >> 61 || q < 2 || !pfo_is_prime(q) || pfo_min_primitive_root(2ull * N, q, &m->psi)) { pfo_ctx_destroy(c); return NULL; }
        m->q = q; m->two_q = 2 * q;
        u128 num = ~(u128)0;                        /* floor((2^128-1)/q) == floor(2^128/q) for q not a power of 2 */
        u128 ratio = num / q;
        m->ratio0 = (uint64_t)ratio; m->ratio1 = (uint64_t)(ratio >> 64);
        m->psi_inv = invmod(m->psi, q);
        m->n_inv = invmod(N % q, q); m->n_inv_quot = shoup_quot(m->n_inv, q);
        m->w = (uint64_t *)calloc(N, 8); m->wq = (uint64_t *)calloc(N, 8);
        m->iw = (uint64_t *)calloc(N, 8); m->iwq = (uint64_t *)calloc(N, 8);
        uint64_t p = 1;
        for (uint32_t i = 0; i < N; i++) {          /* p = psi^i */
            uint32_t r = bitrev(i, c->logn);
            m->w[r] = p; m->wq[r] = shoup_quot(p, q);
            uint64_t ip = invmod(p, q);
            m->iw[r] = ip; m->iwq[r] = shoup_quot(ip, q);
            p = mulmod_u128(p, m->psi, q);
        }
    }
    return c;
}

uint64_t pfo_ctx_psi(const pfo_ctx *c, uint32_t l) { return c->limb[l].psi; }
uint64_t pfo_ctx_modulus(const pfo_ctx *c, uint32_t l) { return c->limb[l].q; }

/* ------------------------------------------------------------------ NTT (SEAL dwthandler.h) */
/* multiply_uint_mod_lazy: result in [0, 2q) */
static inline uint64_t mul_root_lazy(uint64_t y, uint64_t op, uint64_t quot, uint64_t q) {
    uint64_t hi = (uint64_t)(((u128)y * quot) >> 64);
    return y * op - hi * q;
}

/* forward: natural in -> bit-reversed out, values end in [0, q) (ntt_negacyclic_harvey) */
static void ntt_fwd_limb(const pfo_limb *m, uint32_t N, uint64_t *a) {
    const uint64_t q = m->q, two_q = m->two_q;
    uint32_t gap = N >> 1, root = 0;
    for (uint32_t mm = 1; mm < N; mm <<= 1) {
        uint32_t off = 0;
        for (uint32_t i = 0; i < mm; i++) {
            ++root;
            const uint64_t op = m->w[root], qu = m->wq[root];
            uint64_t *x = a + off, *y = x + gap;
            for (uint32_t j = 0; j < gap; j++) {
                uint64_t u = x[j] >= two_q ? x[j] - two_q : x[j];     /* guard */
                uint64_t v = mul_root_lazy(y[j], op, qu, q);
                x[j] = u + v;                                          /* [0,4q) */
                y[j] = u + two_q - v;
            }
            off += gap << 1;
        }
        gap >>= 1;
    }
    for (uint32_t j = 0; j < N; j++) {
        uint64_t v = a[j];
        if (v >= two_q) v -= two_q;
        if (v >= q) v -= q;
        a[j] = v;
    }
}

/* inverse: bit-reversed in -> natural out, scaled by N^-1, values in [0, q)
 * (inverse_ntt_negacyclic_harvey; the scalar is folded into the last layer as SEAL does) */
static void ntt_inv_limb(const pfo_limb *m, uint32_t N, uint64_t *a) {
    const uint64_t q = m->q, two_q = m->two_q;
    uint32_t gap = 1;
    for (uint32_t mm = N >> 1; mm > 1; mm >>= 1) {
        uint32_t off = 0;
        for (uint32_t i = 0; i < mm; i++) {
            const uint64_t op = m->iw[mm + i], qu = m->iwq[mm + i];
            uint64_t *x = a + off, *y = x + gap;
            for (uint32_t j = 0; j < gap; j++) {
                uint64_t u = x[j], v = y[j];
                uint64_t s = u + v;
                x[j] = s >= two_q ? s - two_q : s;                     /* guard(add) */
                y[j] = mul_root_lazy(u + two_q - v, op, qu, q);
            }
            off += gap << 1;
        }
        gap <<= 1;
    }
    {   /* last layer, mm == 1 */
        const uint64_t r = m->iw[1];
        const uint64_t sr = mulmod_u128(r, m->n_inv, q), srq = shoup_quot(sr, q);
        uint64_t *x = a, *y = a + gap;
        for (uint32_t j = 0; j < gap; j++) {
            uint64_t u = x[j] >= two_q ? x[j] - two_q : x[j], v = y[j];
            uint64_t s = u + v; if (s >= two_q) s -= two_q;
            x[j] = mul_root_lazy(s, m->n_inv, m->n_inv_quot, q);
            y[j] = mul_root_lazy(u + two_q - v, sr, srq, q);
        }
    }
    for (uint32_t j = 0; j < N; j++) { uint64_t v = a[j]; if (v >= q) v -= q; a[j] = v; }
}

/* SEAL dyadic_product_coeffmod: Barrett reduction of the 128-bit product */
static inline uint64_t barrett_mul(uint64_t x, uint64_t y, const pfo_limb *m) {
    u128 z = (u128)x * y;
    uint64_t z0 = (uint64_t)z, z1 = (uint64_t)(z >> 64);
    uint64_t carry = (uint64_t)(((u128)z0 * m->ratio0) >> 64);
    u128 t2 = (u128)z0 * m->ratio1;
    u128 s1 = (u128)(uint64_t)t2 + carry;
    uint64_t tmp1 = (uint64_t)s1;
    uint64_t tmp3 = (uint64_t)(t2 >> 64) + (uint64_t)(s1 >> 64);
    t2 = (u128)z1 * m->ratio0;
    u128 s2 = (u128)tmp1 + (uint64_t)t2;
    carry = (uint64_t)(t2 >> 64) + (uint64_t)(s2 >> 64);
    tmp1 = z1 * m->ratio1 + tmp3 + carry;
    tmp3 = z0 - tmp1 * m->q;
    return tmp3 >= m->q ? tmp3 - m->q : tmp3;
}

/* Layout convention shared with the C-ABI (include/prefhetch_hip.h): a buffer of
 * n_limb_polys polynomials of N u64 coefficients; polynomial p belongs to limb (p % L). */
#define PAR_FOR _Pragma("omp parallel for schedule(static) num_threads(nt)")
static int clamp_threads(int nthreads) {
#ifdef _OPENMP
    return nthreads > 0 ? nthreads : omp_get_max_threads();
#else
    (void)nthreads; return 1;
#endif
}

void pfo_ntt_forward(const pfo_ctx *c, uint64_t *polys, size_t n_limb_polys, int nthreads) {
    int nt = clamp_threads(nthreads); (void)nt;
    PAR_FOR
    for (long p = 0; p < (long)n_limb_polys; p++) ntt_fwd_limb(&c->limb[p % c->L], c->N, polys + (size_t)p * c->N);
}

void pfo_ntt_inverse(const pfo_ctx *c, uint64_t *polys, size_t n_limb_polys, int nthreads) {
    int nt = clamp_threads(nthreads); (void)nt;
    PAR_FOR
    for (long p = 0; p < (long)n_limb_polys; p++) ntt_inv_limb(&c->limb[p % c->L], c->N, polys + (size_t)p * c->N);
}

void pfo_dyadic_mul(const pfo_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n_limb_polys, int nthreads) {
    int nt = clamp_threads(nthreads); (void)nt;
    PAR_FOR
    for (long p = 0; p < (long)n_limb_polys; p++) {
        const pfo_limb *m = &c->limb[p % c->L];
        size_t o = (size_t)p * c->N;
        for (uint32_t j = 0; j < c->N; j++) out[o + j] = barrett_mul(a[o + j], b[o + j], m);
    }
}

/* op: 0 add, 1 sub, 2 negate(a) -- SEAL add_/sub_/negate_poly_coeffmod */
void pfo_poly_addsub(const pfo_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n_limb_polys, int op, int nthreads) {
    int nt = clamp_threads(nthreads); (void)nt;
    PAR_FOR
    for (long p = 0; p < (long)n_limb_polys; p++) {
        const uint64_t q = c->limb[p % c->L].q;
        size_t o = (size_t)p * c->N;
        for (uint32_t j = 0; j < c->N; j++) {
            uint64_t x = a[o + j], r;
            if (op == 0) { r = x + b[o + j]; if (r >= q) r -= q; }
            else if (op == 1) { uint64_t y = b[o + j]; r = x >= y ? x - y : x + q - y; }
            else { r = x ? q - x : 0; }
            out[o + j] = r;
        }
    }
}

/* Evaluator::multiply_plain restated on raw buffers.
 * ct  [B][2][L][N], pt_ntt [B or 1][L][N] (NTT form), out [B][2][L][N].
 * flags: bit0 ACCUMULATE (out += product, out must hold a valid operand in the OUTPUT domain),
 *        bit1 IN_NTT (ct already in NTT form), bit2 OUT_NTT (leave result in NTT form). */
void pfo_ct_pt_mul(const pfo_ctx *c, const uint64_t *ct, const uint64_t *pt_ntt, int pt_broadcast,
                   uint64_t *out, size_t B, int flags, int nthreads) {
    int nt = clamp_threads(nthreads); (void)nt;
    const uint32_t N = c->N, L = c->L;
    const long total = (long)(B * 2 * L);
    PAR_FOR
    for (long p = 0; p < total; p++) {
        const uint32_t l = (uint32_t)(p % L);
        const size_t b = (size_t)p / (2 * L);
        const pfo_limb *m = &c->limb[l];
        uint64_t *tmp = (uint64_t *)malloc((size_t)N * 8);
        memcpy(tmp, ct + (size_t)p * N, (size_t)N * 8);
        if (!(flags & 2)) ntt_fwd_limb(m, N, tmp);
        const uint64_t *pp = pt_ntt + ((pt_broadcast ? 0 : b) * L + l) * (size_t)N;
        for (uint32_t j = 0; j < N; j++) tmp[j] = barrett_mul(tmp[j], pp[j], m);
        if (!(flags & 4)) ntt_inv_limb(m, N, tmp);
        uint64_t *o = out + (size_t)p * N;
        if (flags & 1) { for (uint32_t j = 0; j < N; j++) { uint64_t r = o[j] + tmp[j]; o[j] = r >= m->q ? r - m->q : r; } }
        else memcpy(o, tmp, (size_t)N * 8);
        free(tmp);
    }
}

/* ------------------------------------------------------------------ key switching
 * Evaluator::switch_key_inplace restated for the BFV case (target polynomial in coefficient form), SEAL
 * evaluator.cpp [un-vendored; restated from the published algorithm, parity unpinned].  The context holds the KEY
 * moduli q_0..q_{L-1}, P with the special prime P LAST; the ciphertext lives at the data level (first L moduli).
 *   target [B][L][N]      coefficient form, limb I canonical mod q_I          (the polynomial being switched)
 *   ksk    [L][2][K][N]   K = L+1; digit I's key (a PublicKey: 2 polys x K limbs) in NTT form
 *   ct     [B][2][L][N]   coefficient form; the switched polynomial is ADDED into both components
 * For every output modulus m_J (J < K) and component c:  S_J = sum_I NTT_{m_J}(target_I mod m_J) . ksk[I][c][J];
 * then modulus switching with rounding:  t = (INTT_P(S_P) + floor(P/2)) mod P,
 *   ct[c][J] += P^-1 * (INTT_J(S_J) - (t mod q_J) + (floor(P/2) mod q_J))   mod q_J.
 * SEAL accumulates lazily in 128 bits and subtracts lazily; the residues it ends with are the canonical ones
 * computed here. */
void pfo_key_switch(const pfo_ctx *c, const uint64_t *target, const uint64_t *ksk, uint64_t *ct, size_t B, int nthreads) {
    int nt = clamp_threads(nthreads); (void)nt;
    const uint32_t N = c->N, K = c->L, L = K - 1;
    const pfo_limb *mp = &c->limb[K - 1];
    const uint64_t P = mp->q, half = P >> 1;
    PAR_FOR
    for (long b = 0; b < (long)B; b++) {
        uint64_t *acc = (uint64_t *)malloc((size_t)2 * K * N * 8);     /* [2][K][N] products, NTT form then coefficient form */
        uint64_t *tmp = (uint64_t *)malloc((size_t)N * 8);
        for (uint32_t J = 0; J < K; J++) {
            const pfo_limb *m = &c->limb[J];
            uint64_t *a0 = acc + (size_t)J * N, *a1 = acc + ((size_t)K + J) * N;
            memset(a0, 0, (size_t)N * 8); memset(a1, 0, (size_t)N * 8);
            for (uint32_t I = 0; I < L; I++) {
                const uint64_t *d = target + ((size_t)b * L + I) * N;
                for (uint32_t n = 0; n < N; n++) tmp[n] = d[n] % m->q;
                ntt_fwd_limb(m, N, tmp);
                const uint64_t *k0 = ksk + (((size_t)I * 2 + 0) * K + J) * N, *k1 = ksk + (((size_t)I * 2 + 1) * K + J) * N;
                for (uint32_t n = 0; n < N; n++) {
                    uint64_t p0 = barrett_mul(tmp[n], k0[n], m), p1 = barrett_mul(tmp[n], k1[n], m);
                    uint64_t s0 = a0[n] + p0, s1 = a1[n] + p1;
                    a0[n] = s0 >= m->q ? s0 - m->q : s0;
                    a1[n] = s1 >= m->q ? s1 - m->q : s1;
                }
            }
            ntt_inv_limb(m, N, a0);
            ntt_inv_limb(m, N, a1);
        }
        for (uint32_t comp = 0; comp < 2; comp++) {
            uint64_t *tp = acc + ((size_t)comp * K + (K - 1)) * N;       /* special-prime component */
            for (uint32_t n = 0; n < N; n++) { uint64_t t = tp[n] + half; tp[n] = t >= P ? t - P : t; }
            for (uint32_t J = 0; J < L; J++) {
                const pfo_limb *m = &c->limb[J];
                const uint64_t q = m->q, halfj = half % q, pinv = invmod(P % q, q);
                const uint64_t *sj = acc + ((size_t)comp * K + J) * N;
                uint64_t *o = ct + (((size_t)b * 2 + comp) * L + J) * N;
                for (uint32_t n = 0; n < N; n++) {
                    uint64_t v = sj[n] + (q - tp[n] % q) + halfj;       /* < 3q */
                    v %= q;
                    uint64_t r = o[n] + mulmod_u128(v, pinv, q);
                    o[n] = r >= q ? r - q : r;
                }
            }
        }
        free(acc); free(tmp);
    }
}

/* ------------------------------------------------------------------ plaintext-distance rows */
/* Server::preciseSearch, /root/reference/src/server/server_lib.cpp:151-164, literal semantics:
 * float dist; dist += std::pow(float_diff, 2)  ==  dist = (float)((double)dist + pow((double)diff, 2.0)). */
void pfo_precise_search(const float *base, const float *xq, const int64_t *ids, size_t nq, size_t c, size_t d, float *out) {
    for (size_t i = 0; i < nq; i++)
        for (size_t j = 0; j < c; j++) {
            float dist = 0.0;
            const float *row = base + (size_t)ids[i * c + j] * d;
            for (size_t k = 0; k < d; k++) dist += pow((row[k] - xq[i * d + k]), 2);
            out[i * c + j] = dist;
        }
}

/* Server::preciseVectorPIR, server_lib.cpp:169-196: plain row copy */
void pfo_gather_rows(const float *base, const int64_t *ids, size_t n_ids, size_t d, float *out) {
    for (size_t i = 0; i < n_ids; i++) memcpy(out + i * d, base + (size_t)ids[i] * d, d * sizeof(float));
}

/* IndexIVFPQ::search_encrypted restated (fork source absent; semantics from server_lib.cpp:126-135 and its consumer
 * client_lib.cpp:122-156, arithmetic contract in prefhetch_amd/csrc/pf_ivfpq.hip): every vector of every GIVEN list, in
 * order; residual tables in fp32, mul then add, no contraction; distance = in-order fp32 sum over sub-quantizers.
 * Index content is passed in flat form: codes/ids list-contiguous with list_off[nlist+1]. Returns the total count. */
size_t pfo_ivfpq_search_lists(const float *xq, size_t nq, const int64_t *probe, size_t nprobe, const float *centroids, size_t nlist,
                              size_t d, size_t M, const float *codebooks, const uint8_t *codes, const int64_t *ids,
                              const uint64_t *list_off, float *D, int64_t *I, uint64_t *list_sizes) {
    const size_t dsub = d / M;
    size_t out = 0;
    float *lut = (float *)malloc(M * 256 * sizeof(float));
    for (size_t q = 0; q < nq; q++) {
        uint64_t per_q = 0;
        for (size_t j = 0; j < nprobe; j++) {
            const int64_t l = probe[q * nprobe + j];
            if (l < 0 || (size_t)l >= nlist) continue;
            for (size_t m = 0; m < M; m++)
                for (size_t c = 0; c < 256; c++) {
                    float acc = 0.f;
                    for (size_t t = 0; t < dsub; t++) {
                        volatile float r = xq[q * d + m * dsub + t] - centroids[(size_t)l * d + m * dsub + t];
                        volatile float diff = r - codebooks[(m * 256 + c) * dsub + t];
                        volatile float sq = diff * diff;
                        acc = acc + sq;
                    }
                    lut[m * 256 + c] = acc;
                }
            for (uint64_t v = list_off[l]; v < list_off[l + 1]; v++) {
                float dis = 0.f;
                for (size_t m = 0; m < M; m++) dis = dis + lut[m * 256 + codes[v * M + m]];
                D[out] = dis; I[out] = ids[v]; out++; per_q++;
            }
        }
        list_sizes[q] = per_q;
    }
    free(lut);
    return out;
}

typedef struct { float dis; int64_t id; } pfo_hit;
static int hit_cmp(const void *a, const void *b) {
    const pfo_hit *x = (const pfo_hit *)a, *y = (const pfo_hit *)b;
    if (x->dis < y->dis) return -1;
    if (x->dis > y->dis) return 1;
    return (x->id > y->id) - (x->id < y->id);
}

/* IndexFlatL2::search semantics: squared L2, ascending, ties -> smaller id (this build's rule),
 * -1 / +inf padding when k > nb.  Distances are accumulated in double from the float inputs and
 * rounded once to float (the exact value for SIFT-like integer data, and the tightest float for
 * anything else); mode 1 instead reproduces the literal pow/float accumulation of
 * sort_nearest_centroids (/root/reference/src/client/client_lib.cpp:55-67). */
void pfo_flat_l2_search(const float *xb, size_t nb, size_t d, const float *xq, size_t nq, size_t k,
                        float *D, int64_t *I, int mode, int nthreads) {
    int nt = clamp_threads(nthreads); (void)nt;
    PAR_FOR
    for (long i = 0; i < (long)nq; i++) {
        pfo_hit *h = (pfo_hit *)malloc(sizeof(pfo_hit) * (nb ? nb : 1));
        const float *x = xq + (size_t)i * d;
        for (size_t j = 0; j < nb; j++) {
            const float *y = xb + j * d;
            if (mode == 1) {
                float dist = 0.0;
                for (size_t t = 0; t < d; t++) dist += pow(x[t] - y[t], 2);
                h[j].dis = dist;
            } else {
                double acc = 0.0;
                for (size_t t = 0; t < d; t++) { double df = (double)x[t] - (double)y[t]; acc += df * df; }
                h[j].dis = (float)acc;
            }
            h[j].id = (int64_t)j;
        }
        qsort(h, nb, sizeof(pfo_hit), hit_cmp);
        for (size_t j = 0; j < k; j++) {
            if (j < nb) { D[(size_t)i * k + j] = h[j].dis; I[(size_t)i * k + j] = h[j].id; }
            else { D[(size_t)i * k + j] = INFINITY; I[(size_t)i * k + j] = -1; }
        }
        free(h);
    }
}

/* faiss-style fp32 baseline kernel for timing only (cpu_baseline of the pre-filter):
 * blocked direct squared-L2 in float, per-query bounded insertion (k smallest). */
void pfo_flat_l2_search_f32(const float *xb, size_t nb, size_t d, const float *xq, size_t nq, size_t k,
                            float *D, int64_t *I, int nthreads) {
    int nt = clamp_threads(nthreads); (void)nt;
    PAR_FOR
    for (long i = 0; i < (long)nq; i++) {
        float *bd = D + (size_t)i * k; int64_t *bi = I + (size_t)i * k;
        size_t cnt = 0;
        const float *x = xq + (size_t)i * d;
        for (size_t j = 0; j < nb; j++) {
            const float *y = xb + j * d;
            float acc = 0.f;
            for (size_t t = 0; t < d; t++) { float df = x[t] - y[t]; acc += df * df; }
            if (cnt == k && !(acc < bd[k - 1])) continue;
            size_t pos = cnt < k ? cnt++ : k - 1;
            while (pos > 0 && (bd[pos - 1] > acc)) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; pos--; }
            bd[pos] = acc; bi[pos] = (int64_t)j;
        }
        for (size_t j = cnt; j < k; j++) { bd[j] = INFINITY; bi[j] = -1; }
    }
}

/* faiss's exhaustive_L2sqr_blas + ReservoirTopN (what IndexFlatL2::search runs for nq >= 20), selection half, for the
 * cpu_baseline timing: the caller computes ip = xq . xb_block^T with BLAS sgemm (numpy's OpenBLAS) and this routine folds
 * the block into every query's reservoir: dist = |x|^2 + |y|^2 - 2 x.y, kept when below the query's threshold; a full
 * reservoir (capacity cap >= 2k) is cut back to its k smallest (quickselect) and the threshold tightened.
 * res_d / res_i [nq][cap], res_cnt / thr [nq] persist across blocks (thr starts at +inf, res_cnt at 0).
 * finish = 1 additionally sorts each reservoir (distance, then id) and writes the k best to D / I. */
static void pfo_hit_select(pfo_hit *h, size_t n, size_t k) {          /* k smallest of h[0..n) to the front */
    size_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const pfo_hit pv = h[lo + (hi - lo) / 2];
        size_t i = lo, j = hi - 1;
        while (i <= j) {
            while (hit_cmp(&h[i], &pv) < 0) i++;
            while (hit_cmp(&h[j], &pv) > 0) { if (j == 0) break; j--; }
            if (i <= j) { pfo_hit t = h[i]; h[i] = h[j]; h[j] = t; i++; if (j == 0) break; j--; }
        }
        if (k <= j + 1 && j + 1 > lo && j + 1 < hi) hi = j + 1;
        else if (k >= i && i > lo) lo = i;
        else if (k > j + 1 && k < i) return;                            /* the split point falls between the partitions */
        else { qsort(h + lo, hi - lo, sizeof(pfo_hit), hit_cmp); return; }   /* degenerate pivot: finish directly */
    }
}

void pfo_l2_reservoir_block(const float *ip, size_t nq, size_t nblk, size_t ld, const float *qn, const float *bn, int64_t id0,
                            size_t k, size_t cap, float *res_d, int64_t *res_i, uint32_t *res_cnt, float *thr, int finish,
                            float *D, int64_t *I, int nthreads) {
    int nt = clamp_threads(nthreads); (void)nt;
    PAR_FOR
    for (long q = 0; q < (long)nq; q++) {
        float *rd = res_d + (size_t)q * cap; int64_t *ri = res_i + (size_t)q * cap;
        size_t cnt = res_cnt[q];
        float t = thr[q];
        const float *row = ip + (size_t)q * ld;
        const float nq2 = qn[q];
        for (size_t j = 0; j < nblk; j++) {
            float dist = nq2 + bn[j] - 2.f * row[j];
            if (dist < 0.f) dist = 0.f;
            if (!(dist < t)) continue;
            rd[cnt] = dist; ri[cnt] = id0 + (int64_t)j; cnt++;
            if (cnt == cap) {
                pfo_hit *h = (pfo_hit *)malloc(sizeof(pfo_hit) * cap);
                for (size_t e = 0; e < cap; e++) { h[e].dis = rd[e]; h[e].id = ri[e]; }
                pfo_hit_select(h, cap, k);
                float mx = 0.f;
                for (size_t e = 0; e < k; e++) { rd[e] = h[e].dis; ri[e] = h[e].id; if (h[e].dis > mx) mx = h[e].dis; }
                free(h);
                cnt = k; t = mx;          /* keep ties with the k-th out: ids ascend, later ones lose the tie anyway */
            }
        }
        res_cnt[q] = (uint32_t)cnt; thr[q] = t;
        if (finish) {
            pfo_hit *h = (pfo_hit *)malloc(sizeof(pfo_hit) * (cnt ? cnt : 1));
            for (size_t e = 0; e < cnt; e++) { h[e].dis = rd[e]; h[e].id = ri[e]; }
            qsort(h, cnt, sizeof(pfo_hit), hit_cmp);
            for (size_t e = 0; e < k; e++) {
                if (e < cnt) { D[(size_t)q * k + e] = h[e].dis; I[(size_t)q * k + e] = h[e].id; }
                else { D[(size_t)q * k + e] = INFINITY; I[(size_t)q * k + e] = -1; }
            }
            free(h);
        }
    }
}

int pfo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
