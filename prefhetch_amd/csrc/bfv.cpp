// bfv.cpp -- client-side BFV over the C ABI of libprefhetch_hip.so (include/client/bfv.h).  Host C++ only.
// Restates the published RNS-BFV scheme (key generation, public-key encryption, decryption, invariant noise budget);
// the reference leaves these steps as TODOs (/root/reference/include/client/client_lib.h:14,28-30) and pins SEAL for
// them (CMakeLists.txt:33-38), whose sources are not available offline.
#include "../../include/client/bfv.h"

#include <cmath>
#include <cstring>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>

#include "../../include/prefhetch_hip.h"

namespace bfv {

namespace {

using u128 = unsigned __int128;

void check(pf_status st, const char *what) {
    if (st != PF_OK) throw std::runtime_error(std::string(what) + ": " + pf_status_str(st) + " (" + pf_last_error() + ")");
}

uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((u128)a * b % q); }
uint64_t powmod(uint64_t a, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = mulmod(r, a, q);
        a = mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}
uint64_t invmod_prime(uint64_t a, uint64_t q) { return powmod(a, q - 2, q); }     // q prime

// ---- little-endian multiword unsigned integers (a few words: Q has at most 15 x 60 bits) ----------------------
using Words = std::vector<uint64_t>;

void trim(Words &a) { while (a.size() > 1 && a.back() == 0) a.pop_back(); }
int cmp(const Words &a, const Words &b) {
    const size_t n = a.size() > b.size() ? a.size() : b.size();
    for (size_t i = n; i-- > 0;) {
        const uint64_t x = i < a.size() ? a[i] : 0, y = i < b.size() ? b[i] : 0;
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}
Words mul_small(const Words &a, uint64_t m) {
    Words r(a.size() + 1);
    uint64_t carry = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        const u128 p = (u128)a[i] * m + carry;
        r[i] = (uint64_t)p;
        carry = (uint64_t)(p >> 64);
    }
    r[a.size()] = carry;
    trim(r);
    return r;
}
void add_small(Words &a, uint64_t v) {
    for (size_t i = 0; i < a.size() && v; ++i) {
        const uint64_t s = a[i] + v;
        v = s < v ? 1 : 0;
        a[i] = s;
    }
    if (v) a.push_back(v);
}
Words sub(const Words &a, const Words &b) {          // a >= b
    Words r(a.size());
    uint64_t borrow = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        const uint64_t y = i < b.size() ? b[i] : 0;
        const uint64_t d = a[i] - y - borrow;
        borrow = (a[i] < y || (a[i] == y && borrow)) ? 1 : 0;
        r[i] = d;
    }
    trim(r);
    return r;
}
int bit_length(const Words &a) {
    for (size_t i = a.size(); i-- > 0;)
        if (a[i]) return (int)(i * 64 + 64 - __builtin_clzll(a[i]));
    return 0;
}
long double to_ld(const Words &a) {
    long double v = 0;
    for (size_t i = a.size(); i-- > 0;) v = v * 18446744073709551616.0L + (long double)a[i];
    return v;
}
// floor(r / Q) and r mod Q for a quotient known to be below 2^63: floating-point estimate, exact correction
uint64_t divmod_small_quotient(const Words &r, const Words &Q, Words &rem) {
    long double est = floorl(to_ld(r) / to_ld(Q));
    if (est < 0) est = 0;
    if (est > 9.2e18L) est = 9.2e18L;
    uint64_t qq = (uint64_t)est;
    Words prod = mul_small(Q, qq);
    while (cmp(prod, r) > 0) { --qq; prod = sub(prod, Q); }                 // estimate too large
    rem = sub(r, prod);
    while (cmp(rem, Q) >= 0) { ++qq; rem = sub(rem, Q); }                   // estimate too small
    return qq;
}

// ---- samplers ------------------------------------------------------------------------------------------------
struct Buffered {
    ByteSource &src;
    std::vector<uint8_t> buf;
    size_t at = 0;
    explicit Buffered(ByteSource &s) : src(s), buf(1 << 16) { at = buf.size(); }
    void need(size_t n) {
        if (at + n <= buf.size()) return;
        const size_t left = buf.size() - at;
        std::memmove(buf.data(), buf.data() + at, left);
        src(buf.data() + left, buf.size() - left);
        at = 0;
    }
    uint8_t byte() { need(1); return buf[at++]; }
    uint64_t u64() { need(8); uint64_t v; std::memcpy(&v, buf.data() + at, 8); at += 8; return v; }
    uint64_t bits48() { need(6); uint64_t v = 0; std::memcpy(&v, buf.data() + at, 6); at += 6; return v; }
};

void sample_ternary(Buffered &rng, std::vector<int8_t> &out) {
    for (auto &c : out) {
        uint8_t b;
        do b = rng.byte(); while (b == 255);                                  // 255 = 3 * 85: unbiased
        c = (int8_t)(b % 3) - 1;
    }
}
// centred binomial, 21 + 21 coins: variance 10.5, |e| <= 21
void sample_error(Buffered &rng, std::vector<int8_t> &out) {
    for (auto &c : out) {
        const uint64_t v = rng.bits48();
        c = (int8_t)(__builtin_popcountll(v & 0x1FFFFFull) - __builtin_popcountll((v >> 21) & 0x1FFFFFull));
    }
}
void sample_uniform(Buffered &rng, uint64_t q, uint64_t *out, size_t n) {
    const uint64_t limit = ~0ull - (~0ull % q + 1) % q;                        // largest multiple of q, minus one
    for (size_t i = 0; i < n; ++i) {
        uint64_t v;
        do v = rng.u64(); while (v > limit);
        out[i] = v % q;
    }
}
// small signed coefficients -> [L][N] residues
void to_residues(const std::vector<int8_t> &c, const std::vector<uint64_t> &moduli, uint64_t *out) {
    const size_t N = c.size();
    for (size_t l = 0; l < moduli.size(); ++l)
        for (size_t i = 0; i < N; ++i) out[l * N + i] = c[i] >= 0 ? (uint64_t)c[i] : moduli[l] - (uint64_t)(-c[i]);
}

}  // namespace

// ---- randomness ----------------------------------------------------------------------------------------------
ByteSource system_random() {
    auto dev = std::make_shared<std::random_device>();
    return [dev](uint8_t *dst, size_t n) {
        size_t i = 0;
        while (i < n) {
            const unsigned v = (*dev)();
            const size_t k = n - i < sizeof v ? n - i : sizeof v;
            std::memcpy(dst + i, &v, k);
            i += k;
        }
    };
}
ByteSource seeded_random(uint64_t seed) {
    auto state = std::make_shared<uint64_t>(seed);
    return [state](uint8_t *dst, size_t n) {
        size_t i = 0;
        while (i < n) {
            uint64_t z = (*state += 0x9E3779B97F4A7C15ull);                    // splitmix64
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            const size_t k = n - i < 8 ? n - i : 8;
            std::memcpy(dst + i, &z, k);
            i += k;
        }
    };
}

// ---- parameters ----------------------------------------------------------------------------------------------
Params Params::seal_default(uint32_t N, uint64_t t, int device) {
    Params p;
    p.N = N; p.t = t; p.device = device;
    switch (N) {                                   // SEAL CoeffModulus::BFVDefault(N); the last prime is the special one
        case 1024: p.moduli = {0x7E00001}; break;
        case 2048: p.moduli = {0x3FFFFFFF000001}; break;
        case 4096: p.moduli = {0xFFFFEE001, 0xFFFFC4001}; p.special_prime = 0x1FFFFE0001; break;
        case 8192: p.moduli = {0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001}; p.special_prime = 0xFFFFFEBC001; break;
        case 32768:
            p.moduli = {0x7FFFFFFFE90001, 0x7FFFFFFFBF0001, 0x7FFFFFFFBD0001, 0x7FFFFFFFBA0001, 0x7FFFFFFFAA0001, 0x7FFFFFFFA50001,
                        0x7FFFFFFF9F0001, 0x7FFFFFFF7E0001, 0x7FFFFFFF770001, 0x7FFFFFFF380001, 0x7FFFFFFF330001, 0x7FFFFFFF2D0001,
                        0x7FFFFFFF170001, 0x7FFFFFFF150001, 0x7FFFFFFEF00001};
            p.special_prime = 0xFFFFFFFFF70001;
            break;
        default: throw std::invalid_argument("bfv::Params::seal_default: N must be 1024, 2048, 4096, 8192 or 32768");
    }
    return p;
}

// ---- device memory -------------------------------------------------------------------------------------------
DeviceWords::DeviceWords(int device, size_t words) : m_Device(device), m_Words(words) {
    void *p = nullptr;
    check(pf_malloc(device, &p, words * 8), "pf_malloc");
    m_Ptr = static_cast<uint64_t *>(p);
}
DeviceWords::~DeviceWords() { if (m_Ptr) pf_free(m_Device, m_Ptr); }
DeviceWords::DeviceWords(DeviceWords &&o) noexcept : m_Device(o.m_Device), m_Ptr(o.m_Ptr), m_Words(o.m_Words) { o.m_Ptr = nullptr; o.m_Words = 0; }
DeviceWords &DeviceWords::operator=(DeviceWords &&o) noexcept {
    if (this != &o) {
        if (m_Ptr) pf_free(m_Device, m_Ptr);
        m_Device = o.m_Device; m_Ptr = o.m_Ptr; m_Words = o.m_Words;
        o.m_Ptr = nullptr; o.m_Words = 0;
    }
    return *this;
}
void DeviceWords::upload(const uint64_t *src, size_t words, size_t offset) {
    if (offset + words > m_Words) throw std::out_of_range("DeviceWords::upload past the end");
    check(pf_memcpy_h2d(m_Device, m_Ptr + offset, src, words * 8, nullptr), "pf_memcpy_h2d");
    check(pf_stream_synchronize(m_Device, nullptr), "sync");
}
void DeviceWords::download(uint64_t *dst, size_t words, size_t offset) const {
    if (offset + words > m_Words) throw std::out_of_range("DeviceWords::download past the end");
    check(pf_memcpy_d2h(m_Device, dst, m_Ptr + offset, words * 8, nullptr), "pf_memcpy_d2h");
    check(pf_stream_synchronize(m_Device, nullptr), "sync");
}

// ---- context -------------------------------------------------------------------------------------------------
struct Context::Big {
    Words Q, half_Q;                               // product of the moduli, floor(Q / 2)
    std::vector<std::vector<uint64_t>> inv;        // inv[i][j] = q_j^-1 mod q_i, j < i (Garner)
    uint64_t r_t = 0;                              // Q mod t
    std::vector<uint64_t> delta;                   // floor(Q / t) mod q_l
};

Context::Context(const Params &params) : m_Params(params), m_Big(new Big) {
    const auto &q = m_Params.moduli;
    if (q.empty()) throw std::invalid_argument("bfv::Context: no moduli");
    if (m_Params.t < 2 || m_Params.t >= (1ull << 60)) throw std::invalid_argument("bfv::Context: plaintext modulus out of range");
    for (uint64_t m : q)
        if (m_Params.t % m == 0 || std::gcd(m_Params.t, m) != 1) throw std::invalid_argument("bfv::Context: t must be coprime to the moduli");
    check(pf_ctx_create(&m_Ring, m_Params.device, m_Params.N, (uint32_t)q.size(), q.data()), "pf_ctx_create");
    if (m_Params.special_prime) {
        std::vector<uint64_t> key_moduli = q;
        key_moduli.push_back(m_Params.special_prime);
        check(pf_ctx_create(&m_KeyRing, m_Params.device, m_Params.N, (uint32_t)key_moduli.size(), key_moduli.data()), "pf_ctx_create(key moduli)");
    }
    Big &b = *m_Big;
    b.Q = {1};
    b.r_t = 1 % m_Params.t;
    for (uint64_t m : q) {
        b.Q = mul_small(b.Q, m);
        b.r_t = mulmod(b.r_t, m % m_Params.t, m_Params.t);
    }
    b.half_Q = b.Q;                                // >> 1
    for (size_t i = 0; i < b.half_Q.size(); ++i)
        b.half_Q[i] = (b.half_Q[i] >> 1) | (i + 1 < b.half_Q.size() ? b.half_Q[i + 1] << 63 : 0);
    trim(b.half_Q);
    b.inv.resize(q.size());
    for (size_t i = 0; i < q.size(); ++i)
        for (size_t j = 0; j < i; ++j) b.inv[i].push_back(invmod_prime(q[j] % q[i], q[i]));
    for (uint64_t m : q) {                          // floor(Q/t) = (Q - r_t)/t = -r_t * t^-1 (mod q_l), because q_l | Q
        const uint64_t tinv = invmod_prime(m_Params.t % m, m);
        b.delta.push_back(mulmod(m - b.r_t % m, tinv, m) % m);
    }
}

Context::~Context() {
    if (m_Ring) pf_ctx_destroy(m_Ring);
    if (m_KeyRing) pf_ctx_destroy(m_KeyRing);
}

int Context::total_modulus_bits() const { return bit_length(m_Big->Q); }

namespace {
// x in [0, Q) from its residues: Garner's mixed-radix digits, then Horner
Words compose(const std::vector<uint64_t> &q, const std::vector<std::vector<uint64_t>> &inv, const uint64_t *res) {
    const size_t L = q.size();
    uint64_t digit[64];
    for (size_t i = 0; i < L; ++i) {
        uint64_t v = res[i] % q[i];
        for (size_t j = 0; j < i; ++j) {
            const uint64_t dj = digit[j] % q[i];
            v = mulmod(v >= dj ? v - dj : v + q[i] - dj, inv[i][j], q[i]);
        }
        digit[i] = v;
    }
    Words x = {digit[L - 1]};
    for (size_t i = L - 1; i-- > 0;) {
        x = mul_small(x, q[i]);
        add_small(x, digit[i]);
    }
    return x;
}
}  // namespace

uint64_t Context::scale_and_round(const uint64_t *residues) const {
    const Big &b = *m_Big;
    const Words r = mul_small(compose(m_Params.moduli, b.inv, residues), m_Params.t);
    Words rem;
    uint64_t qq = divmod_small_quotient(r, b.Q, rem);
    if (cmp(rem, b.half_Q) > 0) ++qq;                                          // rem > floor(Q/2) <=> 2 rem > Q (Q odd)
    return qq % m_Params.t;
}

int Context::noise_bits(const uint64_t *residues) const {
    const Big &b = *m_Big;
    const Words r = mul_small(compose(m_Params.moduli, b.inv, residues), m_Params.t);
    Words rem;
    (void)divmod_small_quotient(r, b.Q, rem);
    if (cmp(rem, b.half_Q) > 0) rem = sub(b.Q, rem);
    return bit_length(rem);
}

void Context::scaled_message(uint64_t m, uint64_t *out) const {
    const Big &b = *m_Big;
    const uint64_t t = m_Params.t;
    m %= t;
    const uint64_t fix = (uint64_t)(((u128)m * b.r_t + t / 2) / t);            // round(m (Q mod t) / t)
    for (size_t l = 0; l < m_Params.moduli.size(); ++l) {
        const uint64_t q = m_Params.moduli[l];
        out[l] = (mulmod(m % q, b.delta[l], q) + fix % q) % q;
    }
}

// ---- keys ----------------------------------------------------------------------------------------------------
KeyGenerator::KeyGenerator(const Context &ctx, ByteSource rng) : m_Ctx(ctx), m_Rng(std::move(rng)) {
    const size_t N = ctx.N(), L = ctx.L();
    Buffered r(m_Rng);
    m_Secret.coeff.resize(N);
    sample_ternary(r, m_Secret.coeff);
    std::vector<uint64_t> res(L * N);
    to_residues(m_Secret.coeff, ctx.params().moduli, res.data());
    m_Secret.ntt = DeviceWords(ctx.params().device, L * N);
    m_Secret.ntt.upload(res.data(), L * N);
    check(pf_ntt_forward(ctx.ring(), m_Secret.ntt.ptr(), L, nullptr), "pf_ntt_forward");
    if (ctx.key_ring()) {
        std::vector<uint64_t> key_moduli = ctx.params().moduli;
        key_moduli.push_back(ctx.params().special_prime);
        std::vector<uint64_t> kres((L + 1) * N);
        to_residues(m_Secret.coeff, key_moduli, kres.data());
        m_Secret.ntt_key = DeviceWords(ctx.params().device, (L + 1) * N);
        m_Secret.ntt_key.upload(kres.data(), (L + 1) * N);
        check(pf_ntt_forward(ctx.key_ring(), m_Secret.ntt_key.ptr(), L + 1, nullptr), "pf_ntt_forward");
    }
    check(pf_stream_synchronize(ctx.params().device, nullptr), "sync");
}

// Digit I of the key: (-(a_I s + e_I) + [J == I] (P mod q_I) s', a_I) over the key moduli J, NTT form -- the published
// RNS key-switching key with one special prime (SEAL KeyGenerator::generate_one_kswitch_key).
SwitchKey KeyGenerator::create_switch_key(const std::vector<int8_t> &new_secret) {
    if (!m_Ctx.key_ring()) throw std::runtime_error("bfv: these parameters have no special prime: key switching is not available");
    const size_t N = m_Ctx.N(), D = m_Ctx.L(), Kn = D + 1;
    if (new_secret.size() != N) throw std::invalid_argument("bfv: new secret must have N coefficients");
    const int dev = m_Ctx.params().device;
    std::vector<uint64_t> key_moduli = m_Ctx.params().moduli;
    key_moduli.push_back(m_Ctx.params().special_prime);
    const uint64_t P = m_Ctx.params().special_prime;
    Buffered r(m_Rng);
    SwitchKey out;
    out.ksk = DeviceWords(dev, D * 2 * Kn * N);
    // NTT(s') over the key moduli, on the host for the per-digit correction
    std::vector<uint64_t> sp(Kn * N);
    to_residues(new_secret, key_moduli, sp.data());
    DeviceWords d_sp(dev, Kn * N), d_e(dev, Kn * N);
    d_sp.upload(sp.data(), Kn * N);
    check(pf_ntt_forward(m_Ctx.key_ring(), d_sp.ptr(), Kn, nullptr), "pf_ntt_forward");
    d_sp.download(sp.data(), Kn * N);
    std::vector<uint64_t> a(Kn * N), e_res(Kn * N), row(N);
    std::vector<int8_t> e(N);
    for (size_t I = 0; I < D; ++I) {
        uint64_t *k0 = out.ksk.ptr() + (I * 2 + 0) * Kn * N, *k1 = out.ksk.ptr() + (I * 2 + 1) * Kn * N;
        for (size_t J = 0; J < Kn; ++J) sample_uniform(r, key_moduli[J], a.data() + J * N, N);      // a_I, NTT form
        sample_error(r, e);
        to_residues(e, key_moduli, e_res.data());
        out.ksk.upload(a.data(), Kn * N, (I * 2 + 1) * Kn * N);
        d_e.upload(e_res.data(), Kn * N);
        check(pf_ntt_forward(m_Ctx.key_ring(), d_e.ptr(), Kn, nullptr), "pf_ntt_forward");
        check(pf_dyadic_mul(m_Ctx.key_ring(), k1, m_Secret.ntt_key.ptr(), k0, Kn, nullptr), "pf_dyadic_mul");      // a s
        check(pf_poly_add(m_Ctx.key_ring(), k0, d_e.ptr(), k0, Kn, nullptr), "pf_poly_add");                       // + e
        check(pf_poly_negate(m_Ctx.key_ring(), k0, k0, Kn, nullptr), "pf_poly_negate");
        check(pf_stream_synchronize(dev, nullptr), "sync");
        // + (P mod q_I) * s' on modulus I only
        const uint64_t q = key_moduli[I], f = P % q;
        out.ksk.download(row.data(), N, (I * 2 + 0) * Kn * N + I * N);
        for (size_t n = 0; n < N; ++n) row[n] = (row[n] + mulmod(f, sp[I * N + n], q)) % q;
        out.ksk.upload(row.data(), N, (I * 2 + 0) * Kn * N + I * N);
    }
    return out;
}

SwitchKey KeyGenerator::create_galois_key(uint32_t g) {
    const size_t N = m_Ctx.N();
    if (!(g & 1) || g >= 2 * N) throw std::invalid_argument("bfv: galois_elt must be odd and below 2N");
    std::vector<int8_t> sg(N);
    for (size_t i = 0; i < N; ++i) {                                               // s(X^g)
        const size_t j = (i * (size_t)g) % (2 * N);
        sg[j % N] = j >= N ? (int8_t)-m_Secret.coeff[i] : m_Secret.coeff[i];
    }
    SwitchKey k = create_switch_key(sg);
    k.galois_elt = g;
    return k;
}

PublicKey KeyGenerator::create_public_key() {
    const size_t N = m_Ctx.N(), L = m_Ctx.L();
    const int dev = m_Ctx.params().device;
    const auto &q = m_Ctx.params().moduli;
    Buffered r(m_Rng);
    std::vector<uint64_t> a(L * N), e_res(L * N);
    for (size_t l = 0; l < L; ++l) sample_uniform(r, q[l], a.data() + l * N, N);          // a, taken as already in NTT form
    std::vector<int8_t> e(N);
    sample_error(r, e);
    to_residues(e, q, e_res.data());
    PublicKey pk;
    pk.coeffs = DeviceWords(dev, 2 * L * N);
    DeviceWords tmp(dev, L * N);
    uint64_t *pk0 = pk.coeffs.ptr(), *pk1 = pk.coeffs.ptr() + L * N;
    pk.coeffs.upload(a.data(), L * N, L * N);                                                // pk1 <- a (NTT form)
    tmp.upload(e_res.data(), L * N);
    check(pf_dyadic_mul(m_Ctx.ring(), pk1, m_Secret.ntt.ptr(), pk0, L, nullptr), "pf_dyadic_mul");   // a s
    check(pf_ntt_inverse(m_Ctx.ring(), pk.coeffs.ptr(), 2 * L, nullptr), "pf_ntt_inverse");   // both to coefficient form
    check(pf_poly_add(m_Ctx.ring(), pk0, tmp.ptr(), pk0, L, nullptr), "pf_poly_add");         // a s + e
    check(pf_poly_negate(m_Ctx.ring(), pk0, pk0, L, nullptr), "pf_poly_negate");
    check(pf_stream_synchronize(dev, nullptr), "sync");
    return pk;
}

// ---- encryption ----------------------------------------------------------------------------------------------
Encryptor::Encryptor(const Context &ctx, const PublicKey &pk, ByteSource rng) : m_Ctx(ctx), m_Pk(pk), m_Rng(std::move(rng)) {}

void Encryptor::encrypt(const uint64_t *plain, size_t count, Ciphertexts &out) {
    const size_t N = m_Ctx.N(), L = m_Ctx.L();
    const int dev = m_Ctx.params().device;
    const auto &q = m_Ctx.params().moduli;
    out.count = count;
    if (count == 0) return;
    if (out.data.words() < count * 2 * L * N) out.data = DeviceWords(dev, count * 2 * L * N);
    Buffered r(m_Rng);
    std::vector<uint64_t> base(count * 2 * L * N), u_res(count * L * N), m_res(L);
    std::vector<int8_t> u(N), e(N);
    for (size_t c = 0; c < count; ++c) {
        sample_ternary(r, u);
        to_residues(u, q, u_res.data() + c * L * N);
        uint64_t *b0 = base.data() + c * 2 * L * N, *b1 = b0 + L * N;
        sample_error(r, e);
        to_residues(e, q, b0);
        sample_error(r, e);
        to_residues(e, q, b1);
        for (size_t i = 0; i < N; ++i) {                                        // + round(Q m / t)
            const uint64_t m = plain[c * N + i];
            if (m >= m_Ctx.t()) throw std::invalid_argument("bfv::Encryptor: plaintext coefficient not below t");
            if (m == 0) continue;
            m_Ctx.scaled_message(m, m_res.data());
            for (size_t l = 0; l < L; ++l) {
                const uint64_t s = b0[l * N + i] + m_res[l];
                b0[l * N + i] = s >= q[l] ? s - q[l] : s;
            }
        }
    }
    DeviceWords U(dev, count * L * N);
    U.upload(u_res.data(), count * L * N);
    out.data.upload(base.data(), count * 2 * L * N);
    check(pf_ntt_forward(m_Ctx.ring(), U.ptr(), count * L, nullptr), "pf_ntt_forward");
    // the public key is an encryption of zero: every ciphertext is pk x u_c added to (e0 + scaled message, e1)
    check(pf_ct_pt_mul_fanout(m_Ctx.ring(), m_Pk.coeffs.ptr(), U.ptr(), out.data.ptr(), count, (uint32_t)count, PF_CTPT_ACCUMULATE, nullptr),
          "pf_ct_pt_mul_fanout");
    check(pf_stream_synchronize(dev, nullptr), "sync");
}

// ---- decryption ----------------------------------------------------------------------------------------------
Decryptor::Decryptor(const Context &ctx, const SecretKey &sk) : m_Ctx(ctx), m_Sk(sk) {}

void Decryptor::phase(const Ciphertexts &ct, std::vector<uint64_t> &v) {
    const size_t N = m_Ctx.N(), L = m_Ctx.L(), per = 2 * L * N;
    const int dev = m_Ctx.params().device;
    const auto &q = m_Ctx.params().moduli;
    if (ct.data.words() < ct.count * per) throw std::invalid_argument("bfv::Decryptor: ciphertext buffer too small");
    DeviceWords prod(dev, ct.count * per);
    // (c0 s, c1 s): the secret key as a broadcast NTT-form plaintext; only the second component is used
    check(pf_ct_pt_mul(m_Ctx.ring(), ct.data.ptr(), m_Sk.ntt.ptr(), 1, prod.ptr(), ct.count, 0, nullptr), "pf_ct_pt_mul");
    std::vector<uint64_t> c(ct.count * per), p(ct.count * per);
    ct.data.download(c.data(), ct.count * per);
    prod.download(p.data(), ct.count * per);
    v.resize(ct.count * L * N);
    for (size_t i = 0; i < ct.count; ++i)
        for (size_t l = 0; l < L; ++l)
            for (size_t k = 0; k < N; ++k) {
                const uint64_t s = c[i * per + l * N + k] + p[i * per + (L + l) * N + k];
                v[(i * L + l) * N + k] = s >= q[l] ? s - q[l] : s;
            }
}

void Decryptor::decrypt(const Ciphertexts &ct, std::vector<uint64_t> &plain) {
    const size_t N = m_Ctx.N(), L = m_Ctx.L();
    std::vector<uint64_t> v;
    phase(ct, v);
    plain.resize(ct.count * N);
    uint64_t res[64];
    for (size_t i = 0; i < ct.count; ++i)
        for (size_t k = 0; k < N; ++k) {
            for (size_t l = 0; l < L; ++l) res[l] = v[(i * L + l) * N + k];
            plain[i * N + k] = m_Ctx.scale_and_round(res);
        }
}

int Decryptor::invariant_noise_budget(const Ciphertexts &ct, size_t index) {
    if (index >= ct.count) throw std::out_of_range("bfv::Decryptor: ciphertext index");
    const size_t N = m_Ctx.N(), L = m_Ctx.L(), per = 2 * L * N;
    Ciphertexts one;                                                             // a view would do; keep the helper simple
    one.count = 1;
    one.data = DeviceWords(m_Ctx.params().device, per);
    std::vector<uint64_t> tmp(per);
    ct.data.download(tmp.data(), per, index * per);
    one.data.upload(tmp.data(), per);
    std::vector<uint64_t> v;
    phase(one, v);
    int worst = 0;
    uint64_t res[64];
    for (size_t k = 0; k < N; ++k) {
        for (size_t l = 0; l < L; ++l) res[l] = v[l * N + k];
        const int b = m_Ctx.noise_bits(res);
        if (b > worst) worst = b;
    }
    const int budget = m_Ctx.total_modulus_bits() - worst - 1;
    return budget > 0 ? budget : 0;
}

// ---- batching ------------------------------------------------------------------------------------------------
BatchEncoder::BatchEncoder(const Context &ctx) : m_Ctx(ctx) {
    const uint32_t N = ctx.N();
    const uint64_t t = ctx.t();
    check(pf_ctx_create(&m_PlainRing, ctx.params().device, N, 1, &t), "pf_ctx_create(plaintext modulus: must be a prime = 1 mod 2N)");
    int logn = 0;
    while ((1u << logn) < N) ++logn;
    auto reverse_bits = [logn](uint32_t v) {
        uint32_t r = 0;
        for (int b = 0; b < logn; ++b) r |= ((v >> b) & 1u) << (logn - 1 - b);
        return r;
    };
    m_IndexMap.resize(N);
    const uint32_t row = N / 2, m = 2 * N;
    uint64_t pos = 1;
    for (uint32_t i = 0; i < row; ++i) {
        m_IndexMap[i] = reverse_bits((uint32_t)((pos - 1) >> 1));
        m_IndexMap[row | i] = reverse_bits((uint32_t)((m - pos - 1) >> 1));
        pos = (pos * 3) & (m - 1);
    }
}

BatchEncoder::~BatchEncoder() { if (m_PlainRing) pf_ctx_destroy(m_PlainRing); }

void BatchEncoder::encode(const uint64_t *values, uint64_t *plain_out) const {
    const size_t N = m_IndexMap.size();
    std::vector<uint64_t> tmp(N);
    for (size_t i = 0; i < N; ++i) {
        if (values[i] >= m_Ctx.t()) throw std::invalid_argument("bfv::BatchEncoder: value not below t");
        tmp[m_IndexMap[i]] = values[i];
    }
    DeviceWords d(m_Ctx.params().device, N);
    d.upload(tmp.data(), N);
    check(pf_ntt_inverse(m_PlainRing, d.ptr(), 1, nullptr), "pf_ntt_inverse");
    d.download(plain_out, N);
}

void BatchEncoder::decode(const uint64_t *plain, uint64_t *values_out) const {
    const size_t N = m_IndexMap.size();
    DeviceWords d(m_Ctx.params().device, N);
    d.upload(plain, N);
    check(pf_ntt_forward(m_PlainRing, d.ptr(), 1, nullptr), "pf_ntt_forward");
    std::vector<uint64_t> tmp(N);
    d.download(tmp.data(), N);
    for (size_t i = 0; i < N; ++i) values_out[i] = tmp[m_IndexMap[i]];
}

// ---- Galois automorphism on ciphertexts ----------------------------------------------------------------------
void apply_galois_device(const Context &ctx, const uint64_t *in, size_t count, const SwitchKey &key, uint64_t *out, uint64_t *scratch) {
    if (!ctx.key_ring()) throw std::runtime_error("bfv: these parameters have no special prime: key switching is not available");
    if (!key.galois_elt) throw std::invalid_argument("bfv::apply_galois: not a Galois key");
    if (count == 0) return;
    // tau(c0) into the first component, zeros into the second, tau(c1) = the polynomial to switch: one launch for the batch
    check(pf_apply_galois_ct(ctx.ring(), in, out, scratch, count, key.galois_elt, nullptr), "pf_apply_galois_ct");
    check(pf_key_switch(ctx.key_ring(), scratch, key.ksk.ptr(), out, count, nullptr), "pf_key_switch");
}

void apply_galois(const Context &ctx, const Ciphertexts &in, const SwitchKey &key, Ciphertexts &out) {
    const size_t N = ctx.N(), L = ctx.L(), per = 2 * L * N;
    const int dev = ctx.params().device;
    out.count = in.count;
    if (in.count == 0) return;
    if (out.data.words() < in.count * per) out.data = DeviceWords(dev, in.count * per);
    DeviceWords target(dev, in.count * L * N);
    apply_galois_device(ctx, in.data.ptr(), in.count, key, out.data.ptr(), target.ptr());
    check(pf_stream_synchronize(dev, nullptr), "sync");
}

void apply_galois_plain(const uint64_t *plain, uint32_t N, uint64_t t, uint32_t g, uint64_t *out) {
    for (uint32_t i = 0; i < N; ++i) {
        const uint64_t j = ((uint64_t)i * g) % (2ull * N);
        const uint64_t v = plain[i] % t;
        out[j % N] = (j >= N && v) ? t - v : v;
    }
}

// ---- encoding of the encrypted precise search ----------------------------------------------------------------
void encode_query(const float *query, uint32_t d, uint32_t N, uint64_t t, uint64_t *plain_out) {
    if (d > N) throw std::invalid_argument("bfv::encode_query: d > N");
    std::memset(plain_out, 0, (size_t)N * 8);
    for (uint32_t i = 0; i < d; ++i) {
        const long long v = llrintf(query[i]);
        const uint64_t a = (uint64_t)(v < 0 ? -v : v) % t;
        plain_out[i] = v >= 0 ? a : (a ? t - a : 0);
    }
}

void decode_inner_products(const uint64_t *plain, uint32_t d, uint32_t rows, uint64_t t, int64_t *out) {
    for (uint32_t j = 0; j < rows; ++j) {
        const uint64_t v = plain[(size_t)d * j];
        out[j] = v > t / 2 ? (int64_t)v - (int64_t)t : (int64_t)v;
    }
}

}  // namespace bfv
