// bfv.h -- the client-side BFV steps the reference's TODOs point at (/root/reference/include/client/client_lib.h:14,
// 28-30: "Replace std::vector<float> with the corresponding Encrypted Vector type", compute_encrypted_*_query): key
// generation, encryption, decryption and the invariant noise budget, so that the server's ct x pt path
// (include/prefhetch_hip.h) is exercised on REAL ciphertexts.  SURVEY.md 8(f-3).
//
// Scheme: textbook RNS-BFV as Microsoft SEAL 4.1 runs it (the reference pins SEAL @ 7a931d55, CMakeLists.txt:33-38;
// its sources are not available here, so this is a restatement of the published scheme, not of SEAL's code):
//   secret key   s  <- uniform ternary {-1,0,1}^N
//   public key   (pk0, pk1) = (-(a s + e), a),  a uniform mod Q, e <- centred binomial (21 - 21 coin flips, sigma 3.24)
//   Enc(m)       (pk0 u + e0 + round(Q m / t), pk1 u + e1),  u ternary, e0, e1 centred binomial
//   Dec(c0, c1)  round(t (c0 + c1 s mod Q) / Q) mod t
//   noise budget floor(log2 Q) - bits(|t (c0 + c1 s) mod Q|_centred) - 1   (SEAL's invariant noise budget)
// Ring arithmetic (every polynomial product) runs on the GPU through the C ABI; sampling, the scaling of the message
// and the CRT reconstruction of decryption are host code (exact integer arithmetic, no floating point in any result).
// Ciphertext layout = SEAL's Ciphertext::data(): [2][L][N] uint64, coefficient form, canonical residues.
//
// Randomness: every consumer takes a ByteSource.  The default reads the operating system's generator
// (std::random_device); tests pass a seeded deterministic source.
#pragma once

#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

struct pf_ctx;

namespace bfv {

using ByteSource = std::function<void(uint8_t *dst, size_t bytes)>;
ByteSource system_random();
ByteSource seeded_random(uint64_t seed);          // splitmix64 stream: reproducible, NOT for production keys

struct Params {
    uint32_t N = 0;                               // ring degree (1024 .. 32768)
    std::vector<uint64_t> moduli;                 // ciphertext primes q_0 .. q_{L-1}, each = 1 mod 2N
    uint64_t t = 0;                               // plaintext modulus, 2 <= t < 2^60, coprime to every q_l
    uint64_t special_prime = 0;                   // key-switching prime P (0: no key switching), = 1 mod 2N
    int device = 0;
    // SEAL CoeffModulus::BFVDefault(N): all primes but the last are the data level of a fresh ciphertext, the last is
    // the key-switching prime (N = 1024 and 2048 have a single prime: no key switching, as in SEAL)
    static Params seal_default(uint32_t N, uint64_t t, int device = 0);
};

// Device memory owned through the C ABI (pf_malloc / pf_free).
class DeviceWords {
  public:
    DeviceWords() = default;
    DeviceWords(int device, size_t words);
    ~DeviceWords();
    DeviceWords(DeviceWords &&o) noexcept;
    DeviceWords &operator=(DeviceWords &&o) noexcept;
    DeviceWords(const DeviceWords &) = delete;
    DeviceWords &operator=(const DeviceWords &) = delete;
    uint64_t *ptr() const { return m_Ptr; }
    size_t words() const { return m_Words; }
    void upload(const uint64_t *src, size_t words, size_t offset_words = 0);
    void download(uint64_t *dst, size_t words, size_t offset_words = 0) const;

  private:
    int m_Device = 0;
    uint64_t *m_Ptr = nullptr;
    size_t m_Words = 0;
};

class Context {
  public:
    explicit Context(const Params &params);
    ~Context();
    const Params &params() const { return m_Params; }
    uint32_t N() const { return m_Params.N; }
    size_t L() const { return m_Params.moduli.size(); }
    uint64_t t() const { return m_Params.t; }
    pf_ctx *ring() const { return m_Ring; }
    pf_ctx *key_ring() const { return m_KeyRing; }   // data primes + special prime (nullptr without one)
    int total_modulus_bits() const;               // bit length of Q

    // exact host arithmetic on one coefficient given its residues mod q_0 .. q_{L-1}
    uint64_t scale_and_round(const uint64_t *residues) const;                 // round(t x / Q) mod t
    int noise_bits(const uint64_t *residues) const;                            // bits of |t x mod Q| centred
    void scaled_message(uint64_t m, uint64_t *residues_out) const;             // round(Q m / t) mod q_l

  private:
    struct Big;
    Params m_Params;
    pf_ctx *m_Ring = nullptr, *m_KeyRing = nullptr;
    std::unique_ptr<Big> m_Big;
};

struct SecretKey {
    std::vector<int8_t> coeff;                    // ternary, host copy
    DeviceWords ntt;                              // [L][N], NTT form
    DeviceWords ntt_key;                          // [L+1][N], NTT form over the key moduli (with a special prime only)
};
// A key-switching key in the layout pf_key_switch takes: [D][2][D+1][N], NTT form over the key moduli; digit I is an
// encryption under s of P * s' restricted to modulus I (SEAL KSwitchKeys / GaloisKeys entry).
struct SwitchKey {
    DeviceWords ksk;
    uint32_t galois_elt = 0;                      // for Galois keys: the automorphism this key undoes
};
struct PublicKey {
    DeviceWords coeffs;                           // [2][L][N], COEFFICIENT form (the input pf_ct_pt_mul expects)
};
struct Ciphertexts {
    DeviceWords data;                             // [count][2][L][N], coefficient form
    size_t count = 0;
};

class KeyGenerator {
  public:
    KeyGenerator(const Context &ctx, ByteSource rng = system_random());
    const SecretKey &secret_key() const { return m_Secret; }
    PublicKey create_public_key();
    // key that switches a ciphertext component multiplying s'(X) back to s(X); s' given by its small coefficients
    SwitchKey create_switch_key(const std::vector<int8_t> &new_secret_coeff);
    // Galois key of X -> X^galois_elt (galois_elt odd, < 2N): s' = s(X^galois_elt)
    SwitchKey create_galois_key(uint32_t galois_elt);

  private:
    const Context &m_Ctx;
    ByteSource m_Rng;
    SecretKey m_Secret;
};

class Encryptor {
  public:
    Encryptor(const Context &ctx, const PublicKey &pk, ByteSource rng = system_random());
    // plain: [count][N] coefficients in [0, t).  One ciphertext per plaintext.
    void encrypt(const uint64_t *plain, size_t count, Ciphertexts &out);

  private:
    const Context &m_Ctx;
    const PublicKey &m_Pk;
    ByteSource m_Rng;
};

class Decryptor {
  public:
    Decryptor(const Context &ctx, const SecretKey &sk);
    // plain: [count][N] coefficients in [0, t)
    void decrypt(const Ciphertexts &ct, std::vector<uint64_t> &plain);
    // invariant noise budget in bits of ciphertext `index` (0 = decryption no longer guaranteed)
    int invariant_noise_budget(const Ciphertexts &ct, size_t index);

  private:
    void phase(const Ciphertexts &ct, std::vector<uint64_t> &v);              // c0 + c1 s, [count][L][N] residues
    const Context &m_Ctx;
    const SecretKey &m_Sk;
};

// SEAL's BatchEncoder, restated (batchencoder.cpp: slot i of the first row is the plaintext's value at psi_t^(3^i), slot
// i of the second row its value at psi_t^(-3^i), psi_t the minimal primitive 2N-th root of unity mod t): N values mod
// t <-> one plaintext polynomial, so that ciphertext addition / plaintext multiplication act slot-wise and the Galois
// automorphism X -> X^3 rotates both rows left by one, X -> X^(2N-1) swaps the rows.  t must be a prime = 1 mod 2N.
// The transforms run on the GPU through a one-modulus context over t.
class BatchEncoder {
  public:
    explicit BatchEncoder(const Context &ctx);
    ~BatchEncoder();
    BatchEncoder(const BatchEncoder &) = delete;
    BatchEncoder &operator=(const BatchEncoder &) = delete;
    size_t slot_count() const { return m_IndexMap.size(); }
    void encode(const uint64_t *values /* [N], each < t */, uint64_t *plain_out /* [N] */) const;
    void decode(const uint64_t *plain /* [N] */, uint64_t *values_out /* [N] */) const;

  private:
    const Context &m_Ctx;
    pf_ctx *m_PlainRing = nullptr;
    std::vector<uint32_t> m_IndexMap;
};

// Server-side use of a Galois key, here so that the key-switching path can be exercised on real ciphertexts:
// out = Enc(m(X^g)) from in = Enc(m(X)): apply X -> X^g to both components (pf_apply_galois), then switch the second
// one from s(X^g) back to s (pf_key_switch).  SEAL: Evaluator::apply_galois.
void apply_galois(const Context &ctx, const Ciphertexts &in, const SwitchKey &galois_key, Ciphertexts &out);
// The same on device pointers, for callers that keep their own buffers: in, out [count][2][L][N] (distinct), scratch
// [count][L][N] words.  Asynchronous (default stream), no allocation.
void apply_galois_device(const Context &ctx, const uint64_t *in, size_t count, const SwitchKey &galois_key, uint64_t *out, uint64_t *scratch);
// the plaintext side of the same map: m(X) -> m(X^g) mod (X^N + 1, t)
void apply_galois_plain(const uint64_t *plain, uint32_t N, uint64_t t, uint32_t galois_elt, uint64_t *out);

// ---- encoding of the encrypted precise search (pairs with pf_pack_rows on the server) --------------------------
// query q[0..d) (integers after rounding) -> plaintext polynomial sum_i q_i X^i, coefficients mod t
void encode_query(const float *query, uint32_t d, uint32_t N, uint64_t t, uint64_t *plain_out);
// decrypted product polynomial -> the inner products <q, row_j>, j < rows: coefficient d*j, centred mod t
void decode_inner_products(const uint64_t *plain, uint32_t d, uint32_t rows, uint64_t t, int64_t *out);

}  // namespace bfv
