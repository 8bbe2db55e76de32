// pir.cpp -- see include/client/pir.h.  Host C++ over the C ABI (prefhetch_hip.h) and the BFV helpers (bfv.h): every
// ring operation runs on the GPU -- pf_ct_pt_mul (the database products), pf_apply_galois_ct + pf_key_switch
// (bfv::apply_galois), pf_poly_add / pf_poly_sub / pf_poly_mul_monomial (the expansion's butterflies), pf_ntt_forward / pf_ntt_inverse.
#include "../../include/client/pir.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../include/prefhetch_hip.h"

namespace pir {
namespace {

void check(pf_status st, const char *what) {
    if (st != PF_OK) throw std::runtime_error(std::string("pir: ") + what + ": " + pf_status_str(st) + " (" + pf_last_error() + ")");
}

uint64_t powmod(uint64_t a, uint64_t e, uint64_t m) {
    unsigned __int128 r = 1 % m, b = a % m;
    for (; e; e >>= 1, b = b * b % m)
        if (e & 1) r = r * b % m;
    return (uint64_t)r;
}

// inverse of 2^levels modulo an odd t (t need not be prime): ((t + 1) / 2)^levels
uint64_t inv_pow2(uint32_t levels, uint64_t t) {
    if (!(t & 1)) throw std::invalid_argument("pir: the plaintext modulus must be odd (2^levels has to be invertible)");
    return powmod((t + 1) / 2, levels, t);
}

}  // namespace

Layout Layout::make(uint32_t N, uint32_t d, size_t n_rows, size_t max_sel) {
    Layout l;
    if (d == 0 || 2 * (size_t)d > N) throw std::invalid_argument("pir: a row (2 coefficients per value) must fit one polynomial");
    l.N = N; l.d = d; l.n_rows = n_rows;
    l.rows_per_poly = N / (2 * d);
    l.n_polys = (n_rows + l.rows_per_poly - 1) / l.rows_per_poly;
    if (l.n_polys == 0) l.n_polys = 1;
    const size_t cap = max_sel && max_sel < N ? max_sel : N;
    l.n_sel = l.n_polys < cap ? l.n_polys : cap;
    l.n_cols = (l.n_polys + l.n_sel - 1) / l.n_sel;
    while ((size_t{1} << l.levels) < l.n_sel) ++l.levels;
    return l;
}

Database::Database(const bfv::Context &ctx, const float *rows, size_t n_rows, uint32_t d, size_t max_sel)
    : m_Layout(Layout::make(ctx.N(), d, n_rows, max_sel)) {
    if (ctx.t() <= 65536) throw std::invalid_argument("pir: plaintext modulus must exceed 2^16 (two 16-bit halves per value)");
    const size_t N = ctx.N(), L = ctx.L(), P = m_Layout.n_polys;
    // coefficients are below t < every q_l: the lift to the ciphertext moduli repeats the value in each limb
    std::vector<uint64_t> host(P * L * N, 0);
    for (size_t r = 0; r < n_rows; ++r) {
        uint64_t *poly = host.data() + m_Layout.poly_of(r) * L * N;
        const size_t c0 = (size_t)m_Layout.slot_of(r) * 2 * d;
        for (uint32_t i = 0; i < d; ++i) {
            uint32_t bits;
            std::memcpy(&bits, rows + r * d + i, 4);
            for (size_t l = 0; l < L; ++l) {
                poly[l * N + c0 + 2 * i] = bits & 0xFFFFu;
                poly[l * N + c0 + 2 * i + 1] = bits >> 16;
            }
        }
    }
    m_Ntt = bfv::DeviceWords(ctx.params().device, P * L * N);
    m_Ntt.upload(host.data(), host.size());
    check(pf_ntt_forward(ctx.ring(), m_Ntt.ptr(), P * L, nullptr), "pf_ntt_forward");
    check(pf_stream_synchronize(ctx.params().device, nullptr), "sync");
}

std::vector<uint32_t> galois_elements(uint32_t N, uint32_t levels) {
    std::vector<uint32_t> g(levels);
    for (uint32_t j = 0; j < levels; ++j) g[j] = (N >> j) + 1;
    return g;
}

void expand(const bfv::Context &ctx, const bfv::Ciphertexts &query_one, const std::vector<bfv::SwitchKey> &keys, uint32_t levels,
            bfv::Ciphertexts &out, Database::Workspace *ws) {
    if (query_one.count != 1) throw std::invalid_argument("pir::expand: one query ciphertext at a time");
    if (keys.size() < levels) throw std::invalid_argument("pir::expand: a Galois key per round is needed");
    const size_t N = ctx.N(), L = ctx.L(), per = 2 * L * N, n_out = size_t{1} << levels;
    const int dev = ctx.params().device;
    const std::vector<uint32_t> elts = galois_elements(ctx.N(), levels);
    out.count = n_out;
    if (out.data.words() < n_out * per) out.data = bfv::DeviceWords(dev, n_out * per);
    check(pf_memcpy_d2d(dev, out.data.ptr(), query_one.data.ptr(), per * 8, nullptr), "d2d");
    // s_j(c) of a round and the polynomials its key switch works on, sized for the last (largest) round
    Database::Workspace local;
    Database::Workspace &w = ws ? *ws : local;
    if (levels && w.rot.words() < (n_out / 2) * per) w.rot = bfv::DeviceWords(dev, (n_out / 2) * per);
    if (levels && w.scratch.words() < (n_out / 2) * L * N) w.scratch = bfv::DeviceWords(dev, (n_out / 2) * L * N);
    for (uint32_t j = 0; j < levels; ++j) {
        if (keys[j].galois_elt != elts[j]) throw std::invalid_argument("pir::expand: keys[j] must be the Galois key of N / 2^j + 1");
        const size_t B = size_t{1} << j;
        // the first B ciphertexts of `out` are this round's inputs c; they become c + s_j(c) in place, the next B are (c - s_j(c)) * X^(-2^j)
        uint64_t *lo = out.data.ptr(), *hi = out.data.ptr() + B * per;
        bfv::apply_galois_device(ctx, lo, B, keys[j], w.rot.ptr(), w.scratch.ptr());     // automorphism + key switch, B at once
        // X^(-2^j) = X^(2N - 2^j): a signed shift of the coefficients (SealPIR's multiply_power_of_X), fused with the sum and the difference
        check(pf_poly_addsub_monomial(ctx.ring(), lo, w.rot.ptr(), lo, hi, B * 2 * L, 2 * ctx.N() - (1u << j), nullptr), "pf_poly_addsub_monomial");
    }
    check(pf_stream_synchronize(dev, nullptr), "sync");
}

void answer(const bfv::Context &ctx, const Database &db, const bfv::Ciphertexts &query, const std::vector<bfv::SwitchKey> &keys,
            bfv::Ciphertexts &reply) {
    const Layout &lay = db.layout();
    const size_t N = ctx.N(), L = ctx.L(), per = 2 * L * N, P = lay.n_polys;
    const int dev = ctx.params().device;
    const size_t S = lay.n_sel, C = lay.n_cols;
    reply.count = query.count * C;
    if (query.count == 0) return;
    if (reply.data.words() < reply.count * per) reply.data = bfv::DeviceWords(dev, reply.count * per);
    Database::Workspace &w = db.workspace();
    bfv::Ciphertexts &one = w.one, &sel = w.sel;
    one.count = 1;
    if (one.data.words() < per) one.data = bfv::DeviceWords(dev, per);
    const size_t prod_cts = S > C * 16 ? S : C * 16;                                   // the products of a column, or 16 partial sums per column
    if (w.prod.words() < prod_cts * per) w.prod = bfv::DeviceWords(dev, prod_cts * per);
    bfv::DeviceWords &prod = w.prod;
    const bool trace = std::getenv("PF_PIR_TRACE") != nullptr;
    auto stamp = [&](const char *what, std::chrono::steady_clock::time_point &t0) {
        if (!trace) return;
        pf_stream_synchronize(dev, nullptr);
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "pir::answer %-10s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    for (size_t q = 0; q < query.count; ++q) {
        auto t0 = std::chrono::steady_clock::now();
        check(pf_memcpy_d2d(dev, one.data.ptr(), query.data.ptr() + q * per, per * 8, nullptr), "d2d");
        expand(ctx, one, keys, lay.levels, sel, &w);
        stamp("expand", t0);
        // per column: sum_k selection_k x database_k in NTT form, one inverse transform at the end
        const size_t splits = 16;
        if (S % splits == 0 && S >= 256) {
            // the selection ciphertexts go to NTT form once; ONE pass over them and the database forms the sums, split 16 ways along
            // k for parallelism (pf_ct_pt_dot: no product is ever written), and the 16 partial sums of a column are added up
            const size_t chunk = S / splits, G = (P + chunk - 1) / chunk;
            check(pf_ntt_forward(ctx.ring(), sel.data.ptr(), S * 2 * L, nullptr), "pf_ntt_forward");
            check(pf_ct_pt_dot(ctx.ring(), sel.data.ptr(), S, db.ntt(), P, chunk, prod.ptr(), nullptr), "pf_ct_pt_dot");
            for (size_t c = 0; c < C; ++c) {
                uint64_t *part = prod.ptr() + c * splits * per;
                for (size_t n = G - c * splits < splits ? G - c * splits : splits; n > 1;) {
                    const size_t half = n / 2, keep = n - half;
                    check(pf_poly_add(ctx.ring(), part, part + keep * per, part, half * 2 * L, nullptr), "pf_poly_add");
                    n = keep;
                }
                check(pf_ntt_inverse_to(ctx.ring(), part, reply.data.ptr() + (q * C + c) * per, 2 * L, nullptr), "pf_ntt_inverse_to");
            }
        } else {
            // small bases: products in NTT form, summed there by halving (one launch per halving)
            for (size_t c = 0; c < C; ++c) {
                const size_t cnt = P - c * S < S ? P - c * S : S;
                check(pf_ct_pt_mul(ctx.ring(), sel.data.ptr(), db.ntt() + c * S * L * N, cnt, prod.ptr(), cnt, PF_CTPT_OUT_NTT, nullptr), "pf_ct_pt_mul");
                for (size_t n = cnt; n > 1;) {
                    const size_t half = n / 2, keep = n - half;                          // fold the last `half` onto the first `half`
                    check(pf_poly_add(ctx.ring(), prod.ptr(), prod.ptr() + keep * per, prod.ptr(), half * 2 * L, nullptr), "pf_poly_add");
                    n = keep;
                }
                check(pf_ntt_inverse_to(ctx.ring(), prod.ptr(), reply.data.ptr() + (q * C + c) * per, 2 * L, nullptr), "pf_ntt_inverse_to");
            }
        }
        stamp("columns", t0);
    }
    check(pf_stream_synchronize(dev, nullptr), "sync");
}

void encode_query(const Layout &lay, uint64_t t, size_t row, uint64_t *plain_out) {
    if (row >= lay.n_rows) throw std::out_of_range("pir::encode_query: no such row");
    std::memset(plain_out, 0, (size_t)lay.N * 8);
    plain_out[lay.sel_of(row)] = inv_pow2(lay.levels, t);
}

void decode_row(const Layout &lay, const uint64_t *plain, size_t row, float *out) {
    plain += lay.col_of(row) * lay.N;                                                    // the column's plaintext
    const size_t c0 = (size_t)lay.slot_of(row) * 2 * lay.d;
    for (uint32_t i = 0; i < lay.d; ++i) {
        const uint32_t bits = (uint32_t)(plain[c0 + 2 * i] & 0xFFFFu) | ((uint32_t)(plain[c0 + 2 * i + 1] & 0xFFFFu) << 16);
        std::memcpy(out + i, &bits, 4);
    }
}

}  // namespace pir
