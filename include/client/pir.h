// pir.h -- private retrieval of base rows: Server::preciseVectorPIR as an actual PIR.
//
// The reference's handler copies the K requested rows by their ids, which travel in the clear
// (/root/reference/src/server/server_lib.cpp:169-196: "the PIR is a placeholder"; ids sent by
// src/client/client_lib.cpp:210-241).  Here the id never leaves the client: one BFV ciphertext per wanted row goes up,
// one comes back, and the server's work is built from the two hot-path operations of this repository -- the fused
// ciphertext x plaintext product (pf_ct_pt_mul) and key switching (pf_key_switch through bfv::apply_galois).
// SURVEY.md 8(f-4).
//
// Scheme (SealPIR: Angel, Chen, Laine, Setty, "PIR with compressed queries and amortized query processing", S&P 2018,
// one dimension, restated from the paper -- no SealPIR source is available here):
//   database   rows of d floats are packed 2 coefficients per value (the 32 bits of a float as two 16-bit halves, t > 2^16),
//              N / (2 d) rows per plaintext polynomial; the polynomials are lifted to the ciphertext moduli and kept in NTT
//              form: the "plaintext" operand of pf_ct_pt_mul.  n_sel <= 2^levels <= N polynomials per column (Layout).
//   query      Enc(2^-levels * X^p) for the polynomial p that holds the wanted row (one ciphertext).
//   expand     levels rounds; round j maps every ciphertext c to (c + s_j(c), (c - s_j(c)) * X^(-2^j)) with the Galois
//              automorphism s_j: X -> X^(N / 2^j + 1) (pf_apply_galois_ct + pf_key_switch) and the monomial product a signed
//              coefficient shift (pf_poly_mul_monomial): 2^levels ciphertexts, the k-th
//              encrypting 1 if k = p and 0 otherwise.
//   answer     sum_k expanded_k x database_k = Enc(database_p): pf_ct_pt_mul with NTT-form output, a tree of pf_poly_add,
//              one inverse transform.
//   decode     the client decrypts and reads the 2 d coefficients of its row.
// Which polynomial was asked for is hidden by the semantic security of BFV; which row inside it never leaves the client.
// The server learns nothing about the id (the client learns the other rows of the polynomial as well -- there is no data
// privacy in this protocol, as in the reference's plain copy).
#pragma once

#include <cstdint>
#include <vector>

#include "bfv.h"

namespace pir {

struct Layout {
    uint32_t N = 0, d = 0;
    size_t n_rows = 0;
    uint32_t rows_per_poly = 0;       // N / (2 d)
    size_t n_polys = 0;               // ceil(n_rows / rows_per_poly)
    // A query selects one of n_sel <= N polynomials (one expansion, 2^levels >= n_sel selection ciphertexts).  A base with more
    // polynomials than that is laid out in n_cols columns of n_sel: polynomial p sits in column p / n_sel at position p % n_sel,
    // the query names the position only, and the reply is one ciphertext PER COLUMN -- the client keeps the one it wants, so the
    // column never leaves it either (1M rows x 128 floats at N = 8192: 31 250 polynomials = 4 columns, 2 MiB of reply).
    size_t n_sel = 0, n_cols = 1;
    uint32_t levels = 0;              // smallest with 2^levels >= n_sel
    // max_sel: cap on n_sel (0 = the ring degree; smaller values force several columns on a small base: tests)
    static Layout make(uint32_t N, uint32_t d, size_t n_rows, size_t max_sel = 0);
    size_t poly_of(size_t row) const { return row / rows_per_poly; }
    uint32_t slot_of(size_t row) const { return (uint32_t)(row % rows_per_poly); }
    size_t sel_of(size_t row) const { return poly_of(row) % n_sel; }
    size_t col_of(size_t row) const { return poly_of(row) / n_sel; }
};

// ---- server side ------------------------------------------------------------------------------------------------
class Database {
  public:
    // rows: [n_rows][d] floats on the host.  ctx: BFV parameters with a special prime (key switching), t > 65536.
    Database(const bfv::Context &ctx, const float *rows, size_t n_rows, uint32_t d, size_t max_sel = 0);
    const Layout &layout() const { return m_Layout; }
    const uint64_t *ntt() const { return m_Ntt.ptr(); }       // [n_polys][L][N], NTT form
    // Buffers of expand / answer, kept between calls (a retrieval at 262 144 rows works in 13 GB: allocating them per call cost
    // more than the arithmetic).  Not thread-safe: one answer() at a time per Database.
    struct Workspace { bfv::Ciphertexts sel, one; bfv::DeviceWords rot, scratch, prod; };
    Workspace &workspace() const { return m_Ws; }
  private:
    Layout m_Layout;
    bfv::DeviceWords m_Ntt;
    mutable Workspace m_Ws;
};

// The Galois elements the expansion needs, round 0 first: N / 2^j + 1.
std::vector<uint32_t> galois_elements(uint32_t N, uint32_t levels);

// query: `count` ciphertexts (one retrieval each); keys[j] the Galois key of galois_elements()[j].
// reply: `count` x n_cols ciphertexts, the columns of a retrieval together.  All device-side; synchronises before returning.
void answer(const bfv::Context &ctx, const Database &db, const bfv::Ciphertexts &query, const std::vector<bfv::SwitchKey> &keys,
            bfv::Ciphertexts &reply);
// the expansion alone (test hook): one query ciphertext -> 2^levels selection ciphertexts
void expand(const bfv::Context &ctx, const bfv::Ciphertexts &query_one, const std::vector<bfv::SwitchKey> &keys, uint32_t levels,
            bfv::Ciphertexts &out, Database::Workspace *ws = nullptr);

// ---- client side ------------------------------------------------------------------------------------------------
// plaintext of the query for `row`: 2^-levels mod t at coefficient sel_of(row)
void encode_query(const Layout &lay, uint64_t t, size_t row, uint64_t *plain_out /* [N] */);
// the wanted row out of the decrypted reply of one retrieval (its n_cols plaintexts, in order)
void decode_row(const Layout &lay, const uint64_t *plain /* [n_cols][N] */, size_t row, float *out /* [d] */);

}  // namespace pir
