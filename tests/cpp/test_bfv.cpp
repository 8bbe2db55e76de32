// test_bfv.cpp -- the client-side BFV library (include/client/bfv.h) and the encrypted precise search of the server
// (Server::preciseSearchEncrypted) on a real MI355X: encrypt -> server -> decrypt must reproduce the plaintext
// protocol's numbers exactly, with noise budget to spare.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>
#include <vector>

#include "../../include/client/bfv.h"
#include "../../include/client/pir.h"
#include "../../include/client/client_lib.h"
#include "../../include/server/wire.h"
#include "../../include/prefhetch_hip.h"
#include "../../include/server/server_lib.h"

static int fails = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++fails; } } while (0)

int main() {
    // ---- 1. scheme round trips at three parameter sets ------------------------------------------------------
    struct Case { uint32_t N; uint64_t t; };
    for (const Case c : {Case{1024, 257}, Case{4096, 65537}, Case{8192, (1ull << 25) + 0x8001 /* any t coprime to the moduli */}}) {
        bfv::Context ctx(bfv::Params::seal_default(c.N, c.t));
        bfv::KeyGenerator keygen(ctx, bfv::seeded_random(1000 + c.N));
        bfv::PublicKey pk = keygen.create_public_key();
        bfv::Encryptor enc(ctx, pk, bfv::seeded_random(2000 + c.N));
        bfv::Decryptor dec(ctx, keygen.secret_key());
        const size_t count = 3;
        std::vector<uint64_t> plain(count * c.N), back;
        std::mt19937_64 rng(c.N);
        for (auto &v : plain) v = rng() % c.t;
        for (size_t i = 0; i < c.N; ++i) plain[i] = 0;                            // an all-zero plaintext
        for (size_t i = 0; i < c.N; ++i) plain[c.N + i] = c.t - 1;                // the largest coefficients
        bfv::Ciphertexts ct;
        enc.encrypt(plain.data(), count, ct);
        dec.decrypt(ct, back);
        EXPECT(back == plain);
        const int budget = dec.invariant_noise_budget(ct, 2);
        std::printf("N=%u, %zu limb(s), log2 Q = %d, t = %llu: fresh noise budget %d bits\n", c.N, ctx.L(), ctx.total_modulus_bits(),
                    (unsigned long long)c.t, budget);
        EXPECT(budget > 0 && budget < ctx.total_modulus_bits());
        // two encryptions of the same message differ (fresh randomness), the wrong key does not decrypt
        bfv::Ciphertexts ct2;
        enc.encrypt(plain.data(), 1, ct2);
        std::vector<uint64_t> a(2 * ctx.L() * c.N), b(a.size());
        ct.data.download(a.data(), a.size());
        ct2.data.download(b.data(), b.size());
        EXPECT(a != b);
        bfv::KeyGenerator other(ctx, bfv::seeded_random(77));
        bfv::Decryptor wrong(ctx, other.secret_key());
        std::vector<uint64_t> junk;
        wrong.decrypt(ct, junk);
        EXPECT(junk != plain);
        EXPECT(wrong.invariant_noise_budget(ct, 1) == 0);
        // homomorphic addition through the C ABI: Dec(ct_0 + ct_2) = m_0 + m_2 mod t
        bfv::Ciphertexts sum;
        sum.count = 1;
        sum.data = bfv::DeviceWords(0, 2 * ctx.L() * c.N);
        const size_t per = 2 * ctx.L() * c.N;
        EXPECT(pf_poly_add(ctx.ring(), ct.data.ptr(), ct.data.ptr() + 2 * per, sum.data.ptr(), 2 * ctx.L(), nullptr) == PF_OK);
        std::vector<uint64_t> s;
        dec.decrypt(sum, s);
        bool ok = true;
        for (size_t i = 0; i < c.N; ++i) ok = ok && s[i] == (plain[i] + plain[2 * c.N + i]) % c.t;
        EXPECT(ok);
    }

    // ---- 1b. key switching on real ciphertexts: Dec(apply_galois(Enc(m))) = m(X^g), at SEAL's default parameter sets
    // with a special prime, N = 32768 / 15 + 1 primes (BASELINE config 5) included --------------------------------
    for (const Case c : {Case{4096, 65537}, Case{8192, 65537}, Case{32768, 65537}}) {
        bfv::Context ctx(bfv::Params::seal_default(c.N, c.t));
        bfv::KeyGenerator keygen(ctx, bfv::seeded_random(3000 + c.N));
        bfv::PublicKey pk = keygen.create_public_key();
        bfv::Encryptor enc(ctx, pk, bfv::seeded_random(4000 + c.N));
        bfv::Decryptor dec(ctx, keygen.secret_key());
        std::vector<uint64_t> plain(2 * c.N), back, expect(c.N);
        std::mt19937_64 rng(c.N + 1);
        for (auto &v : plain) v = rng() % c.t;
        bfv::Ciphertexts ct, rot, rot2;
        enc.encrypt(plain.data(), 2, ct);
        const int fresh = dec.invariant_noise_budget(ct, 0);
        int after = 0;
        for (const uint32_t g : {3u, 2 * c.N - 1}) {                                   // a row rotation step and the column swap of SEAL's batching
            bfv::SwitchKey gk = keygen.create_galois_key(g);
            bfv::apply_galois(ctx, ct, gk, rot);
            dec.decrypt(rot, back);
            bool ok = true;
            for (size_t i = 0; i < 2; ++i) {
                bfv::apply_galois_plain(plain.data() + i * c.N, c.N, c.t, g, expect.data());
                ok = ok && std::memcmp(expect.data(), back.data() + i * c.N, c.N * 8) == 0;
            }
            EXPECT(ok);
            after = dec.invariant_noise_budget(rot, 1);
            EXPECT(after > 0 && after <= fresh);
            if (g == 3) {                                                              // twice: X -> X^9
                bfv::apply_galois(ctx, rot, gk, rot2);
                dec.decrypt(rot2, back);
                bfv::apply_galois_plain(plain.data(), c.N, c.t, 9, expect.data());
                EXPECT(std::memcmp(expect.data(), back.data(), c.N * 8) == 0);
            }
        }
        // batching: slot-wise semantics of plaintext multiplication and of the two generators of the Galois group
        {
            bfv::BatchEncoder be(ctx);
            const size_t row = c.N / 2;
            std::vector<uint64_t> va(c.N), vb(c.N), pa(c.N), pb(c.N), out(c.N), dec_plain;
            for (auto &v : va) v = rng() % c.t;
            for (auto &v : vb) v = rng() % 7;
            be.encode(va.data(), pa.data());
            be.encode(vb.data(), pb.data());
            be.decode(pa.data(), out.data());
            EXPECT(out == va);                                                          // decode . encode = id
            bfv::Ciphertexts ca, cr;
            enc.encrypt(pa.data(), 1, ca);
            bfv::SwitchKey g3 = keygen.create_galois_key(3), gcol = keygen.create_galois_key(2 * c.N - 1);
            bfv::apply_galois(ctx, ca, g3, cr);                                          // rotate_rows(1)
            dec.decrypt(cr, dec_plain);
            be.decode(dec_plain.data(), out.data());
            bool ok = true;
            for (size_t i = 0; i < row; ++i) ok = ok && out[i] == va[(i + 1) % row] && out[row + i] == va[row + (i + 1) % row];
            EXPECT(ok);
            bfv::apply_galois(ctx, ca, gcol, cr);                                        // rotate_columns
            dec.decrypt(cr, dec_plain);
            be.decode(dec_plain.data(), out.data());
            ok = true;
            for (size_t i = 0; i < row; ++i) ok = ok && out[i] == va[row + i] && out[row + i] == va[i];
            EXPECT(ok);
            // ciphertext x plaintext acts slot-wise: Dec(Enc(a) * b) decodes to a_i * b_i mod t
            std::vector<uint64_t> pb_res(ctx.L() * c.N);
            for (size_t l = 0; l < ctx.L(); ++l)
                for (size_t i = 0; i < c.N; ++i) pb_res[l * c.N + i] = pb[i];           // b's coefficients are below t < q_l
            bfv::DeviceWords d_pb(0, ctx.L() * c.N);
            d_pb.upload(pb_res.data(), pb_res.size());
            EXPECT(pf_ntt_forward(ctx.ring(), d_pb.ptr(), ctx.L(), nullptr) == PF_OK);
            bfv::Ciphertexts prod;
            prod.count = 1;
            prod.data = bfv::DeviceWords(0, 2 * ctx.L() * c.N);
            EXPECT(pf_ct_pt_mul(ctx.ring(), ca.data.ptr(), d_pb.ptr(), 1, prod.data.ptr(), 1, 0, nullptr) == PF_OK);
            dec.decrypt(prod, dec_plain);
            be.decode(dec_plain.data(), out.data());
            ok = true;
            for (size_t i = 0; i < c.N; ++i) ok = ok && out[i] == (unsigned __int128)va[i] * vb[i] % c.t;
            EXPECT(ok);
        }
        std::printf("N=%u, %zu + 1 primes: Galois automorphism + key switch decrypts to m(X^g); noise budget %d -> %d bits\n", c.N, ctx.L(), fresh, after);
    }

    // ---- 2. the encrypted precise search against the plaintext one ------------------------------------------
    {
        std::mt19937 rng(5);
        std::vector<float> base(static_cast<size_t>(NBASE) * 128);
        for (auto &v : base) v = static_cast<float>(rng() % 256);
        auto srv = Server::getInstance();
        srv->init_from_memory(base.data(), NBASE);
        std::array<std::array<float, PRECISE_VECTOR_DIMENSIONS>, NQUERY> query;
        for (auto &q : query) for (auto &v : q) v = static_cast<float>(rng() % 256);
        std::array<std::array<faiss::idx_t, COARSE_PROBE>, NQUERY> ids;
        for (auto &row : ids) for (auto &v : row) v = rng() % NBASE;
        ids[1][7] = ids[1][8];                                                      // a repeated candidate
        std::array<std::array<float, COARSE_PROBE>, NQUERY> plain_dist;
        srv->preciseSearch(query, ids, plain_dist);

        // client: t must exceed twice the largest inner product, 128 * 255^2 = 8 323 200
        const uint64_t t = (1ull << 25) + 0x8001;
        {   // the server's parameter set and the client's table name the same primes (the wire format checks key residues against the server's)
            const bfv::Params pp = bfv::Params::seal_default(Server::ENC_RING_DEGREE, t);
            EXPECT(pp.special_prime == Server::ENC_SPECIAL_PRIME && pp.moduli.size() == Server::ENC_LIMBS);
            for (size_t l = 0; l < pp.moduli.size() && l < Server::ENC_LIMBS; ++l) EXPECT(pp.moduli[l] == Server::ENC_MODULI[l]);
        }
        bfv::Context ctx(bfv::Params::seal_default(Server::ENC_RING_DEGREE, t));
        bfv::KeyGenerator keygen(ctx, bfv::seeded_random(42));
        bfv::PublicKey pk = keygen.create_public_key();
        bfv::Encryptor enc(ctx, pk, bfv::seeded_random(43));
        bfv::Decryptor dec(ctx, keygen.secret_key());
        const size_t N = ctx.N(), L = ctx.L();
        std::vector<uint64_t> qplain(static_cast<size_t>(NQUERY) * N);
        for (size_t i = 0; i < static_cast<size_t>(NQUERY); ++i) bfv::encode_query(query[i].data(), 128, (uint32_t)N, t, qplain.data() + i * N);
        bfv::Ciphertexts qct;
        enc.encrypt(qplain.data(), NQUERY, qct);
        const int fresh = dec.invariant_noise_budget(qct, 0);

        // server: sees ciphertexts and candidate ids only
        bfv::Ciphertexts rct;
        rct.count = static_cast<size_t>(NQUERY) * Server::ENC_POLYS_PER_QUERY;
        rct.data = bfv::DeviceWords(0, rct.count * 2 * L * N);
        srv->preciseSearchEncrypted(qct.data.ptr(), ids, rct.data.ptr());
        Timer timer;                                                                 // second call: workspace and tables are warm
        timer.StartTimer();
        srv->preciseSearchEncrypted(qct.data.ptr(), ids, rct.data.ptr());
        timer.StopTimer();
        long long us = 0, ms = 0;
        timer.getDuration(us, ms);
        std::printf("Server::preciseSearchEncrypted: %lld us for %lld queries x %u ciphertext x plaintext products (pack + NTT + fused product, synchronous)\n",
                    us, (long long)NQUERY, Server::ENC_POLYS_PER_QUERY);

        // client: decrypt, decode, finish the distances with its own ||q||^2 and the row norms
        std::vector<uint64_t> rplain;
        dec.decrypt(rct, rplain);
        const int after = dec.invariant_noise_budget(rct, 0);
        std::printf("encrypted precise search: N=8192, 4 limbs, t=%llu: noise budget fresh %d bits, after ct x pt %d bits\n",
                    (unsigned long long)t, fresh, after);
        EXPECT(fresh > after && after > 60);
        bool all_equal = true;
        for (size_t i = 0; i < static_cast<size_t>(NQUERY); ++i) {
            double qn = 0;
            for (float v : query[i]) qn += double(v) * v;
            for (size_t b = 0; b < Server::ENC_POLYS_PER_QUERY; ++b) {
                int64_t ip[Server::ENC_ROWS_PER_POLY];
                bfv::decode_inner_products(rplain.data() + (i * Server::ENC_POLYS_PER_QUERY + b) * N, 128, Server::ENC_ROWS_PER_POLY, t, ip);
                for (size_t j = 0; j < Server::ENC_ROWS_PER_POLY; ++j) {
                    const size_t cand = b * Server::ENC_ROWS_PER_POLY + j;
                    if (cand >= static_cast<size_t>(COARSE_PROBE)) { all_equal = all_equal && ip[j] == 0; continue; }   // padding rows
                    const float *row = base.data() + ids[i][cand] * 128;
                    long long exact = 0;
                    double xn = 0;
                    for (int k = 0; k < 128; ++k) { exact += (long long)query[i][k] * (long long)row[k]; xn += double(row[k]) * row[k]; }
                    all_equal = all_equal && ip[j] == exact;
                    const double dist = qn - 2.0 * double(ip[j]) + xn;                // integers below 2^24: exact in fp32 too
                    all_equal = all_equal && static_cast<float>(dist) == plain_dist[i][cand];
                }
            }
        }
        EXPECT(all_equal);

        // ---- 3. the same round through the client library and the wire format: base64 ciphertexts in JSON ------
        wire::InProcessTransport link(*srv);
        set_transport(&link);
        std::array<std::vector<DistanceIndexData>, NQUERY> ranked;
        for (size_t i = 0; i < static_cast<size_t>(NQUERY); ++i)
            for (size_t j = 0; j < static_cast<size_t>(COARSE_PROBE); ++j) ranked[i].push_back(DistanceIndexData{float(j), ids[i][j]});
        std::array<std::array<float, COARSE_PROBE>, NQUERY> clear_scores, enc_scores;
        get_precise_scores(ranked, query, clear_scores);                              // the reference's round 3, query in the clear
        const size_t sent0 = link.bytes_sent, recv0 = link.bytes_received;
        get_precise_scores_encrypted(ranked, query, ctx, enc, dec, enc_scores);        // the same round, query encrypted
        EXPECT(std::memcmp(clear_scores.data(), plain_dist.data(), sizeof plain_dist) == 0);
        EXPECT(std::memcmp(enc_scores.data(), plain_dist.data(), sizeof plain_dist) == 0);
        std::printf("encrypted round over the wire format: %zu request bytes, %zu response bytes (plaintext round: %zu / %zu)\n",
                    link.bytes_sent - sent0, link.bytes_received - recv0, sent0, recv0);
        // base64 and malformed payloads
        const uint8_t raw[5] = {0, 255, 16, 32, 7};
        EXPECT(wire::base64_encode(raw, 5) == "AP8QIAc=" && wire::base64_decode("AP8QIAc=") == std::vector<uint8_t>(raw, raw + 5));
        bool threw = false;
        try { link.post("precisesearch-encrypted", "{\"nearestCoarseVectorIndexes\": [], \"queryCiphertexts\": \"AAAA\"}"); } catch (const std::out_of_range &) { threw = true; }
        EXPECT(threw);
        // ---- 3b. round 4 with private ids: the rows come back equal to the plain copy's, bit for bit -------------------
        {
            PreciseRanking nearest;
            for (size_t i = 0; i < static_cast<size_t>(NQUERY); ++i)
                for (size_t j = 0; j < static_cast<size_t>(COARSE_PROBE); ++j) nearest[i][j] = DistanceIndexData{clear_scores[i][j], ids[i][(j * 7 + i) % COARSE_PROBE]};
            auto plain_rows = std::make_unique<ResultVectors>();
            auto private_rows = std::make_unique<ResultVectors>();
            ResultIds plain_ids, private_ids;
            get_precise_vectors_pir(nearest, *plain_rows, plain_ids);                   // the reference's round 4: ids in the clear
            bfv::Context pctx(bfv::Params::seal_default(8192, Server::PIR_PLAIN_MODULUS));
            bfv::KeyGenerator pkeygen(pctx, bfv::seeded_random(31));
            bfv::PublicKey ppk = pkeygen.create_public_key();
            bfv::Encryptor penc(pctx, ppk, bfv::seeded_random(32));
            bfv::Decryptor pdec(pctx, pkeygen.secret_key());
            const size_t fetch = 2, sent1 = link.bytes_sent, recv1 = link.bytes_received;
            const auto t0 = std::chrono::steady_clock::now();
            get_precise_vectors_pir_private(nearest, pctx, pkeygen, penc, pdec, *private_rows, private_ids, fetch);
            const double pir_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            bool same = private_ids == plain_ids;
            for (size_t i = 0; i < static_cast<size_t>(NQUERY); ++i)
                for (size_t j = 0; j < fetch; ++j) same = same && std::memcmp((*private_rows)[i][j].data(), (*plain_rows)[i][j].data(), sizeof(float) * 128) == 0;
            EXPECT(same);
            std::printf("private retrieval over the wire format: %zu rows of %zu (levels %u), %zu request bytes (Galois keys once), %zu response bytes\n",
                        fetch * static_cast<size_t>(NQUERY), srv->pirRows(), srv->pirLevels(), link.bytes_sent - sent1, link.bytes_received - recv1);
            std::printf("  %.0f ms for the round (key generation, base64, %u expansion rounds of 2^j key switches and %zu products per row included)\n", pir_ms,
                        srv->pirLevels(), (srv->pirRows() + 31) / 32);
        }
        set_transport(nullptr);
    }
    // ---- 4. private row retrieval (include/client/pir.h): the id never leaves the client ----------------------------------
    {
        bfv::Context ctx(bfv::Params::seal_default(8192, 65537));
        bfv::KeyGenerator keygen(ctx, bfv::seeded_random(4242));
        bfv::PublicKey pk = keygen.create_public_key();
        bfv::Encryptor enc(ctx, pk, bfv::seeded_random(4243));
        bfv::Decryptor dec(ctx, keygen.secret_key());
        const uint32_t d = 128;
        const size_t n_rows = 1000;                                                    // 32 polynomials of 32 rows: five expansion rounds
        std::vector<float> base(n_rows * d);
        std::mt19937_64 rng(99);
        for (auto &v : base) v = (float)((double)(int64_t)(rng() % 2000001) / 1000.0 - 1000.0);   // fractions, negatives: every bit of a float matters
        base[5 * d + 3] = -0.0f; base[5 * d + 4] = 3.4e38f; base[5 * d + 5] = 1e-40f;              // signed zero, near the largest, a subnormal
        pir::Database db(ctx, base.data(), n_rows, d);
        const pir::Layout &lay = db.layout();
        EXPECT(lay.rows_per_poly == 32 && lay.n_polys == 32 && lay.levels == 5);
        std::vector<bfv::SwitchKey> keys;
        for (uint32_t g : pir::galois_elements(ctx.N(), lay.levels)) keys.push_back(keygen.create_galois_key(g));
        // the expansion alone: 2^levels ciphertexts, a one at the asked position and zeros elsewhere
        {
            std::vector<uint64_t> plain(ctx.N()), back;
            pir::encode_query(lay, ctx.t(), 21 * 32 + 7, plain.data());                  // polynomial 21
            bfv::Ciphertexts q, sel;
            enc.encrypt(plain.data(), 1, q);
            pir::expand(ctx, q, keys, lay.levels, sel);
            dec.decrypt(sel, back);
            bool ok = sel.count == 32;
            for (size_t k = 0; ok && k < 32; ++k)
                for (size_t i = 0; ok && i < ctx.N(); ++i) ok = back[k * ctx.N() + i] == (k == 21 && i == 0 ? 1u : 0u);
            EXPECT(ok);
            std::printf("PIR expansion: 1 -> 32 ciphertexts, noise budget %d -> %d bits\n", dec.invariant_noise_budget(q, 0), dec.invariant_noise_budget(sel, 31));
        }
        const size_t wanted[4] = {0, 5, 517, 999};
        std::vector<uint64_t> plain(4 * ctx.N()), back;
        for (size_t i = 0; i < 4; ++i) pir::encode_query(lay, ctx.t(), wanted[i], plain.data() + i * ctx.N());
        bfv::Ciphertexts query, reply;
        enc.encrypt(plain.data(), 4, query);
        pir::answer(ctx, db, query, keys, reply);
        dec.decrypt(reply, back);
        for (size_t i = 0; i < 4; ++i) {
            float row[128];
            pir::decode_row(lay, back.data() + i * ctx.N(), wanted[i], row);
            EXPECT(std::memcmp(row, base.data() + wanted[i] * d, sizeof row) == 0);     // bit for bit, -0.0f and the subnormal included
        }
        std::printf("PIR: 4 rows of %zu retrieved privately (1 ciphertext up, 1 down each); reply noise budget %d bits\n", n_rows,
                    dec.invariant_noise_budget(reply, 3));
        EXPECT(dec.invariant_noise_budget(reply, 3) > 0);
        bool threw = false;
        try { pir::encode_query(lay, ctx.t(), n_rows, plain.data()); } catch (const std::out_of_range &) { threw = true; }
        EXPECT(threw);
        // more polynomials than one query selects among (forced here: 8 per column -> 4 columns, 3 expansion rounds; 1M rows at
        // N = 8192 is the same with 8192 per column): one reply ciphertext per column, the client keeps its own
        {
            pir::Database db4(ctx, base.data(), n_rows, d, 8);
            const pir::Layout &l4 = db4.layout();
            EXPECT(l4.n_sel == 8 && l4.n_cols == 4 && l4.levels == 3);
            const size_t want4[3] = {7, 8 * 32 + 5, 999};                               // columns 0, 1 and 3
            std::vector<uint64_t> p4(3 * ctx.N()), b4;
            for (size_t i = 0; i < 3; ++i) pir::encode_query(l4, ctx.t(), want4[i], p4.data() + i * ctx.N());
            bfv::Ciphertexts q4, r4;
            enc.encrypt(p4.data(), 3, q4);
            pir::answer(ctx, db4, q4, keys, r4);
            EXPECT(r4.count == 12);
            dec.decrypt(r4, b4);
            for (size_t i = 0; i < 3; ++i) {
                float row[128];
                pir::decode_row(l4, b4.data() + i * l4.n_cols * ctx.N(), want4[i], row);
                EXPECT(std::memcmp(row, base.data() + want4[i] * d, sizeof row) == 0);
            }
            std::printf("PIR with 4 columns: 3 rows retrieved, 4 reply ciphertexts each\n");
        }
    }
    if (fails) std::printf("test_bfv: %d FAILURES\n", fails);
    else std::printf("test_bfv: OK\n");
    return fails ? 1 : 0;
}
