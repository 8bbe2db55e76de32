// time_pir.cpp -- the private row retrieval (include/client/pir.h) at size: build the database, expand one query, answer
// `n_query` of them, check the rows bit for bit and print the times.  Usage: time_pir [n_rows (default 262144: the most one
// column holds at N = 8192, d = 128; 1000000 = the reference's base, 4 columns)] [n_query (default 2)].  Built by `make -C prefhetch_amd/csrc time_pir`.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../include/client/pir.h"
#include "../include/prefhetch_hip.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const size_t n_rows = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 262144;
    const size_t n_query = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 2;
    const uint32_t d = 128;
    bfv::Context ctx(bfv::Params::seal_default(8192, 65537));
    bfv::KeyGenerator keygen(ctx, bfv::seeded_random(11));
    bfv::PublicKey pk = keygen.create_public_key();
    bfv::Encryptor enc(ctx, pk, bfv::seeded_random(12));
    bfv::Decryptor dec(ctx, keygen.secret_key());
    std::vector<float> base(n_rows * d);
    std::mt19937_64 rng(5);
    for (auto &v : base) v = (float)((double)(int64_t)(rng() % 2000001) / 1000.0 - 1000.0);
    double t0 = now_ms();
    pir::Database db(ctx, base.data(), n_rows, d);
    const double db_ms = now_ms() - t0;
    const pir::Layout &lay = db.layout();
    t0 = now_ms();
    std::vector<bfv::SwitchKey> keys;
    for (uint32_t g : pir::galois_elements(ctx.N(), lay.levels)) keys.push_back(keygen.create_galois_key(g));
    const double key_ms = now_ms() - t0;
    std::vector<size_t> wanted(n_query);
    for (size_t i = 0; i < n_query; ++i) wanted[i] = i == 0 ? n_rows - 1 : rng() % n_rows;
    std::vector<uint64_t> plain(n_query * ctx.N()), back;
    for (size_t i = 0; i < n_query; ++i) pir::encode_query(lay, ctx.t(), wanted[i], plain.data() + i * ctx.N());
    bfv::Ciphertexts query, reply, one, sel;
    enc.encrypt(plain.data(), n_query, query);
    // warm-up (workspaces are allocated on first use), then the timed calls
    enc.encrypt(plain.data(), 1, one);
    pir::expand(ctx, one, keys, lay.levels, sel);
    t0 = now_ms();
    pir::expand(ctx, one, keys, lay.levels, sel);
    const double expand_ms = now_ms() - t0;
    pir::answer(ctx, db, query, keys, reply);
    t0 = now_ms();
    pir::answer(ctx, db, query, keys, reply);
    const double answer_ms = (now_ms() - t0) / (double)n_query;
    dec.decrypt(reply, back);
    int bad = 0;
    for (size_t i = 0; i < n_query; ++i) {
        float row[128];
        pir::decode_row(lay, back.data() + i * lay.n_cols * ctx.N(), wanted[i], row);
        bad += std::memcmp(row, base.data() + wanted[i] * d, sizeof row) != 0;
    }
    const int budget = dec.invariant_noise_budget(reply, reply.count - 1);
    std::printf("{\"n_rows\": %zu, \"d\": %u, \"polynomials\": %zu, \"levels\": %u, \"columns\": %zu, \"database_build_ms\": %.1f, \"galois_keygen_ms\": %.1f, "
                "\"expand_ms\": %.2f, \"answer_ms_per_query\": %.2f, \"key_switches_per_query\": %zu, \"products_per_query\": %zu, "
                "\"rows_bit_identical\": %s, \"reply_noise_budget_bits\": %d, \"query_bytes\": %zu, \"reply_bytes\": %zu}\n",
                n_rows, d, lay.n_polys, lay.levels, lay.n_cols, db_ms, key_ms, expand_ms, answer_ms, (size_t{1} << lay.levels) - 1, lay.n_polys,
                bad ? "false" : "true", budget, 2 * ctx.L() * ctx.N() * 8, lay.n_cols * 2 * ctx.L() * ctx.N() * 8);
    return bad || budget <= 0;
}
