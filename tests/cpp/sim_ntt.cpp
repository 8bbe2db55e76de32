// sim_ntt.cpp -- host execution of the device NTT core (prefhetch_amd/csrc/ntt_core.hpp), one OS
// thread per GPU lane, std::barrier for s_barrier.  TEST INFRASTRUCTURE: checks index maps, LDS
// slot permutations and the FP64 error analysis bit-for-bit against the oracle without a GPU.
// Build: g++ -std=c++20 -O2 -mfma -ffp-contract=off -pthread -shared -fPIC
#include <barrier>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../prefhetch_amd/csrc/ntt_core.hpp"
#include "../../prefhetch_amd/csrc/tables.hpp"

using namespace pf;

template <class G, class A, class Body>
static void run_wg(Body &&body) {
    std::vector<typename A::V> lds(Xchg<G, A>::LDS_ENTRIES);
    std::barrier bar(G::T);
    std::vector<std::thread> th;
    th.reserve(G::T);
    for (int tid = 0; tid < G::T; ++tid)
        th.emplace_back([&, tid] { auto sync = [&] { bar.arrive_and_wait(); }; body(lds.data(), tid, sync); });
    for (auto &t : th) t.join();
}

template <int LOGN, class A>
static void run_op(int op, int flags, const A &ar, const typename A::Tw *tw, const typename A::Tw *itw,
                   const uint64_t *src, const uint64_t *pt, uint64_t *dst) {
    using G = Geo<LOGN>;
    run_wg<G, A>([&](typename A::V *lds, int tid, auto &sync) {
        if (op == 0) body_ntt_fwd<G, A>(ar, tw, src, dst, lds, tid, sync);
        else if (op == 1) body_ntt_inv<G, A>(ar, itw, src, dst, lds, tid, sync);
        else {
            switch (flags) {
#define CASE(F) case F: body_ctpt<G, A, F>(ar, tw, itw, src, pt, dst, lds, tid, sync); break;
                CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7)
#undef CASE
            }
        }
    });
}

template <class A>
static void dispatch(int logn, int op, int flags, const A &ar, const typename A::Tw *tw, const typename A::Tw *itw,
                     const uint64_t *src, const uint64_t *pt, uint64_t *dst) {
    switch (logn) {
        case 10: run_op<10, A>(op, flags, ar, tw, itw, src, pt, dst); break;
        case 11: run_op<11, A>(op, flags, ar, tw, itw, src, pt, dst); break;
        case 12: run_op<12, A>(op, flags, ar, tw, itw, src, pt, dst); break;
        case 13: run_op<13, A>(op, flags, ar, tw, itw, src, pt, dst); break;
        case 14: run_op<14, A>(op, flags, ar, tw, itw, src, pt, dst); break;
        case 15: run_op<15, A>(op, flags, ar, tw, itw, src, pt, dst); break;
    }
}

// op: 0 fwd, 1 inv, 2 ctpt(flags).  arith: 0 f64, 1 u64 (Harvey), 2 u64 lazy (q < 2^56).  One limb-polynomial per call.
extern "C" int pf_sim_run(int logn, uint64_t q, int arith, int op, int flags, const uint64_t *src,
                          const uint64_t *pt, uint64_t *dst) {
    LimbTables t;
    std::string err;
    if (logn < 10 || logn > 15) return -1;
    if (!build_limb_tables(1u << logn, q, t, err)) return -2;
    if (arith == 0) {
        if (!t.f64_ok) return -3;
        ArithF64 ar{(double)q, 1.0 / (double)q};
        dispatch<ArithF64>(logn, op, flags, ar, t.fwd_f.data(), t.inv_f.data(), src, pt, dst);
    } else if (arith == 2) {
        if (!u64_lazy_ok(q, logn)) return -3;
        ArithU64L ar{q, 2 * q, t.ratio0, t.ratio1};
        dispatch<ArithU64L>(logn, op, flags, ar, t.fwd_u.data(), t.inv_u.data(), src, pt, dst);
    } else {
        ArithU64 ar{q, 2 * q, t.ratio0, t.ratio1};
        dispatch<ArithU64>(logn, op, flags, ar, t.fwd_u.data(), t.inv_u.data(), src, pt, dst);
    }
    return 0;
}

// number of range-analysis violations seen so far by the 64-bit lazy butterflies (must stay 0)
extern "C" unsigned long long pf_sim_range_violations() { return pf::pf_range_violations; }

extern "C" uint64_t pf_sim_psi(int logn, uint64_t q) {
    LimbTables t; std::string err;
    return build_limb_tables(1u << logn, q, t, err) ? t.psi : 0;
}

// ---- the two-pass key switch at N = 32768 (prefhetch_amd/csrc/ks_split.hpp), one OS thread per lane.  moduli[K] (special prime
// last), target [D][N], ksk [D][2][K][N].  ct == nullptr: passes A + B<INV = false> -> acc [2][K][N] (NTT form, canonical): what
// k_ksA + k_ksB<false> compute for one ciphertext.  ct != nullptr ([2][D][N], updated in place): passes A + B<INV = true> + C, the
// whole of pf_key_switch's fused path (acc is scratch).
#include "../../prefhetch_amd/csrc/ks_split.hpp"

template <int T, class Body>
static void run_threads(Body &&body) {
    std::barrier bar(T);
    std::vector<std::thread> th;
    th.reserve(T);
    for (int tid = 0; tid < T; ++tid) th.emplace_back([&, tid] { auto sync = [&] { bar.arrive_and_wait(); }; body(tid, sync); });
    for (auto &t : th) t.join();
}

extern "C" int pf_sim_ks_split(int D, int K, const uint64_t *moduli, const uint64_t *target, const uint64_t *ksk, uint64_t *acc, uint64_t *ct) {
    constexpr size_t N = KsGeo::N;
    std::vector<LimbTables> tabs(K);
    std::string err;
    for (int j = 0; j < K; ++j) {
        if (!u64_lazy_ok(moduli[j], 15)) return -3;
        if (!build_limb_tables((uint32_t)N, moduli[j], tabs[j], err)) return -2;
    }
    std::vector<uint64_t> x((size_t)D * K * N);
    for (int I = 0; I < D; ++I)
        for (int J = 0; J < K; ++J) {
            const ArithU64L ar{moduli[J], 2 * moduli[J], tabs[J].ratio0, tabs[J].ratio1};
            for (int cb = 0; cb < KsGeo::A_TILES; ++cb) {
                std::vector<uint64_t> lds(KsGeo::A_LDS);
                run_threads<KsGeo::A_T>([&](int tid, auto &sync) {
                    body_ksA<ArithU64L>(ar, tabs[J].fwd_u.data(), target + (size_t)I * N, x.data() + ((size_t)I * K + J) * N, cb, lds.data(), tid, sync);
                });
            }
        }
    for (int J = 0; J < K; ++J) {
        const ArithU64L ar{moduli[J], 2 * moduli[J], tabs[J].ratio0, tabs[J].ratio1};
        for (int chunk = 0; chunk < KsGeo::B_CHUNKS; ++chunk) {
            std::vector<uint64_t> lds(KsGeo::B_LDS + 2);
            uint64_t *l16 = reinterpret_cast<uint64_t *>((reinterpret_cast<uintptr_t>(lds.data()) + 15) & ~uintptr_t(15));
            run_threads<KsGeo::B_T>([&](int tid, auto &wsync) {          // a barrier over the whole workgroup is a (stronger) wave barrier
                if (ct) body_ksB<ArithU64L, true>(ar, tabs[J].fwd_u.data(), tabs[J].inv_u.data(), x.data() + (size_t)J * N, (size_t)K * N, ksk + (size_t)J * N,
                                                  (size_t)K * N, acc + (size_t)J * N, acc + ((size_t)K + J) * N, D, chunk, l16, tid, wsync);
                else body_ksB<ArithU64L, false>(ar, tabs[J].fwd_u.data(), tabs[J].inv_u.data(), x.data() + (size_t)J * N, (size_t)K * N, ksk + (size_t)J * N,
                                                (size_t)K * N, acc + (size_t)J * N, acc + ((size_t)K + J) * N, D, chunk, l16, tid, wsync);
            });
        }
    }
    if (!ct) return 0;
    const uint64_t P = moduli[K - 1];
    auto limb = [&](int J) {
        KsLimbC c{moduli[J], tabs[J].ratio0, tabs[J].ratio1, 0, 0, 0, tabs[J].inv_u.data()};
        if (J + 1 < K) {
            c.half_mod = (P >> 1) % moduli[J];
            c.pinv = h_powmod(P % moduli[J], moduli[J] - 2, moduli[J]);
            c.pinv_quot = (uint64_t)((((u128_t)c.pinv) << 64) / moduli[J]);
        }
        return c;
    };
    for (int comp = 0; comp < 2; ++comp)
        for (int j0 = 0; j0 < D; j0 += 2)                               // limb groups of two (the kernel's groups are larger)
            for (int cb = 0; cb < KsGeo::A_TILES; ++cb) {
                std::vector<uint64_t> lds(KsGeo::A_LDS);
                run_threads<KsGeo::A_T>([&](int tid, auto &sync) {
                    body_ksC<ArithU64L>(limb, acc, ct, comp, D, K, j0, j0 + 2 < D ? j0 + 2 : D, cb, lds.data(), tid, sync);
                });
            }
    return 0;
}
