"""CPU tests of the oracle itself: Oracle A (C, SEAL-style lazy butterflies) against Oracle B (big-int
mathematical definition), the committed golden vectors, the SURVEY prime/root table, and algebraic
properties.  The reference ships no tests or fixtures for this path, so this is the
pinning the oracle gets."""
import numpy as np
import pytest

import oracle
from oracle import bigint_ref as B
from conftest import edge_poly


def test_bfv_default_primes_and_minimal_roots():
    for N, qs in oracle.BFV_DEFAULT.items():
        for q in qs:
            assert B.is_prime(q) and (q - 1) % (2 * N) == 0
        o = oracle.Oracle(N, qs)
        if N in oracle.MIN_PSI:
            assert [o.psi(l) for l in range(len(qs))] == oracle.MIN_PSI[N]
    bits = {N: sum(q.bit_length() for q in qs) for N, qs in oracle.BFV_DEFAULT.items()}
    assert (bits[1024], bits[4096], bits[8192], bits[32768]) == (27, 109, 218, 881)   # SEAL BFVDefault totals


def test_minimal_root_matches_bruteforce():
    for N, q in [(8, 0x7E00001), (16, 0xFFFFEE001), (64, 0x7FFFFFD8001), (128, 0x7FFFFFFFE90001)]:
        assert oracle.Oracle(N, [q]).psi(0) == B.min_psi(N, q)


def test_golden_vectors(golden):
    for ci in range(int(golden["n_cases"])):
        g = lambda k: golden[f"c{ci}_{k}"]
        N, q = int(g("N")), int(g("q"))
        o = oracle.Oracle(N, [q])
        assert o.psi(0) == int(g("psi"))
        assert (o.ntt_forward(g("a")) == g("ntt_a")).all()
        assert (o.ntt_forward(g("b")) == g("ntt_b")).all()
        assert (o.ntt_inverse(g("ntt_a")) == g("a")).all()
        assert (o.ntt_inverse(o.dyadic_mul(g("ntt_a"), g("ntt_b"))) == g("a_times_b")).all()
        assert (o.addsub(g("a"), g("b"), o.ADD) == g("a_plus_b")).all()
        assert (o.addsub(g("a"), g("b"), o.SUB) == g("a_minus_b")).all()
        assert (o.addsub(g("a"), None, o.NEG) == g("neg_a")).all()
        ct = np.stack([g("a"), g("b")]).reshape(1, 2, 1, N)
        out = o.ct_pt_mul(ct, g("ntt_b").reshape(1, 1, N)).reshape(2, N)
        assert (out[0] == g("a_times_b")).all()


@pytest.mark.parametrize("N,q", [(256, 0x7E00001), (512, 0xFFFFEE001), (256, 0x7FFFFFD8001), (256, 0x7FFFFFFFE90001)])
def test_ntt_against_direct_evaluation(N, q):
    rng = np.random.default_rng(N + q % 1000)
    o = oracle.Oracle(N, [q])
    for kind in range(5):
        a = edge_poly(rng, N, q, kind)
        assert [int(x) for x in o.ntt_forward(a)] == B.ntt_direct([int(x) for x in a], q, o.psi(0))
        assert (o.ntt_inverse(o.ntt_forward(a)) == a).all()


def test_properties_full_size():
    """N = 8192, 4 limbs: round trip, linearity, X-shift (multiplication by the monomial X)."""
    N, qs = 8192, oracle.BFV_DEFAULT[8192][:4]
    o = oracle.Oracle(N, qs)
    rng = np.random.default_rng(3)
    a = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs])
    b = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs])
    A, Bn = o.ntt_forward(a), o.ntt_forward(b)
    assert (o.ntt_inverse(A) == a).all()
    assert (o.ntt_forward(o.addsub(a, b, o.ADD)) == o.addsub(A, Bn, o.ADD)).all()
    x = np.zeros_like(a)
    x[:, 1] = 1                                            # the monomial X
    shifted = o.ntt_inverse(o.dyadic_mul(A, o.ntt_forward(x)))
    expect = np.roll(a, 1, axis=1)
    for l, q in enumerate(qs):
        expect[l, 0] = (q - int(a[l, -1])) % q             # X^N = -1
    assert (shifted == expect).all()


def test_multi_limb_ct_pt_against_crt():
    """Two-limb product checked through CRT reconstruction against a big-int negacyclic product."""
    N, qs = 64, [0x7FFFFFD8001, 0xFFFFFFFC001]
    assert all((q - 1) % (2 * N) == 0 for q in qs)
    o = oracle.Oracle(N, qs)
    rng = np.random.default_rng(11)
    M = qs[0] * qs[1]
    big_a = [int(rng.integers(0, 2**62)) * int(rng.integers(0, 2**20)) % M for _ in range(N)]
    big_b = [int(rng.integers(0, 2**62)) % M for _ in range(N)]
    a = np.array([[x % q for x in big_a] for q in qs], dtype=np.uint64)
    b = np.array([[x % q for x in big_b] for q in qs], dtype=np.uint64)
    ct = np.stack([a, a]).reshape(1, 2, 2, N)
    out = o.ct_pt_mul(ct, o.ntt_forward(b).reshape(1, 2, N)).reshape(2, 2, N)
    ref = B.negacyclic_mul(big_a, big_b, M)
    for j in range(N):
        x, _ = B.crt([int(out[0, 0, j]), int(out[0, 1, j])], qs)
        assert x == ref[j]


def test_ct_pt_flags():
    N, qs = 1024, [0x7E00001]
    o = oracle.Oracle(N, qs)
    rng = np.random.default_rng(5)
    ct = rng.integers(0, qs[0], (3, 2, 1, N), dtype=np.uint64)
    pt = o.ntt_forward(rng.integers(0, qs[0], (3, 1, N), dtype=np.uint64))
    acc = rng.integers(0, qs[0], (3, 2, 1, N), dtype=np.uint64)
    plain = o.ct_pt_mul(ct, pt)
    assert (o.ct_pt_mul(o.ntt_forward(ct), pt, o.IN_NTT) == plain).all()
    assert (o.ntt_inverse(o.ct_pt_mul(ct, pt, o.OUT_NTT)) == plain).all()
    assert (o.ct_pt_mul(ct, pt, o.ACCUMULATE, acc=acc) == o.addsub(acc, plain, o.ADD)).all()
    assert (o.ct_pt_mul(ct, pt[:1]) [0] == plain[0]).all()     # broadcast plaintext


def test_precise_search_matches_reference_semantics():
    """server_lib.cpp:151-162: float accumulator, pow() through double; exact on SIFT-like integer data."""
    rng = np.random.default_rng(9)
    base = rng.integers(0, 256, (500, 128)).astype(np.float32)
    xq = rng.integers(0, 256, (5, 128)).astype(np.float32)
    ids = rng.integers(0, 500, (5, 200)).astype(np.int64)
    got = oracle.precise_search(base, xq, ids)
    exact = ((base[ids].astype(np.float64) - xq[:, None, :].astype(np.float64)) ** 2).sum(-1)
    assert (got.astype(np.float64) == exact).all()
    # literal python restatement of the loop on gaussian data
    gb, gq = rng.standard_normal((50, 16)).astype(np.float32), rng.standard_normal((2, 16)).astype(np.float32)
    gi = rng.integers(0, 50, (2, 7)).astype(np.int64)
    got = oracle.precise_search(gb, gq, gi)
    for i in range(2):
        for j in range(7):
            dist = np.float32(0.0)
            for k in range(16):
                diff = np.float32(gb[gi[i, j], k] - gq[i, k])
                dist = np.float32(np.float64(dist) + np.float64(diff) ** 2)
            assert got[i, j] == dist


def test_flat_l2_search_oracle():
    rng = np.random.default_rng(2)
    xb = rng.integers(0, 256, (3000, 128)).astype(np.float32)
    xq = rng.integers(0, 256, (7, 128)).astype(np.float32)
    D, I = oracle.flat_l2_search(xb, xq, 20)
    d = ((xb[None].astype(np.float64) - xq[:, None].astype(np.float64)) ** 2).sum(-1)
    for i in range(7):
        order = np.lexsort((np.arange(3000), d[i]))[:20]
        assert (I[i] == order).all() and (D[i].astype(np.float64) == d[i][order]).all()
    D2, I2 = oracle.flat_l2_search(xb, xq, 20, mode=1)      # literal client_lib.cpp:55-67 accumulation
    assert (I2 == I).all() and (D2 == D).all()
    D3, I3 = oracle.flat_l2_search(xb[:5], xq, 8)           # k > nb: (+inf, -1) padding
    assert (I3[:, 5:] == -1).all() and np.isinf(D3[:, 5:]).all()


def test_key_switch_oracle_a_vs_bigint():
    """pfo_key_switch (SEAL switch_key_inplace restated, NTT-domain products) against the coefficient-domain
    big-integer restatement: RNS digits, one special prime, division by P with rounding."""
    N, qs = 64, [0x7FFFFFFFE90001, 0x7FFFFFFFBF0001, 0xFFFFFFFFF70001]       # 2 data primes + special, all = 1 mod 128
    o = oracle.Oracle(N, qs)
    psis = [o.psi(l) for l in range(3)]
    rng = np.random.default_rng(3)
    L, K = 2, 3
    target = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    target[0, 0], target[1, 1] = qs[0] - 1, qs[1] - 1
    ksk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(L)])
    ct = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]]) for _ in range(2)])
    got = o.key_switch(target.reshape(1, L, N), ksk, ct.reshape(1, 2, L, N)).reshape(2, L, N)
    as_int = lambda a: [int(x) for x in a]
    ref = B.key_switch([as_int(r) for r in target], [[[as_int(ksk[i][c][j]) for j in range(K)] for c in range(2)] for i in range(L)],
                       [[as_int(ct[c][j]) for j in range(L)] for c in range(2)], qs, psis)
    assert (got == np.array(ref, dtype=np.uint64)).all()
    # linearity in the target: switching t1 + t2 equals switching t1 then t2 up to the rounding of the division by P
    t2 = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
    zero = np.zeros((1, 2, L, N), np.uint64)
    a = o.key_switch(target.reshape(1, L, N), ksk, zero)
    b = o.key_switch(t2.reshape(1, L, N), ksk, zero)
    tsum = np.stack([(target[l].astype(object) + t2[l].astype(object)) % qs[l] for l in range(L)]).astype(np.uint64)
    # digits of (t1+t2) differ from digit sums only by multiples of q_I, so compare through the exact big-int model
    ref_sum = B.key_switch([as_int(r) for r in tsum], [[[as_int(ksk[i][c][j]) for j in range(K)] for c in range(2)] for i in range(L)],
                           [[[0] * N for _ in range(L)] for _ in range(2)], qs, psis)
    assert (o.key_switch(tsum.reshape(1, L, N), ksk, zero).reshape(2, L, N) == np.array(ref_sum, dtype=np.uint64)).all()
    assert a.shape == b.shape


def test_barrett_dyadic_against_python_modulo():
    """SURVEY 8c known-answer test 7: 128->64 Barrett product vs Python's %, including operands q-1, for every
    BFVDefault prime (27 to 56 bits)."""
    rng = np.random.default_rng(77)
    for N, qs in oracle.BFV_DEFAULT.items():
        for q in qs:
            o = oracle.Oracle(N, [q]) if (q - 1) % (2 * N) == 0 else None
            if o is None:
                continue
            a = rng.integers(0, q, N, dtype=np.uint64)
            b = rng.integers(0, q, N, dtype=np.uint64)
            a[:4] = [q - 1, q - 1, 0, 1]
            b[:4] = [q - 1, 1, q - 1, q - 1]
            got = o.dyadic_mul(a, b)
            assert [int(x) for x in got[:64]] == [int(x) * int(y) % q for x, y in zip(a[:64], b[:64])]
            assert int(got[0]) == (q - 1) * (q - 1) % q


try:
    from hypothesis import given, settings, strategies as st
    _HAVE_HYPOTHESIS = True
except Exception:        # pragma: no cover
    _HAVE_HYPOTHESIS = False


if _HAVE_HYPOTHESIS:
    _PRIMES_N64 = [q for qs in oracle.BFV_DEFAULT.values() for q in qs if (q - 1) % 128 == 0]

    @settings(max_examples=25, deadline=None)
    @given(st.sampled_from(_PRIMES_N64), st.integers(0, 2**32 - 1))
    def test_hypothesis_random_polys_oracle_a_vs_b(q, seed):
        """SURVEY 8c known-answer test 6: random polynomials per prime, lazy-butterfly C oracle against the direct
        big-integer evaluation (N = 64 keeps the O(N^2) side fast)."""
        N = 64
        rng = np.random.default_rng(seed)
        o = oracle.Oracle(N, [q])
        a = rng.integers(0, q, N, dtype=np.uint64)
        b = rng.integers(0, q, N, dtype=np.uint64)
        A = o.ntt_forward(a)
        assert [int(x) for x in A] == B.ntt_direct([int(x) for x in a], q, o.psi(0))
        prod = o.ntt_inverse(o.dyadic_mul(A, o.ntt_forward(b)))
        assert [int(x) for x in prod] == B.negacyclic_mul([int(x) for x in a], [int(x) for x in b], q)


def test_pack_rows_encoding_yields_inner_products():
    """the packing of the encrypted precise search: coefficient d*j of q(X) p(X) mod (X^N + 1) is <q, row_j>
    (checked with the big-integer schoolbook product, no NTT involved)"""
    from oracle import bigint_ref as B
    rng = np.random.default_rng(11)
    N, d, rows = 1024, 128, 8
    base = rng.integers(0, 256, (40, d)).astype(np.float32)
    ids = np.array([[3, 39, 0, 17, 5, 5, 21, 8], [1, 2, -1, 4, 400, 6, 7, 9]], dtype=np.int64)   # -1 / 400: outside the base
    qmod = 0x7E00001
    packed = oracle.pack_rows(base, ids, N, [qmod])
    query = rng.integers(0, 256, d)
    qpoly = [int(v) for v in query] + [0] * (N - d)
    for p in range(2):
        coeff = [int(c) if int(c) < qmod // 2 else int(c) - qmod for c in packed[p, 0]]           # centred lift
        prod = B.negacyclic_mul(qpoly, coeff, 1 << 62)
        for j in range(rows):
            rid = int(ids[p, j])
            want = int(np.dot(query, base[rid].astype(np.int64))) if 0 <= rid < 40 else 0
            got = prod[d * j] if prod[d * j] < (1 << 61) else prod[d * j] - (1 << 62)
            assert got == want, (p, j)
