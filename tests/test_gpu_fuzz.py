"""Randomised parity of the polynomial path against the oracle: primes of every width the three arithmetic families take
(including the ones just below and just above the family thresholds 2^44 / 2^45 / 2^56 and the largest supported ones),
ring degrees 1024 .. 32768, mixed-width modulus sets, every ct x pt flag combination, edge-value polynomials."""
import numpy as np
import pytest

import oracle
from conftest import edge_poly

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
sympy = pytest.importorskip("sympy")


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    return "cuda:0"


def ntt_prime_below(bound, N, rng, spread=4096):
    """a prime = 1 mod 2N below `bound`, picked among the first few thousand candidates"""
    m = 2 * N
    c = (bound - 2) // m * m + 1
    spread = max(1, min(spread, bound // (4 * m)))            # stay in the upper quarter below the bound
    c -= int(rng.integers(0, spread)) * m
    while not sympy.isprime(c):
        c -= m
        assert c > m, "no NTT prime below the bound"
    return int(c)


def ntt_prime_above(bound, N):
    m = 2 * N
    c = (bound // m + 1) * m + 1
    while not sympy.isprime(c):
        c += m
    return int(c)


CASES = []
_rng = np.random.default_rng(20250801)
for _N in (1024, 2048, 4096, 8192, 16384, 32768):
    _th = 44 if _N == 32768 else 45
    _sets = [
        [ntt_prime_below(1 << _th, _N, _rng, 1), ntt_prime_below(1 << _th, _N, _rng)],          # largest exact-FP64 primes
        [ntt_prime_above(1 << _th, _N), ntt_prime_below(1 << 50, _N, _rng)],                      # just above: lazy 64-bit family
        [ntt_prime_below(1 << 56, _N, _rng, 1), ntt_prime_below(1 << int(_rng.integers(30, 56)), _N, _rng)],
        [ntt_prime_above(1 << 56, _N), ntt_prime_below(1 << 44, _N, _rng)],                      # just above 2^56: Harvey family
        [ntt_prime_below(1 << 61, _N, _rng, 1), ntt_prime_below(1 << 61, _N, _rng)],              # the largest moduli accepted
        [ntt_prime_below(1 << int(_rng.integers(27, 61)), _N, _rng) for _ in range(3)],           # mixed widths
    ]
    for _s in _sets:
        if len(set(_s)) == len(_s):
            CASES.append((_N, _s))


@pytest.mark.parametrize("N,qs", CASES, ids=[f"N{n}-" + "-".join(str(q.bit_length()) for q in qs) for n, qs in CASES])
def test_random_moduli_parity(N, qs):
    import prefhetch_amd as pf
    dev = _dev()
    L = len(qs)
    rng = np.random.default_rng(N + sum(qs) % 1000)
    o = oracle.Oracle(N, qs)
    ctx = pf.RnsContext(N, qs, dev)
    fam = ctx.info()["arith_path"][0]
    th = 44 if N == 32768 else 45
    expect = 0 if all(q < 1 << th for q in qs) else (2 if all(q < 1 << 56 for q in qs) else 1)
    assert fam == expect
    B = 2
    ct = np.stack([np.stack([np.stack([edge_poly(rng, N, q, kind) for q in qs]) for kind in (0, 1)]) for _ in range(B)])   # [B][2][L][N]
    ct[1] = np.stack([np.stack([edge_poly(rng, N, q, kind) for q in qs]) for kind in (3, 0)])
    pt = np.stack([np.stack([edge_poly(rng, N, q, kind) for q in qs]) for kind in (0, 1)])                                    # [B][L][N]
    d_ct, d_pt = pf.to_device_u64(ct, dev), pf.to_device_u64(pt, dev)
    f = ctx.ntt_forward(d_ct)
    assert (pf.to_host_u64(f) == o.ntt_forward(ct)).all()
    assert (pf.to_host_u64(ctx.ntt_inverse(f)) == ct).all()
    pt_ntt = o.ntt_forward(pt)
    d_ptn = pf.to_device_u64(pt_ntt, dev)
    for flags in range(8):
        src = pf.to_host_u64(f) if flags & 2 else ct
        acc0 = np.stack([np.stack([np.stack([edge_poly(rng, N, q, 0) for q in qs]) for _ in range(2)]) for _ in range(B)])
        d_out = pf.to_device_u64(acc0, dev)
        ctx.ct_pt_mul(pf.to_device_u64(src, dev), d_ptn, out=d_out, flags=flags)
        assert (pf.to_host_u64(d_out) == o.ct_pt_mul(src, pt_ntt, flags, acc=acc0)).all(), flags
    a, b = ct[0], ct[1]
    assert (pf.to_host_u64(ctx.dyadic_mul(pf.to_device_u64(a, dev), pf.to_device_u64(b, dev))) == o.dyadic_mul(a, b)).all()


KS_CASES = []
for _N, _D in ((1024, 1), (2048, 3), (4096, 2), (8192, 5), (16384, 2), (32768, 3)):
    _th = 44 if _N == 32768 else 45
    for _hi in (_th, 56, 61):                       # one key-modulus set per arithmetic family
        _qs = []
        while len(_qs) < _D + 1:
            _q = ntt_prime_below(1 << int(_rng.integers(max(30, _hi - 10), _hi + 1)), _N, _rng)
            if _q not in _qs:
                _qs.append(_q)
        KS_CASES.append((_N, _qs))


@pytest.mark.parametrize("N,qs", KS_CASES, ids=[f"N{n}-" + "-".join(str(q.bit_length()) for q in qs) for n, qs in KS_CASES])
def test_random_moduli_key_switch(N, qs):
    """pf_key_switch against the oracle for random key-modulus sets (last prime = special prime) of every family."""
    import prefhetch_amd as pf
    dev = _dev()
    K, D, B = len(qs), len(qs) - 1, 3
    rng = np.random.default_rng(N * 7 + K)
    o = oracle.Oracle(N, qs)
    ctx = pf.RnsContext(N, qs, dev)
    target = np.stack([rng.integers(0, q, (B, N), dtype=np.uint64) for q in qs[:D]], axis=1)
    target[0, :, 0] = np.array(qs[:D], dtype=np.uint64) - 1
    target[1] = 0
    ksk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(D)])
    ct = np.stack([rng.integers(0, q, (B, 2, N), dtype=np.uint64) for q in qs[:D]], axis=2)
    d_ct = pf.to_device_u64(ct, dev)
    ctx.key_switch_(pf.to_device_u64(target, dev), pf.to_device_u64(ksk, dev), d_ct)
    assert (pf.to_host_u64(d_ct) == o.key_switch(target, ksk, ct)).all()
