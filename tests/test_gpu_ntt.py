"""GPU parity tests (run on MI355X with -m gpu): HIP kernels called through the C ABI against the CPU
oracle on the same seeded inputs -- bit-exact, since this is integer modular arithmetic."""
import numpy as np
import pytest

import oracle
from conftest import edge_poly

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device (no silent CPU fallback)")
    return "cuda:0"


@pytest.fixture(scope="module")
def pf():
    import prefhetch_amd
    return prefhetch_amd


def _ctx(pf, N, qs, force_u64=0):
    c = pf.RnsContext(N, qs, _dev())
    if force_u64:
        c.force_u64(int(force_u64))
    return c


CONFIGS = [  # (N, moduli, force_u64: 0 automatic, 1 64-bit integer butterflies, 2 general Harvey butterflies)
    (1024, oracle.BFV_DEFAULT[1024], 2),
    (8192, oracle.BFV_DEFAULT[8192], 2),
    (8192, [0x7FFFFFFFE90001, 0x7FFFFFFFBF0001], 2),
    (16384, [0xFFFFFFFFFFC0001, 0x7FFFFFFFE90001], False),    # a 60-bit prime: general Harvey butterflies automatically
    (32768, [0x1FFFFFFFFFE10001, 0xFFFFFFFFF70001], False),   # a 61-bit prime at N=32768
    (32768, oracle.BFV_DEFAULT[32768][:2] + oracle.BFV_DEFAULT[32768][-1:], 2),
    (1024, oracle.BFV_DEFAULT[1024], False),
    (1024, oracle.BFV_DEFAULT[1024], True),
    (2048, oracle.BFV_DEFAULT[2048], False),              # 54-bit prime -> u64 path automatically
    (4096, oracle.BFV_DEFAULT[4096][:2], False),
    (4096, oracle.BFV_DEFAULT[4096], True),
    (8192, oracle.BFV_DEFAULT[8192][:4], False),
    (8192, oracle.BFV_DEFAULT[8192], True),
    (8192, [0x7FFFFFFFE90001, 0x7FFFFFFFBF0001], False),   # 55-bit primes at N=8192 -> u64 path
    (16384, [0x7FFFFFFFE90001, 0x7FFFFFD8001], False),  # mixed widths -> u64 path
    (16384, [0x7FFFFFD8001, 0x7FFFFFC8001], False),      # exact-FP64 path at N=16384
    (32768, oracle.BFV_DEFAULT[32768][:2] + oracle.BFV_DEFAULT[32768][-1:], False),   # config 5 ring: 55/56-bit primes, u64 path
    (32768, [0x7FFFFDB0001, 0x7FFFFD20001], False),          # 43-bit primes at N=32768: exact-FP64 path, 64 coefficients per thread
]


@pytest.mark.parametrize("N,qs,force", CONFIGS)
def test_ntt_roundtrip_and_parity(pf, N, qs, force):
    L = len(qs)
    rng = np.random.default_rng(N + L + force)
    o = oracle.Oracle(N, qs)
    c = _ctx(pf, N, qs, force)
    info = c.info()
    assert info["psi"] == [o.psi(l) for l in range(L)]
    wide = force == 2 or any(q >= 1 << 56 for q in qs)
    expect_path = (1 if wide else 2) if (force or any(q >= 1 << (44 if N == 32768 else 45) for q in qs)) else 0
    assert info["arith_path"] == [expect_path] * L
    polys = np.stack([np.stack([edge_poly(rng, N, q, kind) for q in qs]) for kind in (0, 1, 2, 3, 4, 0, 0)])  # [7][L][N]
    d = pf.to_device_u64(polys, _dev())
    c.ntt_forward_(d)
    ref = o.ntt_forward(polys)
    assert (pf.to_host_u64(d) == ref).all()
    c.ntt_inverse_(d)
    assert (pf.to_host_u64(d) == polys).all()


@pytest.mark.parametrize("N,qs,force", CONFIGS)
def test_elementwise_parity(pf, N, qs, force):
    L = len(qs)
    rng = np.random.default_rng(N * 3 + L)
    o = oracle.Oracle(N, qs)
    c = _ctx(pf, N, qs, force)
    a = np.stack([np.stack([edge_poly(rng, N, q, kind) for q in qs]) for kind in (0, 1, 2, 4, 0)])
    b = np.stack([np.stack([edge_poly(rng, N, q, kind) for q in qs]) for kind in (0, 1, 1, 1, 3)])
    da, db = pf.to_device_u64(a, _dev()), pf.to_device_u64(b, _dev())
    assert (pf.to_host_u64(c.dyadic_mul(da, db)) == o.dyadic_mul(a, b)).all()
    assert (pf.to_host_u64(c.add(da, db)) == o.addsub(a, b, o.ADD)).all()
    assert (pf.to_host_u64(c.sub(da, db)) == o.addsub(a, b, o.SUB)).all()
    assert (pf.to_host_u64(c.negate(da)) == o.addsub(a, None, o.NEG)).all()
    # in place (out aliases a)
    c.add(da, db, out=da)
    assert (pf.to_host_u64(da) == o.addsub(a, b, o.ADD)).all()


@pytest.mark.parametrize("N,qs,force", CONFIGS)
def test_ct_pt_mul_all_flags(pf, N, qs, force):
    L = len(qs)
    B = 3
    rng = np.random.default_rng(N * 5 + L + force)
    o = oracle.Oracle(N, qs)
    c = _ctx(pf, N, qs, force)

    def rand(shape_prefix):
        return np.stack([rng.integers(0, q, shape_prefix + (N,), dtype=np.uint64) for q in qs], axis=len(shape_prefix))

    ct = rand((B, 2))                       # [B][2][L][N]
    ct[0, 0] = np.stack([np.full(N, q - 1, dtype=np.uint64) for q in qs])
    pt = o.ntt_forward(rand((B,)))          # [B][L][N] NTT form
    pt[1] = np.stack([np.full(N, q - 1, dtype=np.uint64) for q in qs])
    acc = rand((B, 2))
    ct_ntt = o.ntt_forward(ct)
    for flags in range(8):
        src = ct_ntt if flags & pf.IN_NTT else ct
        exp = o.ct_pt_mul(src, pt, flags, acc=acc)
        d_out = pf.to_device_u64(acc, _dev())
        got = c.ct_pt_mul(pf.to_device_u64(src, _dev()), pf.to_device_u64(pt, _dev()), out=d_out if flags & pf.ACCUMULATE else None, flags=flags)
        assert (pf.to_host_u64(got) == exp).all(), flags
    # broadcast plaintext, in-place output
    d_ct = pf.to_device_u64(ct, _dev())
    c.ct_pt_mul(d_ct, pf.to_device_u64(pt[:1], _dev()), out=d_ct)
    assert (pf.to_host_u64(d_ct) == o.ct_pt_mul(ct, pt[:1])).all()


def test_golden_vectors_on_gpu(pf, golden):
    for ci in (4, 5):                                      # the N = 1024 fixtures (smaller N is below the kernel range)
        g = lambda k: golden[f"c{ci}_{k}"]
        q = int(g("q"))
        for force in (0, 1, 2):
            c = _ctx(pf, 1024, [q], force)
            d = pf.to_device_u64(np.stack([g("a"), g("b")]), _dev())
            c.ntt_forward_(d)
            assert (pf.to_host_u64(d) == np.stack([g("ntt_a"), g("ntt_b")])).all()
            ct = pf.to_device_u64(np.stack([g("a"), g("b")]).reshape(1, 2, 1, 1024), _dev())
            out = pf.to_host_u64(c.ct_pt_mul(ct, pf.to_device_u64(g("ntt_b").reshape(1, 1, 1024), _dev()))).reshape(2, 1024)
            assert (out[0] == g("a_times_b")).all()
            da, db = pf.to_device_u64(g("a"), _dev()), pf.to_device_u64(g("b"), _dev())
            assert (pf.to_host_u64(c.add(da, db)) == g("a_plus_b")).all()
            assert (pf.to_host_u64(c.sub(da, db)) == g("a_minus_b")).all()
            assert (pf.to_host_u64(c.negate(da)) == g("neg_a")).all()


def test_config2_n4096_batch256(pf):
    """BASELINE config 2: N=4096, 2 limbs, batch 256 ct x pt, whole batch against the oracle."""
    N, qs, B = 4096, oracle.BFV_DEFAULT[4096][:2], 256
    rng = np.random.default_rng(20250801 + 2)
    o = oracle.Oracle(N, qs)
    c = _ctx(pf, N, qs)
    ct = np.stack([rng.integers(0, q, (B, 2, N), dtype=np.uint64) for q in qs], axis=2)
    pt = np.stack([rng.integers(0, q, (B, N), dtype=np.uint64) for q in qs], axis=1)
    got = pf.to_host_u64(c.ct_pt_mul(pf.to_device_u64(ct, _dev()), pf.to_device_u64(pt, _dev())))
    assert (got == o.ct_pt_mul(ct, pt)).all()


def test_config3_full_size_properties(pf):
    """BASELINE config 3 sizes (N=8192, 4 limbs, batch 1024): a 64-ciphertext slice against the oracle,
    the whole batch through size-independent properties (round trip, linearity, fused == unfused)."""
    N, qs, B = 8192, oracle.BFV_DEFAULT[8192][:4], 1024
    L = len(qs)
    o = oracle.Oracle(N, qs)
    c = _ctx(pf, N, qs)
    g = torch.Generator(device="cpu").manual_seed(20250801 + 3)
    qt = torch.tensor(qs, dtype=torch.int64).view(1, 1, L, 1)
    ct = (torch.randint(0, 2**62, (B, 2, L, N), generator=g, dtype=torch.int64) % qt).to(_dev())
    pt = (torch.randint(0, 2**62, (B, L, N), generator=g, dtype=torch.int64) % qt[0]).to(_dev())
    out = c.ct_pt_mul(ct, pt)
    sl = slice(480, 544)
    exp = o.ct_pt_mul(pf.to_host_u64(ct[sl]), pf.to_host_u64(pt[sl]))
    assert (pf.to_host_u64(out[sl]) == exp).all()
    # fused == unfused pipeline on the whole batch
    tmp = ct.clone()
    c.ntt_forward_(tmp)
    tmp = c.dyadic_mul(tmp, pt.unsqueeze(1).expand(B, 2, L, N).contiguous())
    c.ntt_inverse_(tmp)
    assert torch.equal(tmp, out)
    # round trip and linearity on the whole batch
    rt = ct.clone()
    c.ntt_forward_(rt)
    f_ct = rt.clone()
    c.ntt_inverse_(rt)
    assert torch.equal(rt, ct)
    other = torch.roll(ct, 1, 0).contiguous()
    s = c.add(ct, other)
    c.ntt_forward_(s)
    f_other = other.clone()
    c.ntt_forward_(f_other)
    assert torch.equal(s, c.add(f_ct, f_other))
    # accumulate: out + out == 2*out
    acc = out.clone()
    c.ct_pt_mul(ct, pt, out=acc, flags=pf.ACCUMULATE)
    assert torch.equal(acc, c.add(out, out))


def test_config5_ring_ct_pt_slice(pf):
    """BASELINE config 5 ring (N=32768, 15 data primes): ct x pt on an 8-ciphertext batch against the oracle."""
    N, qs, B = 32768, oracle.BFV_DEFAULT[32768][:15], 8
    rng = np.random.default_rng(20250801 + 5)
    o = oracle.Oracle(N, qs)
    c = _ctx(pf, N, qs)
    ct = np.stack([rng.integers(0, q, (B, 2, N), dtype=np.uint64) for q in qs], axis=2)
    pt = np.stack([rng.integers(0, q, (B, N), dtype=np.uint64) for q in qs], axis=1)
    got = pf.to_host_u64(c.ct_pt_mul(pf.to_device_u64(ct, _dev()), pf.to_device_u64(pt, _dev())))
    assert (got == o.ct_pt_mul(ct, pt)).all()
    d = pf.to_device_u64(ct, _dev())
    c.ntt_forward_(d)
    assert (pf.to_host_u64(d) == o.ntt_forward(ct)).all()
    c.ntt_inverse_(d)
    assert (pf.to_host_u64(d) == ct).all()


@pytest.mark.parametrize("mask,L,B", [(7, 15, 3), (7, 2, 5), (0, 15, 2), (5, 4, 40)])
def test_config5_split_passes(pf, monkeypatch, mask, L, B):
    """N = 32768, lazy 64-bit family: the transforms and the fused ct x pt as passes of small workgroups (ks_split.hpp: pass A,
    body_nsB, body_nsC), every mode switched on (PF_NS_SPLIT is read when the context is created), a round shorter than the batch,
    in place and out of place, edge values -- against the oracle, bit for bit; mask 0 = the single-kernel path on the same data."""
    monkeypatch.setenv("PF_NS_SPLIT", str(mask))
    monkeypatch.setenv("PF_NS_ROUND", str(4 * L))                     # two ciphertexts per round: several rounds, the last one short
    N, qs = 32768, oracle.BFV_DEFAULT[32768][:L]
    rng = np.random.default_rng(32768 + 17 * mask + L)
    o = oracle.Oracle(N, qs)
    c = pf.RnsContext(N, qs, _dev())                                  # (not the cached context: the knobs are per context)
    ct = np.stack([rng.integers(0, q, (B, 2, N), dtype=np.uint64) for q in qs], axis=2)
    ct[0, 0, :, :4] = np.array(qs, dtype=np.uint64)[:, None] - 1      # q - 1 and 0 in the first slots
    ct[0, 1, :, :4] = 0
    pt = np.stack([rng.integers(0, q, (B, N), dtype=np.uint64) for q in qs], axis=1)
    d_ct, d_pt = pf.to_device_u64(ct, _dev()), pf.to_device_u64(pt, _dev())
    exp = o.ct_pt_mul(ct, pt)
    assert (pf.to_host_u64(c.ct_pt_mul(d_ct, d_pt)) == exp).all()                        # out of place
    assert (pf.to_host_u64(d_ct) == ct).all()                                            # ... and the input untouched
    assert (pf.to_host_u64(c.ct_pt_mul(d_ct, d_pt[:1])) == o.ct_pt_mul(ct, pt[:1])).all()   # broadcast plaintext
    d = d_ct.clone()
    assert (pf.to_host_u64(c.ct_pt_mul(d, d_pt, out=d)) == exp).all()                    # in place
    f = o.ntt_forward(ct)
    d = d_ct.clone()
    c.ntt_forward_(d)
    assert (pf.to_host_u64(d) == f).all()
    c.ntt_inverse_(d)
    assert (pf.to_host_u64(d) == ct).all()


KS_CONFIGS = [  # (N, key moduli: data primes + special prime last, batch)
    (1024, [0x7FFFFFFFE90001, 0x7FFFFFFFBF0001, 0xFFFFFFFFF70001], 3),
    (4096, oracle.BFV_DEFAULT[4096], 5),                        # 36/36-bit data + 37-bit special: exact-FP64 NTTs
    (8192, oracle.BFV_DEFAULT[8192], 18),                       # 4 data + special, more than one workspace round
    (32768, oracle.BFV_DEFAULT[32768][:3] + oracle.BFV_DEFAULT[32768][-1:], 2),
    (32768, oracle.BFV_DEFAULT[32768][3:4] + oracle.BFV_DEFAULT[32768][-1:], 3),          # one digit: the smallest two-pass key switch
    (32768, oracle.BFV_DEFAULT[32768][:5] + oracle.BFV_DEFAULT[32768][-1:], 2),           # five digits: uneven limb groups in pass C
]


@pytest.mark.parametrize("N,qs,B", KS_CONFIGS)
def test_key_switch_parity(pf, N, qs, B):
    K, D = len(qs), len(qs) - 1
    rng = np.random.default_rng(N + K)
    o = oracle.Oracle(N, qs)
    c = _ctx(pf, N, qs)
    target = np.stack([rng.integers(0, q, (B, N), dtype=np.uint64) for q in qs[:D]], axis=1)          # [B][D][N]
    target[0, :, 0] = np.array(qs[:D], dtype=np.uint64) - 1
    ksk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(D)])
    ct = np.stack([rng.integers(0, q, (B, 2, N), dtype=np.uint64) for q in qs[:D]], axis=2)            # [B][2][D][N]
    d_ct = pf.to_device_u64(ct, _dev())
    c.key_switch_(pf.to_device_u64(target, _dev()), pf.to_device_u64(ksk, _dev()), d_ct)
    assert (pf.to_host_u64(d_ct) == o.key_switch(target, ksk, ct)).all()


def test_config5_key_switch_full_ring(pf):
    """BASELINE config 5: N=32768, 15 data primes + special prime, key [15][2][16][32768] (126 MB): one ciphertext
    against the oracle, and the same switch inside a batch of 40 (a full workspace round of 32 and a partial one of 8) gives identical rows."""
    N, qs = 32768, oracle.BFV_DEFAULT[32768]
    K, D = 16, 15
    rng = np.random.default_rng(20250801 + 5)
    o = oracle.Oracle(N, qs)
    c = _ctx(pf, N, qs)
    ksk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(D)])
    target = np.stack([rng.integers(0, q, (1, N), dtype=np.uint64) for q in qs[:D]], axis=1)
    ct = np.stack([rng.integers(0, q, (1, 2, N), dtype=np.uint64) for q in qs[:D]], axis=2)
    exp = o.key_switch(target, ksk, ct)
    d_ksk = pf.to_device_u64(ksk, _dev())
    d_ct = pf.to_device_u64(ct, _dev())
    c.key_switch_(pf.to_device_u64(target, _dev()), d_ksk, d_ct)
    assert (pf.to_host_u64(d_ct) == exp).all()
    big_t = pf.to_device_u64(np.repeat(target, 40, axis=0), _dev())
    big_c = pf.to_device_u64(np.repeat(ct, 40, axis=0), _dev())
    c.key_switch_(big_t, d_ksk, big_c)
    assert (pf.to_host_u64(big_c) == np.repeat(exp, 40, axis=0)).all()


def test_config5_key_switch_batch_256(pf):
    """BASELINE config 5 at its batch size: 256 switched polynomials in one call (16 workspace rounds); rows 0, 17 and 255 are
    independent random ciphertexts checked against the oracle, the rest repeat row 0 and must equal its result."""
    N, qs = 32768, oracle.BFV_DEFAULT[32768]
    D, B = 15, 256
    rng = np.random.default_rng(20250801 + 55)
    o = oracle.Oracle(N, qs)
    c = _ctx(pf, N, qs)
    ksk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(D)])
    t3 = np.stack([rng.integers(0, q, (3, N), dtype=np.uint64) for q in qs[:D]], axis=1)
    c3 = np.stack([rng.integers(0, q, (3, 2, N), dtype=np.uint64) for q in qs[:D]], axis=2)
    exp3 = o.key_switch(t3, ksk, c3)
    rows = {0: 0, 17: 1, 255: 2}
    sel = np.array([rows.get(b, 0) for b in range(B)])
    d_t = pf.to_device_u64(t3, _dev())[torch.from_numpy(sel).to(_dev())].contiguous()
    d_c = pf.to_device_u64(c3, _dev())[torch.from_numpy(sel).to(_dev())].contiguous()
    c.key_switch_(d_t, pf.to_device_u64(ksk, _dev()), d_c)
    got = pf.to_host_u64(d_c)
    for b in (0, 17, 255, 1, 16, 128, 254):
        assert (got[b] == exp3[sel[b]]).all(), b
    assert (got[sel == 0] == exp3[0]).all()


def test_config5_ct_pt_batch_256(pf):
    """BASELINE config 5's fused ct x pt at its batch size (256 ciphertexts x 15 limbs at N=32768): rows 0, 17 and 255 are
    independent random ciphertexts / plaintexts checked against the oracle, the others repeat row 0 and must equal its result."""
    N, qs, B = 32768, oracle.BFV_DEFAULT[32768][:15], 256
    rng = np.random.default_rng(20250801 + 555)
    o = oracle.Oracle(N, qs)
    c = _ctx(pf, N, qs)
    ct3 = np.stack([rng.integers(0, q, (3, 2, N), dtype=np.uint64) for q in qs], axis=2)
    pt3 = np.stack([rng.integers(0, q, (3, N), dtype=np.uint64) for q in qs], axis=1)
    exp3 = o.ct_pt_mul(ct3, pt3)
    rows = {0: 0, 17: 1, 255: 2}
    sel = torch.from_numpy(np.array([rows.get(b, 0) for b in range(B)])).to(_dev())
    d_ct = pf.to_device_u64(ct3, _dev())[sel].contiguous()
    d_pt = pf.to_device_u64(pt3, _dev())[sel].contiguous()
    got = pf.to_host_u64(c.ct_pt_mul(d_ct, d_pt))
    for b in (0, 17, 255, 1, 128, 254):
        assert (got[b] == exp3[rows.get(b, 0)]).all(), b
    others = np.array([b for b in range(B) if b not in rows])
    assert (got[others] == exp3[0]).all()


def test_key_switch_reserved_is_graph_capturable(pf):
    """After pf_key_switch_reserve the call neither allocates nor synchronises nor reads the environment: the two-pass path at
    N=32768 (and the single-kernel path at N=8192) is captured into a hipGraph, replayed on fresh inputs, checked against the oracle."""
    for N, qs, B in ((32768, oracle.BFV_DEFAULT[32768][:2] + oracle.BFV_DEFAULT[32768][-1:], 2), (8192, oracle.BFV_DEFAULT[8192], 3)):
        K, D = len(qs), len(qs) - 1
        rng = np.random.default_rng(N + 17)
        o = oracle.Oracle(N, qs)
        c = _ctx(pf, N, qs)
        c.key_switch_reserve(B)
        ksk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(D)])
        d_ksk = pf.to_device_u64(ksk, _dev())
        d_t = torch.empty((B, D, N), dtype=torch.int64, device=_dev())
        d_c = torch.empty((B, 2, D, N), dtype=torch.int64, device=_dev())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):                       # first call is INSIDE the capture: nothing may allocate
            c.key_switch_(d_t, d_ksk, d_c)
        for _ in range(2):
            target = np.stack([rng.integers(0, q, (B, N), dtype=np.uint64) for q in qs[:D]], axis=1)
            ct = np.stack([rng.integers(0, q, (B, 2, N), dtype=np.uint64) for q in qs[:D]], axis=2)
            d_t.copy_(pf.to_device_u64(target, _dev())); d_c.copy_(pf.to_device_u64(ct, _dev()))
            graph.replay()
            torch.cuda.synchronize()
            assert (pf.to_host_u64(d_c) == o.key_switch(target, ksk, ct)).all(), N


def test_errors_are_statuses(pf):
    c = _ctx(pf, 1024, oracle.BFV_DEFAULT[1024])
    with pytest.raises(pf.PfError):
        pf.RnsContext(1024, [0x7E00003], _dev())          # not prime / not 1 mod 2N
    with pytest.raises(pf.PfError):
        pf.RnsContext(65536, [0x7FFFFFFFE90001], _dev())              # degree not built
    empty = torch.empty((0, 1024), dtype=torch.int64, device=_dev())
    c.ntt_forward_(empty)                                  # empty batch is a no-op
    with pytest.raises(ValueError):
        c.ntt_forward_(torch.zeros(1000, dtype=torch.int64, device=_dev()))


@pytest.mark.parametrize("N,qs,fanout,B", [(1024, oracle.BFV_DEFAULT[1024], 3, 7), (8192, oracle.BFV_DEFAULT[8192][:4], 4, 20),
                                           (4096, oracle.BFV_DEFAULT[4096][:2], 5, 5)])
def test_ct_pt_mul_fanout(pf, N, qs, fanout, B):
    """out[b] = ct[b // fanout] x pt[b]: the shape of the encrypted precise search (one query ciphertext against several
    packed plaintext blocks), plain and accumulating, against the oracle's per-pair product."""
    L = len(qs)
    rng = np.random.default_rng(N + fanout)
    o = oracle.Oracle(N, qs)
    c = _ctx(pf, N, qs)
    n_ct = -(-B // fanout)
    ct = np.stack([np.stack([rng.integers(0, q, (2, N), dtype=np.uint64) for q in qs], axis=1) for _ in range(n_ct)])   # [n_ct][2][L][N]
    pt = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(B)])                     # [B][L][N]
    exp = o.ct_pt_mul(ct[np.arange(B) // fanout], pt)
    d_ct, d_pt = pf.to_device_u64(ct, _dev()), pf.to_device_u64(pt, _dev())
    got = c.ct_pt_mul_fanout(d_ct, d_pt, fanout)
    assert (pf.to_host_u64(got) == exp).all()
    acc0 = np.stack([np.stack([rng.integers(0, q, (2, N), dtype=np.uint64) for q in qs], axis=1) for _ in range(B)])
    d_acc = pf.to_device_u64(acc0, _dev())
    c.ct_pt_mul_fanout(d_ct, d_pt, fanout, out=d_acc, flags=1)
    exp_acc = o.ct_pt_mul(ct[np.arange(B) // fanout], pt, 1, acc=acc0)
    assert (pf.to_host_u64(d_acc) == exp_acc).all()
    with pytest.raises(ValueError):
        c.ct_pt_mul_fanout(d_ct, d_pt, 0)


@pytest.mark.parametrize("N,qs", [(1024, oracle.BFV_DEFAULT[1024]), (8192, oracle.BFV_DEFAULT[8192][:4]), (32768, oracle.BFV_DEFAULT[32768][:3])])
def test_apply_galois(pf, N, qs):
    """X -> X^g on coefficient-form polynomials against the definition, plus the group law tau_g(tau_h(a)) = tau_{gh}(a)
    and tau_g(a) * tau_g(b) = tau_g(a * b) through the NTT product."""
    L = len(qs)
    rng = np.random.default_rng(N)
    a = np.stack([np.stack([edge_poly(rng, N, q, kind) for q in qs]) for kind in (0, 1, 3, 0)])      # [4][L][N]
    c = _ctx(pf, N, qs)
    d = pf.to_device_u64(a, _dev())
    for g in (3, 9, 2 * N - 1, 5, 1):
        got = pf.to_host_u64(c.apply_galois(d, g))
        assert (got == oracle.apply_galois(a, g, qs)).all(), g
    gh = (3 * 5) % (2 * N)
    assert (pf.to_host_u64(c.apply_galois(c.apply_galois(d, 5), 3)) == oracle.apply_galois(a, gh, qs)).all()
    o = oracle.Oracle(N, qs)
    prod = o.ntt_inverse(o.dyadic_mul(o.ntt_forward(a[0]), o.ntt_forward(a[3])))
    ta, tb = oracle.apply_galois(a[0], 3, qs), oracle.apply_galois(a[3], 3, qs)
    assert (o.ntt_inverse(o.dyadic_mul(o.ntt_forward(ta), o.ntt_forward(tb))) == oracle.apply_galois(prod, 3, qs)).all()
    with pytest.raises(pf.PfError):
        c.apply_galois(d, 4)
    with pytest.raises(pf.PfError):
        c.apply_galois(d, 3, out=d)
    # the ciphertext form (one launch for a batch, laid out for pf_key_switch): [2 ciphertexts][2][L][N]
    for g in (3, 2 * N - 1, N + 1):
        ct_out, target = (pf.to_host_u64(t) for t in c.apply_galois_ct(d.view(2, 2, L, N), g))
        exp = oracle.apply_galois(a, g, qs).reshape(2, 2, L, N)
        assert (ct_out[:, 0] == exp[:, 0]).all() and (ct_out[:, 1] == 0).all() and (target == exp[:, 1]).all(), g


@pytest.mark.parametrize("N,qs", [(1024, oracle.BFV_DEFAULT[1024]), (8192, oracle.BFV_DEFAULT[8192][:4])])
def test_mul_monomial(pf, N, qs):
    """pf_poly_mul_monomial against the definition (coefficient i -> i + k mod 2N, negated past N) and against the NTT product
    with the monomial as a polynomial."""
    rng = np.random.default_rng(N + 1)
    a = np.stack([np.stack([edge_poly(rng, N, q, kind) for q in qs]) for kind in (0, 1, 3)])       # [3][L][N]
    c = _ctx(pf, N, qs)
    d = pf.to_device_u64(a, _dev())
    o = oracle.Oracle(N, qs)
    qv = np.array(qs, dtype=np.uint64)[None, :, None]
    for k in (0, 1, N - 1, N, N + 5, 2 * N - 1, 2 * N - 64):
        idx = (np.arange(N) + k) % (2 * N)
        exp = np.zeros_like(a)
        neg = np.where(a == 0, a, qv - a)
        exp[..., idx % N] = np.where(idx >= N, neg, a)
        got = pf.to_host_u64(c.mul_monomial(d, k))
        assert (got == exp).all(), k
        mono = np.zeros((len(qs), N), dtype=np.uint64)
        for l, q in enumerate(qs):
            mono[l, k % N] = 1 if k < N else q - 1
        prod = o.ntt_inverse(o.dyadic_mul(o.ntt_forward(a[0]), o.ntt_forward(mono)))
        assert (got[0] == prod).all(), k
    # sum and shifted difference in one pass, the sum in place
    b = pf.to_device_u64(a[::-1].copy(), _dev())
    for k in (1, 2 * N - 4, N):
        exp_sum, exp_diff = pf.to_host_u64(c.add(d, b)), pf.to_host_u64(c.mul_monomial(c.sub(d, b), k))
        a2 = d.clone()
        s_, d_ = c.addsub_monomial(a2, b, k, sum_out=a2)
        assert (pf.to_host_u64(s_) == exp_sum).all() and (pf.to_host_u64(d_) == exp_diff).all(), k
    with pytest.raises(pf.PfError):
        c.addsub_monomial(d, b, 1, diff_out=d)
    with pytest.raises(pf.PfError):
        c.mul_monomial(d, 2 * N)
    with pytest.raises(pf.PfError):
        c.mul_monomial(d, 3, out=d)


@pytest.mark.parametrize("N,qs", [(1024, oracle.BFV_DEFAULT[1024]), (8192, oracle.BFV_DEFAULT[8192][:4])])
def test_ct_pt_dot(pf, N, qs):
    """pf_ct_pt_dot against ct_pt_mul(IN_NTT | OUT_NTT) + additions through the oracle: groups of `chunk` plaintexts, the
    ciphertexts reused cyclically, a ragged last group."""
    rng = np.random.default_rng(N + 7)
    L, n_ct, n_pt, chunk = len(qs), 6, 16, 3
    ct = np.stack([np.stack([rng.integers(0, q, (2, N), dtype=np.uint64) for q in qs], axis=1) for _ in range(n_ct)])   # [n_ct,2,L,N]
    ct[0, 0, 0, :5] = np.uint64(qs[0] - 1)
    pt = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_pt)])                 # [n_pt,L,N]
    pt[1, 0, :5] = np.uint64(qs[0] - 1)
    c = _ctx(pf, N, qs)
    o = oracle.Oracle(N, qs)
    got = pf.to_host_u64(c.ct_pt_dot(pf.to_device_u64(ct, _dev()), pf.to_device_u64(pt, _dev()), chunk))
    assert got.shape == (6, 2, L, N)
    for g in range(6):
        acc = np.zeros((2, L, N), dtype=np.uint64)
        for p_ in range(g * chunk, min((g + 1) * chunk, n_pt)):
            for comp in range(2):
                acc[comp] = o.addsub(acc[comp], o.dyadic_mul(ct[p_ % n_ct, comp], pt[p_]), o.ADD)
        assert (got[g] == acc).all(), g
    with pytest.raises(pf.PfError):
        c.ct_pt_dot(pf.to_device_u64(ct, _dev()), pf.to_device_u64(pt, _dev()), 4)      # 4 does not divide 6


def test_out_of_place_transforms(pf):
    N, qs = 4096, oracle.BFV_DEFAULT[4096][:2]
    rng = np.random.default_rng(9)
    o = oracle.Oracle(N, qs)
    c = _ctx(pf, N, qs)
    a = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(3)])
    d = pf.to_device_u64(a, _dev())
    f = c.ntt_forward(d)
    assert (pf.to_host_u64(d) == a).all()                              # source untouched
    assert (pf.to_host_u64(f) == o.ntt_forward(a)).all()
    assert (pf.to_host_u64(c.ntt_inverse(f)) == a).all()
