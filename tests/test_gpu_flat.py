"""GPU parity tests for the plaintext distance stages: IndexFlatL2-style search (top-k indices bit-exact,
fp32 distances within 1e-5 relative -- exact on SIFT-like integer data), Server::preciseSearch semantics
(bit-exact, any data) and the row gather."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

RTOL = 1e-5   # north_star: fp32 distances within 1e-5 relative


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device (no silent CPU fallback)")
    return "cuda:0"


@pytest.fixture(scope="module")
def pf():
    import prefhetch_amd
    return prefhetch_amd


def _sift_like(rng, n, d=128):
    return rng.integers(0, 256, (n, d)).astype(np.float32)


@pytest.mark.parametrize("nb,nq,k", [(10000, 5, 100), (10000, 5, 200), (10000, 130, 20), (256, 5, 256), (1000, 1, 1),
                                      (33000, 257, 200), (100, 3, 200), (4097, 7, 1024),
                                      (300000, 3, 200), (70000, 48, 100), (70000, 64, 50), (150000, 33, 10)])   # 32- / 64-row query tiles, streaming chunks
def test_flat_search_integer_data_exact(pf, nb, nq, k):
    rng = np.random.default_rng(nb + nq + k)
    xb, xq = _sift_like(rng, nb), _sift_like(rng, nq)
    xb[nb // 2] = xb[0]                                    # exact duplicate rows: tie -> smaller id first
    idx = pf.FlatL2(xb, _dev())
    D, I = idx.search(torch.from_numpy(xq).to(_dev()), k)
    Dr, Ir = oracle.flat_l2_search(xb, xq, k)
    assert (I.cpu().numpy() == Ir).all()
    assert (D.cpu().numpy() == Dr).all()


def test_flat_search_gaussian_tolerance(pf):
    rng = np.random.default_rng(4)
    nb, nq, k, d = 20000, 64, 100, 128
    xb, xq = rng.standard_normal((nb, d)).astype(np.float32), rng.standard_normal((nq, d)).astype(np.float32)
    idx = pf.FlatL2(xb, _dev())
    D, I = idx.search(torch.from_numpy(xq).to(_dev()), k)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    exact = ((xb[None].astype(np.float64) - xq[:, None].astype(np.float64)) ** 2).sum(-1)     # [nq][nb]
    assert (np.diff(D, axis=1) >= 0).all()
    for i in range(nq):
        di = exact[i][I[i]]
        assert np.allclose(D[i], di, rtol=RTOL, atol=0)
        kth = np.sort(exact[i])[k - 1]
        assert (di <= kth * (1 + 4 * RTOL)).all()           # tie-tolerant: every returned id is a true top-k within tolerance
        assert len(set(I[i])) == k


def test_flat_search_adversarial_order_overflow_fallback(pf):
    """Base rows ordered by DEcreasing distance to every query: each streamed row beats the running k-th distance,
    the filtered candidate lists overflow, and the exact recomputation path has to produce the answer."""
    nb, nq, k, d = 60000, 6, 200, 16
    xb = np.zeros((nb, d), np.float32)
    xb[:, 0] = np.arange(nb, 0, -1, dtype=np.float32) % 4096          # exact in fp32; plenty of ties as well
    xb[:, 1] = (np.arange(nb) // 4096).astype(np.float32)[::-1]
    xq = np.zeros((nq, d), np.float32)
    xq[:, 2] = np.arange(nq)
    idx = pf.FlatL2(xb, _dev())
    D, I = idx.search(torch.from_numpy(xq).to(_dev()), k)
    Dr, Ir = oracle.flat_l2_search(xb, xq, k)
    assert (I.cpu().numpy() == Ir).all() and (D.cpu().numpy() == Dr).all()


def test_flat_edge_cases(pf):
    """Empty base, empty query batch, out-of-range ids: defined results, no faults."""
    rng = np.random.default_rng(1)
    empty = pf.FlatL2(np.zeros((0, 128), np.float32), _dev())
    D, I = empty.search(torch.from_numpy(_sift_like(rng, 3)).to(_dev()), 5)
    assert bool(torch.isinf(D).all()) and bool((I == -1).all())
    base = _sift_like(rng, 300)
    idx = pf.FlatL2(base, _dev())
    D0, I0 = idx.search(torch.empty((0, 128), dtype=torch.float32, device=_dev()), 4)
    assert D0.shape == (0, 4) and I0.shape == (0, 4)
    xq = _sift_like(rng, 2)
    ids = np.array([[0, 299, -1, 300], [5, 5, 10**12, 7]], dtype=np.int64)
    got = idx.l2_gathered(torch.from_numpy(xq).to(_dev()), torch.from_numpy(ids).to(_dev())).cpu().numpy()
    ok = (ids >= 0) & (ids < 300)
    ref = oracle.precise_search(base, xq, np.where(ok, ids, 0))
    assert (got[ok] == ref[ok]).all() and np.isinf(got[~ok]).all()
    rows = idx.gather_rows(torch.from_numpy(ids).to(_dev())).cpu().numpy()
    assert (rows[ok] == base[ids[ok]]).all() and np.isnan(rows[~ok]).all()


def test_flat_search_odd_dimension(pf):
    rng = np.random.default_rng(6)
    xb, xq = _sift_like(rng, 3000, 30), _sift_like(rng, 9, 30)
    idx = pf.FlatL2(xb, _dev())
    D, I = idx.search(torch.from_numpy(xq).to(_dev()), 10)
    Dr, Ir = oracle.flat_l2_search(xb, xq, 10)
    assert (I.cpu().numpy() == Ir).all() and (D.cpu().numpy() == Dr).all()


def test_centroid_shortlist_matches_client(pf):
    """sort_nearest_centroids (client_lib.cpp:50-81) on the reference shapes: NQUERY=5, NLIST=256, NPROBE=20."""
    rng = np.random.default_rng(8)
    cent, xq = _sift_like(rng, 256), _sift_like(rng, 5)
    idx = pf.FlatL2(cent, _dev())
    D, I = idx.search(torch.from_numpy(xq).to(_dev()), 20)
    Dr, Ir = oracle.flat_l2_search(cent, xq, 20, mode=1)
    assert (I.cpu().numpy() == Ir).all() and (D.cpu().numpy() == Dr).all()


@pytest.mark.parametrize("gaussian", [False, True])
def test_precise_search_bit_exact(pf, gaussian):
    """Server::preciseSearch (server_lib.cpp:140-167) on the reference shapes NBASE=10000, NQUERY=5, COARSE_PROBE=200."""
    rng = np.random.default_rng(10 + gaussian)
    if gaussian:
        base, xq = rng.standard_normal((10000, 128)).astype(np.float32), rng.standard_normal((5, 128)).astype(np.float32)
    else:
        base, xq = _sift_like(rng, 10000), _sift_like(rng, 5)
    ids = rng.integers(0, 10000, (5, 200)).astype(np.int64)
    idx = pf.FlatL2(base, _dev())
    got = idx.l2_gathered(torch.from_numpy(xq).to(_dev()), torch.from_numpy(ids).to(_dev())).cpu().numpy()
    assert (got == oracle.precise_search(base, xq, ids)).all()


def test_gather_rows(pf):
    rng = np.random.default_rng(12)
    base = rng.standard_normal((10000, 128)).astype(np.float32)
    ids = rng.integers(0, 10000, (5, 100)).astype(np.int64)
    idx = pf.FlatL2(base, _dev())
    got = idx.gather_rows(torch.from_numpy(ids).to(_dev())).cpu().numpy()
    assert (got == oracle.gather_rows(base, ids)).all()


def test_config3_prefilter_full_size(pf):
    """BASELINE config 3 pre-filter: 1M x 128 base, 1024 queries, k=200.  A 16-query slice against the oracle;
    all queries through properties (sorted, unique, distances equal to recomputed exact distances)."""
    rng = np.random.default_rng(20250801 + 3)
    nb, nq, k = 1_000_000, 1024, 200
    xb = rng.integers(0, 256, (nb, 128), dtype=np.uint8).astype(np.float32)
    xq = rng.integers(0, 256, (nq, 128), dtype=np.uint8).astype(np.float32)
    idx = pf.FlatL2(xb, _dev())
    assert idx.operands8() and idx.operands16() == 2        # the path the benchmark times: int8 tiles (a silent switch-off must not stay green)
    dq = torch.from_numpy(xq).to(_dev())
    D, I = idx.search(dq, k)
    assert bool((D[:, 1:] >= D[:, :-1]).all())
    exact = idx.l2_gathered(dq, I)                          # exact on integer data, any summation order
    assert torch.equal(exact, D)
    assert all(len(set(r)) == k for r in I[:64].cpu().numpy())
    Dr, Ir = oracle.flat_l2_search(xb, xq[500:516], k)
    assert (I[500:516].cpu().numpy() == Ir).all() and (D[500:516].cpu().numpy() == Dr).all()


def test_config3_prefilter_full_size_gaussian(pf):
    """BASELINE config 3 pre-filter on N(0,1) data (SURVEY 8(d)'s second law; the fp32-operand path): 1M x 128 base, 1024
    queries, k = 200.  A 16-query slice against float64 distances: every returned distance within the north-star's 1e-5
    relative tolerance of the exact one, every returned id a true top-k member within that tolerance (tie-tolerant), ids
    unique, distances ascending; for all queries: ascending and equal to the gathered recomputation within tolerance."""
    g = torch.Generator(device=_dev()).manual_seed(20250801 + 3)
    nb, nq, k = 1_000_000, 1024, 200
    xb = torch.randn((nb, 128), generator=g, device=_dev())
    xq = torch.randn((nq, 128), generator=g, device=_dev())
    idx = pf.FlatL2(xb, _dev())
    assert not idx.exact16()                                 # Gaussian data never qualifies for 16-bit operands
    D, I = idx.search(xq, k)
    assert bool((D[:, 1:] >= D[:, :-1]).all())
    assert torch.allclose(idx.l2_gathered(xq, I), D, rtol=RTOL, atol=0)
    xb_h = xb.cpu().numpy().astype(np.float64)
    for i in range(700, 716):
        q = xq[i].cpu().numpy().astype(np.float64)
        exact = ((xb_h - q) ** 2).sum(-1)
        ids = I[i].cpu().numpy()
        assert len(set(ids)) == k
        assert np.allclose(D[i].cpu().numpy(), exact[ids], rtol=RTOL, atol=0)
        kth = np.partition(exact, k - 1)[k - 1]
        assert (exact[ids] <= kth * (1 + 4 * RTOL)).all()


@pytest.mark.parametrize("d,M,nlist,n,nq,nprobe", [(128, 32, 256, 10000, 5, 20), (64, 8, 16, 3000, 9, 3), (24, 6, 7, 500, 4, 7)])
def test_ivfpq_search_lists_bit_exact(pf, d, M, nlist, n, nq, nprobe):
    """IndexIVFPQ::search_encrypted semantics (Server::coarseSearch, server_lib.cpp:111-138): ADC over the GIVEN lists,
    all stored vectors, unsorted, bit-exact against the oracle on the same index content (reference shapes first)."""
    rng = np.random.default_rng(d + M + n)
    cent = rng.standard_normal((nlist, d)).astype(np.float32) * 30
    books = rng.standard_normal((M, 256, d // M)).astype(np.float32) * 5
    lists = rng.integers(0, nlist, n).astype(np.int64)
    lists[lists == 1] = 0                                           # list 1 stays empty
    codes = rng.integers(0, 256, (n, M)).astype(np.uint8)
    ids = rng.permutation(n).astype(np.int64)
    idx = pf.IvfPq(cent, books, _dev())
    idx.add_encoded(lists[: n // 2], codes[: n // 2], ids[: n // 2])           # two appends: insertion order per list
    idx.add_encoded(lists[n // 2:], codes[n // 2:], ids[n // 2:])
    xq = rng.standard_normal((nq, d)).astype(np.float32) * 30
    probe = np.stack([rng.permutation(nlist)[:nprobe] for _ in range(nq)]).astype(np.int64)
    probe[0, 0] = 1                                                  # an empty list
    if nprobe > 2:
        probe[1, 1] = -1                                             # faiss convention for "no list"
    D, I, sizes = idx.search_lists(torch.from_numpy(xq).to(_dev()), probe)
    order = np.argsort(lists, kind="stable")                         # flat index content: list-contiguous, insertion order
    off = np.concatenate([[0], np.cumsum(np.bincount(lists, minlength=nlist))]).astype(np.uint64)
    Dr, Ir, sr = oracle.ivfpq_search_lists(xq, probe, cent, books, codes[order], ids[order], off)
    assert (sizes == sr).all() and int(sizes.sum()) == Dr.size
    assert (I.cpu().numpy() == Ir).all()
    assert (D.cpu().numpy().view(np.uint32) == Dr.view(np.uint32)).all()


@pytest.mark.parametrize("N,qs,rows", [(1024, oracle.BFV_DEFAULT[1024], 8), (8192, oracle.BFV_DEFAULT[8192][:4], 64),
                                       (8192, oracle.BFV_DEFAULT[8192][:4], 8), (4096, oracle.BFV_DEFAULT[4096][:2], 32)])
def test_pack_rows_matches_the_definition(N, qs, rows):
    """pf_pack_rows (plaintext polynomials of the encrypted precise search) against oracle.pack_rows, bit-exact,
    including ids outside the base (zero rows), and fractional values (rounded to nearest)."""
    import prefhetch_amd as pf
    dev = _dev()
    rng = np.random.default_rng(N + rows)
    nb, d = 3000, 128
    base = rng.integers(0, 256, (nb, d)).astype(np.float32)
    base[7] += np.float32(0.25)                      # rint
    base[8] = -base[8]                               # negative entries
    ids = rng.integers(0, nb, (5, rows)).astype(np.int64)
    ids[0, 0], ids[1, rows - 1], ids[2, 1], ids[3, 0] = 7, 8, -1, nb
    flat = pf.FlatL2(base, dev)
    ctx = pf.RnsContext(N, qs, dev)
    got = pf.to_host_u64(ctx.pack_rows(flat, torch.from_numpy(ids).to(dev)))
    exp = oracle.pack_rows(base, ids, N, qs)
    assert got.shape == exp.shape and (got == exp).all()
    with pytest.raises(pf.PfError):
        ctx.pack_rows(flat, torch.zeros((1, N // d + 1), dtype=torch.int64, device=dev))


@pytest.mark.parametrize("N,qs,rows,force,d", [(8192, oracle.BFV_DEFAULT[8192][:4], 64, 0, 128), (1024, oracle.BFV_DEFAULT[1024], 8, 0, 128),
                                               (4096, oracle.BFV_DEFAULT[4096][:2], 17, 1, 128), (8192, oracle.BFV_DEFAULT[8192][:4], 64, 2, 128),
                                               # row lengths that divide neither the workgroup size nor N: the wrap-around of row 0
                                               # covers coefficients N-d+1 .. N-1 whatever N mod d is
                                               (8192, oracle.BFV_DEFAULT[8192][:4], 81, 0, 100), (8192, oracle.BFV_DEFAULT[8192][:4], 85, 0, 96),
                                               (1024, oracle.BFV_DEFAULT[1024], 10, 0, 100), (4096, oracle.BFV_DEFAULT[4096][:2], 40, 1, 100)])
def test_pack_rows_ntt_equals_pack_then_transform(N, qs, rows, force, d):
    """pf_pack_rows_ntt (packing fused into the forward transform) against pack_rows -> ntt_forward and against the
    oracle, bit for bit, on every arithmetic back-end."""
    import prefhetch_amd as pf
    dev = _dev()
    rng = np.random.default_rng(N * 3 + rows)
    nb, n_polys = 2000, 7
    base = rng.integers(0, 256, (nb, d)).astype(np.float32)
    base[5] = -base[5]
    ids = rng.integers(0, nb, (n_polys, rows)).astype(np.int64)
    ids[0, 0], ids[1, rows - 1], ids[2, 0] = 5, -1, nb + 3
    flat = pf.FlatL2(base, dev)
    ctx = pf.RnsContext(N, qs, dev)
    if force:
        ctx.force_u64(force)
    d_ids = torch.from_numpy(ids).to(dev)
    fused = pf.to_host_u64(ctx.pack_rows(flat, d_ids, ntt=True))
    two = pf.to_host_u64(ctx.ntt_forward_(ctx.pack_rows(flat, d_ids)))
    assert (fused == two).all()
    assert (fused == oracle.Oracle(N, qs).ntt_forward(oracle.pack_rows(base, ids, N, qs))).all()


@pytest.mark.parametrize("N,qs,rows,force,d,B,fan", [(8192, oracle.BFV_DEFAULT[8192][:4], 64, 0, 128, 12, 4), (8192, oracle.BFV_DEFAULT[8192][:4], 64, 2, 128, 7, 3),
                                                     (8192, oracle.BFV_DEFAULT[8192][:4], 64, 1, 128, 5, 1), (1024, oracle.BFV_DEFAULT[1024], 8, 0, 128, 9, 2),
                                                     (4096, oracle.BFV_DEFAULT[4096][:2], 40, 1, 100, 10, 4), (16384, [0x7FFFFFD8001, 0x7FFFFFC8001], 128, 0, 128, 5, 2),
                                                     (16384, [0x7FFFFFFFE90001, 0x7FFFFFD8001], 128, 0, 128, 3, 2),
                                                     (8192, oracle.BFV_DEFAULT[8192][:4], 81, 0, 100, 6, 6)])
def test_ct_rows_mul_equals_pack_transform_multiply(N, qs, rows, force, d, B, fan):
    """pf_ct_rows_mul (rows -> plaintext -> NTT -> x both ciphertext components -> inverse NTT in one workgroup) against
    pack_rows_ntt + ct_pt_mul_fanout(IN_NTT) and against the oracle, bit for bit, on every arithmetic back-end; B not a
    multiple of the fan-out, ids out of range, a negated row."""
    import prefhetch_amd as pf
    dev = _dev()
    rng = np.random.default_rng(N + 31 * B + fan)
    nb = 3000
    base = rng.integers(0, 256, (nb, d)).astype(np.float32)
    base[5] = -base[5]
    ids = rng.integers(0, nb, (B, rows)).astype(np.int64)
    ids[0, 0], ids[1, rows - 1], ids[2, 0] = 5, -1, nb + 3
    flat = pf.FlatL2(base, dev)
    ctx = pf.RnsContext(N, qs, dev)
    if force:
        ctx.force_u64(force)
    L, n_ct = len(qs), -(-B // fan)
    ct = np.stack([rng.integers(0, q, (n_ct, 2, N), dtype=np.uint64) for q in qs], axis=2)           # [n_ct,2,L,N]
    d_ids, d_ct = torch.from_numpy(ids).to(dev), pf.to_device_u64(ct, dev)
    ctn = ctx.ntt_forward(d_ct)
    one = pf.to_host_u64(ctx.ct_rows_mul(ctn, flat, d_ids, fan))
    two = pf.to_host_u64(ctx.ct_pt_mul_fanout(ctn, ctx.pack_rows(flat, d_ids, ntt=True), fan, flags=2))
    assert one.shape == (B, 2, L, N) and (one == two).all()
    o = oracle.Oracle(N, qs)
    pt_ntt = o.ntt_forward(oracle.pack_rows(base, ids, N, qs))
    exp = o.ct_pt_mul(np.ascontiguousarray(ct[np.arange(B) // fan]), pt_ntt)
    assert (one == exp.reshape(one.shape)).all()
    with pytest.raises(pf.PfError):
        ctx.ct_rows_mul(ctn, flat, torch.zeros((B, N // d + 1), dtype=torch.int64, device=dev), fan)


def test_step_captured_in_a_hip_graph(pf, capsys):
    """include/prefhetch_hip.h promises that the polynomial calls and pf_flat_search (after pf_flat_reserve) neither
    allocate nor synchronise: the whole step is captured into a hipGraph, replayed on fresh inputs and compared with the
    oracle; the replay time of a single-query search (a chain of ~13 dependent launches) is printed beside the eager one."""
    import time
    dev = _dev()
    rng = np.random.default_rng(77)
    N, qs = 4096, oracle.BFV_DEFAULT[4096][:2]
    nb, nq, k = 60000, 3, 50
    xb = _sift_like(rng, nb)
    flat = pf.FlatL2(xb, dev)
    flat.reserve(nq, k)
    ctx = pf.RnsContext(N, qs, dev)
    o = oracle.Oracle(N, qs)
    xq = torch.empty((nq, 128), dtype=torch.float32, device=dev)
    ct = torch.empty((2, 2, len(qs), N), dtype=torch.int64, device=dev)
    pt = torch.empty((2, len(qs), N), dtype=torch.int64, device=dev)
    out = torch.empty_like(ct)

    def fresh():
        hq = _sift_like(rng, nq)
        hct = np.stack([np.stack([rng.integers(0, q, (2, N), dtype=np.uint64) for q in qs], axis=1) for _ in range(2)])
        hpt = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)])
        xq.copy_(torch.from_numpy(hq)); ct.copy_(pf.to_device_u64(hct, dev)); pt.copy_(pf.to_device_u64(hpt, dev))
        return hq, hct, hpt

    fresh()
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):                       # warm-up outside capture (workspace, lazy module loads)
        flat.search(xq, k)
        ctx.ct_pt_mul(ct, pt, out=out)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        D, I = flat.search(xq, k)
        ctx.ct_pt_mul(ct, pt, out=out)
    for _ in range(3):
        hq, hct, hpt = fresh()
        graph.replay()
        torch.cuda.synchronize()
        Dr, Ir = oracle.flat_l2_search(xb, hq, k)
        assert (I.cpu().numpy() == Ir).all() and (D.cpu().numpy() == Dr).all()
        assert (pf.to_host_u64(out) == o.ct_pt_mul(hct, hpt)).all()
    # a batch (bf16 tiles fed by LDS-DMA, merges by a wave per query) captured the same way
    nqb = 300
    flat.reserve(nqb, k)
    xqb = torch.empty((nqb, 128), dtype=torch.float32, device=dev)
    xqb.copy_(torch.from_numpy(_sift_like(rng, nqb)))
    with torch.cuda.stream(side):
        flat.search(xqb, k)
    torch.cuda.synchronize()
    gb = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gb):
        Db, Ib = flat.search(xqb, k)
    for _ in range(2):
        hq = _sift_like(rng, nqb)
        xqb.copy_(torch.from_numpy(hq))
        gb.replay()
        torch.cuda.synchronize()
        Dr, Ir = oracle.flat_l2_search(xb, hq[:6], k)
        assert (Ib[:6].cpu().numpy() == Ir).all() and (Db[:6].cpu().numpy() == Dr).all()
        De, Ie = flat.search(xqb, k)
        assert (Ib == Ie).all() and (Db == De).all()
    # launch-bound case: one query against 1M rows, eager against replay
    big = pf.FlatL2(_sift_like(rng, 1_000_000), dev)
    big.reserve(1, 200)
    q1 = torch.from_numpy(_sift_like(rng, 1)).to(dev)
    with torch.cuda.stream(side):
        big.search(q1, 200)
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        D1, I1 = big.search(q1, 200)
    De, Ie = big.search(q1, 200)
    g1.replay()
    torch.cuda.synchronize()
    assert (I1 == Ie).all() and (D1 == De).all()

    def ms(fn, reps=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / reps

    eager, replay = ms(lambda: big.search(q1, 200)), ms(g1.replay)
    with capsys.disabled():
        print(f"\n[hipGraph] single-query search over 1M x 128: eager {eager:.3f} ms, graph replay {replay:.3f} ms")


def _search_both(pf, xb, xq, k):
    """(D, I) with the 16-bit operand path on and off, plus whether the index took the path"""
    dev = _dev()
    f = pf.FlatL2(xb, dev)
    q = torch.from_numpy(xq).to(dev)
    active = f.exact16()
    D1, I1 = f.search(q, k)
    f.exact16(0)
    assert not f.exact16()
    D0, I0 = f.search(q, k)
    return active, (D1.cpu().numpy(), I1.cpu().numpy()), (D0.cpu().numpy(), I0.cpu().numpy())


@pytest.mark.parametrize("d,nq,k", [(128, 300, 200), (64, 129, 10), (128, 1024, 100), (64, 65, 64),
                                    (16, 200, 33), (32, 257, 100), (48, 130, 200), (80, 300, 50), (96, 513, 100), (112, 129, 17),   # every multiple of 16 up to 128
                                    (144, 130, 100), (160, 300, 33), (176, 129, 200), (192, 257, 64), (208, 65, 10), (224, 200, 100), (240, 140, 17),
                                    (256, 300, 200)])                                                                                # ... and up to 256: 128 x 64 tiles
def test_exact16_path_is_bit_identical_on_integer_data(d, nq, k):
    """exactly-representable data (integers, |v| <= 256): the bf16-operand tiles must return the fp32 loop's (D, I) and the
    oracle's, bit for bit -- including the extreme values +-256 and heavy ties"""
    import prefhetch_amd as pf
    rng = np.random.default_rng(d + nq)
    xb = rng.integers(-256, 257, (40000, d)).astype(np.float32)
    xq = rng.integers(-256, 257, (nq, d)).astype(np.float32)
    xb[:50] = 256.0; xq[0] = -256.0; xq[1] = 256.0                 # largest products, largest sums
    xb[100:400] = xb[100]                                            # a plateau of ties
    active, got16, got32 = _search_both(pf, xb, xq, k)
    assert active
    assert (got16[1] == got32[1]).all() and (got16[0].view(np.uint32) == got32[0].view(np.uint32)).all()
    # against the oracle on the 8-bit range.  d <= 128: |x|^2 + |y|^2 stays below 2^24, so the fp32 formula
    # |x|^2 + |y|^2 - 2 x.y is exact and distances match bit for bit; wider rows round that sum once (both paths alike)
    xb, xq = np.abs(xb).clip(0, 255), np.abs(xq).clip(0, 255)
    active, got16, got32 = _search_both(pf, xb, xq, k)
    Dr, Ir = oracle.flat_l2_search(xb, xq, k)
    assert active and (got16[0].view(np.uint32) == got32[0].view(np.uint32)).all() and (got16[1] == got32[1]).all()
    if d <= 128:
        assert (got16[1] == Ir).all() and (got16[0] == Dr).all()
    else:
        assert np.allclose(got16[0], Dr, rtol=RTOL, atol=0) and (np.sort(got16[1], axis=1) == np.sort(Ir, axis=1)).mean() > 0.99


@pytest.mark.parametrize("d,nb,nq,k", [(128, 60000, 300, 200), (64, 40000, 129, 10), (32, 30000, 257, 100), (96, 50000, 513, 64), (128, 9000, 1024, 1024),
                                       (128, 200000, 70, 1)])
def test_int8_tiles_are_bit_identical_on_8bit_data(d, nb, nq, k):
    """8-bit data (integers in [0, 255], SIFT's range): the int8 matrix instruction with integer thresholds against the bf16 tiles on the same
    data, the fp32-operand loop and the oracle -- bit for bit, with the extreme values, plateaus of ties and duplicates of queries in the base"""
    import prefhetch_amd as pf
    rng = np.random.default_rng(d * 7 + nq)
    xb = rng.integers(0, 256, (nb, d)).astype(np.float32)
    xq = rng.integers(0, 256, (nq, d)).astype(np.float32)
    xb[:40] = 255.0; xb[40:80] = 0.0; xq[0] = 0.0; xq[1] = 255.0        # largest products and sums, both signs of (value - 128)
    xb[100:400] = xb[100]                                                 # a plateau of ties
    xb[9000 - 50:9000 - 40] = xq[2]                                       # exact hits (distance 0) behind the bootstrap chunk
    dev = _dev()
    f = pf.FlatL2(xb, dev)
    q = torch.from_numpy(xq).to(dev)
    assert f.operands8() and f.operands16() == 2
    D8, I8 = f.search(q, k)
    assert f.operands8(0) is False and f.operands16() == 2               # bf16 tiles on the same data
    D16, I16 = f.search(q, k)
    f.operands16(0)
    D32, I32 = f.search(q, k)
    for Dx, Ix in ((D16, I16), (D32, I32)):
        assert (I8 == Ix).all() and (D8.view(torch.int32) == Dx.view(torch.int32)).all()
    Dr, Ir = oracle.flat_l2_search(xb, xq[:3], k)
    assert (I8[:3].cpu().numpy() == Ir).all() and (D8[:3].cpu().numpy() == Dr).all()
    if k >= 10 and nb > 9000:
        assert (D8[2, :10].cpu().numpy() == 0).all()


def test_int8_tiles_only_where_both_sides_are_8bit():
    """the int8 image exists only for a base of integers in [0, 255] with d a multiple of 32 up to 128, and a query tile with any value outside
    that range (negative, 256, a fraction) takes the bf16 tiles inside the same launch -- results identical either way"""
    import prefhetch_amd as pf
    dev = _dev()
    rng = np.random.default_rng(99)
    base = rng.integers(0, 256, (30000, 128)).astype(np.float32)
    assert pf.FlatL2(base, dev).operands8()
    for bad in (-1.0, 256.0, 0.5):
        xb = base.copy(); xb[777, 3] = bad
        assert not pf.FlatL2(xb, dev).operands8()
    for d in (16, 48, 80, 144, 256):                                      # not whole 32-deep k-steps, or beyond 128
        assert not pf.FlatL2(rng.integers(0, 256, (1000, d)).astype(np.float32), dev).operands8()
    f = pf.FlatL2(base, dev)
    xq = rng.integers(0, 256, (640, 128)).astype(np.float32)             # five query tiles: 0 and 3 stay 8-bit
    xq[130, 5] = -3.0; xq[300, 77] = 256.0; xq[600, 0] = 0.25            # tiles 1, 2 (exact, not 8-bit) and 4 (inexact: filter + fp32 chain)
    q = torch.from_numpy(xq).to(dev)
    D8, I8 = f.search(q, 100)
    f.operands8(0)
    D16, I16 = f.search(q, 100)
    f.operands16(0)
    D32, I32 = f.search(q, 100)
    for Dx, Ix in ((D16, I16), (D32, I32)):
        assert (I8 == Ix).all() and (D8.view(torch.int32) == Dx.view(torch.int32)).all()
    rows = [0, 130, 300, 500]
    Dr, Ir = oracle.flat_l2_search(base, xq[rows], 100)
    assert (I8[rows].cpu().numpy() == Ir).all() and (D8[rows].cpu().numpy() == Dr).all()


@pytest.mark.parametrize("nq", [200, 300])                            # the LDS-tiled int8 walk (batches up to 256 queries) and the streamed one
def test_int8_tiles_dense_survivors_and_overflow(nq):
    """base rows ordered by DEcreasing distance to every query (every streamed row passes the integer threshold: lists overflow into the exact rescan,
    flushes in mid-walk), and thresholds at both ends of the int32 range of the row half"""
    import prefhetch_amd as pf
    nb, k, d = 40000, 100, 128
    m = (np.arange(nb)[::-1] * (d + 1) // nb)
    xb = (np.arange(d)[None, :] < m[:, None]).astype(np.float32) * 255.0
    rng = np.random.default_rng(5)
    xq = rng.integers(0, 2, (nq, d)).astype(np.float32)
    dev = _dev()
    f = pf.FlatL2(xb, dev)
    assert f.operands8()
    q = torch.from_numpy(xq).to(dev)
    D8, I8 = f.search(q, k)
    f.operands16(0)
    D32, I32 = f.search(q, k)
    assert (I8 == I32).all() and (D8.view(torch.int32) == D32.view(torch.int32)).all()
    Dr, Ir = oracle.flat_l2_search(xb, xq[:4], k)
    assert (I8[:4].cpu().numpy() == Ir).all() and (D8[:4].cpu().numpy() == Dr).all()


@pytest.mark.parametrize("d", [128, 256])
def test_exact16_dense_survivors_and_far_thresholds(d):
    """bf16 tiles under stress: base rows ordered by DEcreasing distance (every streamed row passes the filter: the verdict words are
    worked off in several rounds per flush, the candidate lists overflow into the exact rescan), and queries whose k-th distance lies
    far above their own norm (threshold term R below -2^22: the conservative-margin branch of the ninth k-step)"""
    import prefhetch_amd as pf
    nb, nq, k = 40000, 130, 100
    m = (np.arange(nb)[::-1] * (d + 1) // nb)                       # row i has m(i) entries of 2: its distance to a small query falls with i
    xb = (np.arange(d)[None, :] < m[:, None]).astype(np.float32) * 2.0
    rng = np.random.default_rng(11)
    xq = rng.integers(0, 2, (nq, d)).astype(np.float32)
    active, got16, got32 = _search_both(pf, xb, xq, k)
    assert active
    assert (got16[1] == got32[1]).all() and (got16[0].view(np.uint32) == got32[0].view(np.uint32)).all()
    Dr, Ir = oracle.flat_l2_search(xb, xq[:8], k)
    assert (got16[1][:8] == Ir).all() and (got16[0][:8] == Dr).all()
    # far thresholds: the base sits at +-256 in every coordinate, some queries at the origin / at the opposite corner
    xb = (rng.integers(0, 2, (20000, d)) * 512 - 256).astype(np.float32)
    xq = np.zeros((nq, d), np.float32)
    xq[1::3] = -256.0
    xq[2::3] = rng.integers(-256, 257, (len(xq[2::3]), d)).astype(np.float32)
    active, got16, got32 = _search_both(pf, xb, xq, k)
    assert active
    assert (got16[1] == got32[1]).all() and (got16[0].view(np.uint32) == got32[0].view(np.uint32)).all()


@pytest.mark.parametrize("d", [128, 256])
@pytest.mark.parametrize("law", ["gaussian", "mixed_norms", "integers_with_one_fraction"])
def test_bf16_filter_over_inexact_operands_is_bit_identical(law, d):
    """operands that are NOT exactly representable: the bf16 tiles run as a conservative filter (thresholds lowered by the bound
    on the operands' rounding) and every survivor is re-evaluated by the fp32 chain -- (D, I) must equal the fp32-operand
    loop's bit for bit, whatever the norms look like"""
    import prefhetch_amd as pf
    rng = np.random.default_rng(17)
    nb, nq, k = 60000, 200, 100
    if law == "gaussian":
        xb, xq = rng.standard_normal((nb, d)), rng.standard_normal((nq, d))
    elif law == "mixed_norms":                                       # rows of very different length: the margin is priced on the largest
        xb = rng.standard_normal((nb, d)) * rng.choice([0.01, 1.0, 30.0], (nb, 1))
        xq = rng.standard_normal((nq, d)) * rng.choice([0.1, 1.0, 10.0], (nq, 1))
    else:
        top = 256 if d <= 128 else 128                               # (|x|^2 + |y|^2 below 2^24)
        xb, xq = rng.integers(0, top, (nb, d)).astype(np.float64), rng.integers(0, top, (nq, d)).astype(np.float64)
        xb[777, 5] = 0.5                                             # one inexact value: the whole base counts as inexact
    xb, xq = xb.astype(np.float32), xq.astype(np.float32)
    f = pf.FlatL2(xb, _dev())
    assert f.operands16() == 1 and not f.exact16()
    q = torch.from_numpy(xq).to(_dev())
    D1, I1 = f.search(q, k)
    assert f.operands16(0) == 0
    D0, I0 = f.search(q, k)
    assert (I1 == I0).all() and (D1.view(torch.int32) == D0.view(torch.int32)).all()
    Dr, Ir = oracle.flat_l2_search(xb, xq[:4], k)                     # double accumulation, rounded once
    if law == "integers_with_one_fraction":                          # every distance is a multiple of 1/4 below 2^24: exact either way
        assert (I1[:4].cpu().numpy() == Ir).all() and (D1[:4].cpu().numpy() == Dr).all()
    else:
        assert np.allclose(D1[:4].cpu().numpy(), Dr, rtol=RTOL, atol=0)


@pytest.mark.parametrize("nb,nq,k,d,law", [(100, 200, 50, 128, "int"), (8192, 130, 20, 128, "int"), (8193, 130, 20, 128, "int"), (20000, 65, 1024, 128, "int"),
                                            (50000, 300, 1024, 64, "int"), (30000, 129, 7, 64, "gauss"), (9000, 70, 200, 128, "gauss"),
                                            (70000, 257, 300, 128, "mixed"), (60000, 300, 100, 96, "gauss"), (60000, 200, 40, 48, "mixed"),
                                            (40000, 129, 64, 16, "gauss"), (50000, 260, 200, 112, "int"), (50000, 140, 10, 80, "gauss"),
                                            (50000, 260, 200, 256, "int"), (60000, 300, 100, 256, "gauss"), (40000, 130, 1024, 192, "mixed"), (100, 200, 50, 144, "int"),
                                            (8193, 129, 20, 240, "gauss"), (70000, 140, 40, 160, "mixed")])
def test_bf16_tiles_edge_shapes(nb, nq, k, d, law):
    """shapes around the seams of the batch path: a base smaller than one tile, exactly / one past the bootstrap chunk, the smallest
    batch, k = 1024 (merges of 2048 keys by one wave), d = 64, a single filtered tile -- bf16 tiles vs fp32 operands, bit for bit"""
    import prefhetch_amd as pf
    rng = np.random.default_rng(nb + nq)
    if law == "int":
        top = 256 if d <= 128 else 128                               # |x|^2 + |y|^2 below 2^24: the fp32 formula is exact and the oracle's distance is the same number
        xb, xq = rng.integers(0, top, (nb, d)), rng.integers(0, top, (nq, d))
    elif law == "gauss":
        xb, xq = rng.standard_normal((nb, d)), rng.standard_normal((nq, d))
    else:
        xb, xq = rng.standard_normal((nb, d)) * rng.choice([0.01, 1, 50], (nb, 1)), rng.standard_normal((nq, d))
    xb, xq = xb.astype(np.float32), xq.astype(np.float32)
    f = pf.FlatL2(xb, _dev())
    q = torch.from_numpy(xq).to(_dev())
    assert f.operands16() == (2 if law == "int" else 1)
    D1, I1 = f.search(q, k)
    f.operands16(0)
    D0, I0 = f.search(q, k)
    assert (I1 == I0).all() and (D1.view(torch.int32) == D0.view(torch.int32)).all()
    if law == "int":
        Dr, Ir = oracle.flat_l2_search(xb, xq[:3], k)
        assert (I1[:3].cpu().numpy() == Ir).all() and (D1[:3].cpu().numpy() == Dr).all()


@pytest.mark.parametrize("nq", [130, 300])                            # (8-bit data: the LDS-tiled int8 walk up to 256 queries, the streamed one above)
@pytest.mark.parametrize("d", [30, 100, 130, 200, 250])
@pytest.mark.parametrize("law", ["u8", "int", "gauss", "mixed"])
def test_padded_row_lengths_take_the_tile_path(d, law, nq):
    """[r4] row lengths that are not whole k-steps of the matrix instructions (any d up to 256): the operand images are padded with zeros and the
    tile paths run -- int8 tiles on 8-bit data (d <= 128), bf16 tiles exact on integers / as a filter otherwise -- bit-identical to the
    fp32-operand tiles and, on integer data, to the oracle.  Batches large enough for filtered chunks behind the bootstrap."""
    import prefhetch_amd as pf
    rng = np.random.default_rng(1000 * d + len(law))
    nb, k = 21000, 50
    if law == "u8":
        xb, xq = rng.integers(0, 256, (nb, d)), rng.integers(0, 256, (nq, d))
    elif law == "int":
        top = 128 if d > 128 else 256
        xb, xq = rng.integers(-top, top + 1, (nb, d)), rng.integers(-top, top + 1, (nq, d))
    elif law == "gauss":
        xb, xq = rng.standard_normal((nb, d)), rng.standard_normal((nq, d))
    else:                                                            # an 8-bit base, queries that are not (a fraction somewhere): bf16 filter over an exact base
        xb, xq = rng.integers(0, 256, (nb, d)), rng.integers(0, 256, (nq, d)) + (rng.random((nq, d)) < 0.01) * 0.5
    xb, xq = xb.astype(np.float32), xq.astype(np.float32)
    f = pf.FlatL2(xb, _dev())
    q = torch.from_numpy(xq).to(_dev())
    assert f.operands16() == (1 if law == "gauss" else 2)            # an image exists at this row length
    assert f.operands8() == (law in ("u8", "mixed") and d <= 128)
    D1, I1 = f.search(q, k)
    f.operands16(0)
    D0, I0 = f.search(q, k)                                          # fp32 operands
    assert (I1 == I0).all() and (D1.view(torch.int32) == D0.view(torch.int32)).all()
    if f.operands8(-1) is not None and law == "u8" and d <= 128:
        f.operands16(1); f.operands8(0)
        D2, I2 = f.search(q, k)                                      # the same data through the bf16 tiles
        assert (I2 == I0).all() and (D2.view(torch.int32) == D0.view(torch.int32)).all()
    if law in ("u8", "int"):
        Dr, Ir = oracle.flat_l2_search(xb, xq[:4], k)
        assert (I1[:4].cpu().numpy() == Ir).all() and (D1[:4].cpu().numpy() == Dr).all()


@pytest.mark.parametrize("d", [257, 260, 384, 500, 1024])
@pytest.mark.parametrize("law", ["u8", "gauss", "mixed_norms"])
def test_rows_longer_than_256_take_the_slab_tiles(d, law):
    """[r4] rows longer than 256 values: the filtered chunks run bf16 tiles with both operands staged in k-slabs (a conservative filter) and every
    candidate's distance comes from the fp32 chain afterwards -- (D, I) bit-identical to the fp32-operand tiles and, on integer data, to the oracle.
    Batches above 64 queries, a base long enough for filtered chunks behind the bootstrap, a ragged last query tile and column tile."""
    import prefhetch_amd as pf
    rng = np.random.default_rng(7 * d + len(law))
    nb, nq, k = 20011, 150, 40
    if law == "u8":
        xb, xq = rng.integers(0, 256, (nb, d)), rng.integers(0, 256, (nq, d))
    elif law == "gauss":
        xb, xq = rng.standard_normal((nb, d)), rng.standard_normal((nq, d))
    else:                                                            # rows of very different norms, queries close to some of them
        xb = rng.standard_normal((nb, d)) * rng.choice([0.01, 1.0, 30.0], (nb, 1))
        xq = xb[rng.integers(8192, nb, nq)] + 0.05 * rng.standard_normal((nq, d))
    xb, xq = xb.astype(np.float32), xq.astype(np.float32)
    f = pf.FlatL2(xb, _dev())
    q = torch.from_numpy(xq).to(_dev())
    assert f.operands16() == 1                                       # the slab image exists; no exactness claim at these lengths
    D1, I1 = f.search(q, k)
    f.operands16(0)
    D0, I0 = f.search(q, k)                                          # fp32 operands
    assert (I1 == I0).all() and (D1.view(torch.int32) == D0.view(torch.int32)).all()
    if law == "u8":
        Dr, Ir = oracle.flat_l2_search(xb, xq[:3], k)
        if d * 255 * 255 < 2 ** 24:                                  # every partial sum an integer fp32 holds: the decomposition is exact
            assert (I1[:3].cpu().numpy() == Ir).all() and (D1[:3].cpu().numpy() == Dr).all()
        else:                                                        # beyond, |x|^2 + |y|^2 - 2 x.y rounds where the oracle's double sum does not
            # (the fp32 chain of 1024 products of magnitude 2^16 drifts by 3e-5 of the distance: the decomposition's rounding, the same with fp32 operands)
            np.testing.assert_allclose(D1[:3].cpu().numpy(), Dr, rtol=RTOL if d <= 512 else 1e-4)


def test_rows_longer_than_256_edge_shapes_and_graph_capture():
    """[r4] the slab-tile path at its edges: a base shorter than the bootstrap chunk and one of a single row (no filtered chunk ever runs), batches
    of 1 / 64 / 65 queries (the path starts at 65), k = 1 and k = 1024 (many small chunks), a row length that is not a multiple of 4 (the fix-up's
    scalar chain); and, after pf_flat_reserve, a search captured into a hipGraph and replayed on fresh queries -- every result equal to the
    fp32-operand tiles' bit for bit."""
    import prefhetch_amd as pf
    dev = _dev()
    rng = np.random.default_rng(41)
    for d, nb, nq, k in ((300, 1, 70, 1), (300, 5000, 70, 7), (515, 20000, 1, 10), (515, 20000, 64, 10), (515, 20000, 65, 10), (384, 30000, 130, 1), (384, 30000, 130, 1024)):
        xb = rng.standard_normal((nb, d)).astype(np.float32)
        xq = rng.standard_normal((nq, d)).astype(np.float32)
        f = pf.FlatL2(xb, dev)
        q = torch.from_numpy(xq).to(dev)
        D1, I1 = f.search(q, k)
        f.operands16(0)
        D0, I0 = f.search(q, k)
        assert (I1 == I0).all() and (D1.view(torch.int32) == D0.view(torch.int32)).all(), (d, nb, nq, k)
    d, nb, nq, k = 512, 40000, 200, 20
    xb = rng.standard_normal((nb, d)).astype(np.float32)
    f = pf.FlatL2(xb, dev)
    f.reserve(nq, k)
    xq = torch.empty((nq, d), dtype=torch.float32, device=dev)
    xq.copy_(torch.from_numpy(rng.standard_normal((nq, d)).astype(np.float32)))
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        f.search(xq, k)                                               # warm-up outside capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        D, I = f.search(xq, k)
    for _ in range(2):
        hq = rng.standard_normal((nq, d)).astype(np.float32)
        xq.copy_(torch.from_numpy(hq))
        g.replay()
        torch.cuda.synchronize()
        Dg, Ig = D.clone(), I.clone()
        f.operands16(0)
        D0, I0 = f.search(torch.from_numpy(hq).to(dev), k)
        f.operands16(1)
        assert (Ig == I0).all() and (Dg.view(torch.int32) == D0.view(torch.int32)).all()


def test_slab_tiles_worst_case_rounding_and_nonfinite_queries():
    """d = 512: every coordinate +-(1 + 2^-8) c (halfway between two bf16 values: every product loses the full rounding), queries copies of base rows
    behind the bootstrap chunk -- the margin of the slab tiles' filter met with equality; and a batch with NaN / Inf / 1e30 inside some queries:
    both equal the fp32-operand tiles bit for bit"""
    import prefhetch_amd as pf
    rng = np.random.default_rng(29)
    d, nb, nq, k = 512, 24000, 140, 10
    mag = np.float32(4.0) * np.float32(1.0 + 2.0 ** -8)      # 4 + half a bf16 ulp: exactly halfway, ties-to-even rounds it down
    xb = (rng.integers(0, 2, (nb, d)) * 2 - 1).astype(np.float32) * mag
    xq = xb[rng.integers(8192, nb, nq)].copy()
    flip = rng.integers(0, d, (nq, 3))
    for i in range(nq):
        xq[i, flip[i]] *= -1                                          # three coordinates away from its row
    f = pf.FlatL2(xb, _dev())
    q = torch.from_numpy(xq).to(_dev())
    D1, I1 = f.search(q, k)
    f.operands16(0)
    D0, I0 = f.search(q, k)
    assert (I1 == I0).all() and (D1.view(torch.int32) == D0.view(torch.int32)).all()
    xq2 = rng.standard_normal((nq, d)).astype(np.float32)
    xq2[3, 7] = np.nan; xq2[70, 0] = np.inf; xq2[139, 500] = 1e30; xq2[5, 100] = -np.inf
    g = pf.FlatL2(rng.standard_normal((nb, d)).astype(np.float32), _dev())
    q2 = torch.from_numpy(xq2).to(_dev())
    D1, I1 = g.search(q2, k)
    g.operands16(0)
    D0, I0 = g.search(q2, k)
    assert (I1 == I0).all() and (D1.view(torch.int32) == D0.view(torch.int32)).all()


@pytest.mark.parametrize("d", [128, 256])
def test_bf16_filter_worst_case_rounding(d):
    """the bound the filter margin is priced on, met with equality: every coordinate is +-(1 + 2^-8) c -- exactly halfway between two
    bf16 values, rounded DOWN by ties-to-even, so every product loses the full 2^-7 -- and queries are copies of base rows (x parallel
    to y, Cauchy-Schwarz tight, all rows of one norm): the true neighbours are where the bf16 dot product is furthest from the fp32 one"""
    import prefhetch_amd as pf
    rng = np.random.default_rng(23)
    nb, nq, k = 30000, 160, 10
    mag = np.float32(4.0) * np.float32(1.0 + 2.0 ** -8)      # 4 + half a bf16 ulp: exactly halfway, ties-to-even rounds it down
    xb = (rng.integers(0, 2, (nb, d)) * 2 - 1).astype(np.float32) * mag
    xq = (rng.integers(0, 2, (nq, d)) * 2 - 1).astype(np.float32) * mag
    picks = rng.integers(8192, nb, 100)                                # rows behind the bootstrap chunk: only the filter can find them
    xq[:100] = xb[picks]
    xq[50:100, :3] *= -1                                              # near-duplicates: three coordinates flipped
    f = pf.FlatL2(xb, _dev())
    assert f.operands16() == 1
    q = torch.from_numpy(xq).to(_dev())
    D1, I1 = f.search(q, k)
    f.operands16(0)
    D0, I0 = f.search(q, k)
    assert (I1 == I0).all() and (D1.view(torch.int32) == D0.view(torch.int32)).all()
    got = I1.cpu().numpy()
    assert all(picks[i] in got[i] for i in range(100))               # each planted row is among its query's results
    assert (D1[:50, 0].cpu().numpy() == 0).all()


def test_exact16_path_refuses_inexact_data():
    """one value outside the exactly-representable set switches the path off: a fraction, a large integer, a huge d"""
    import prefhetch_amd as pf
    dev = _dev()
    rng = np.random.default_rng(8)
    base = rng.integers(0, 256, (5000, 128)).astype(np.float32)
    for bad in (0.5, 257.0, -300.0, 1e6):
        xb = base.copy()
        xb[1234, 77] = bad
        f = pf.FlatL2(xb, dev)
        assert not f.exact16() and f.operands16() == 1                                                 # the image stays, as a filter's operand
    assert not pf.FlatL2(rng.standard_normal((5000, 128)).astype(np.float32), dev).exact16()
    for d in (264, 272, 320):                                                                          # [r4] above 256: the slab tiles' image, a filter's operand whatever the data
        assert pf.FlatL2(rng.integers(0, 256, (500, d)).astype(np.float32), dev).operands16() == 1
    assert pf.FlatL2(rng.integers(0, 256, (500, 100)).astype(np.float32), dev).operands16() == 2      # [r4] any row length up to 256: the image is padded with zeros
    assert pf.FlatL2(base, dev).operands16() == 2


def test_exact16_query_tile_falls_back_per_tile():
    """the base is exact, ONE query is not: its 128-query tile runs the fp32 loop, the others the bf16 loop; every result
    still equals the fp32 path's"""
    import prefhetch_amd as pf
    rng = np.random.default_rng(9)
    xb = rng.integers(0, 256, (30000, 128)).astype(np.float32)
    xq = rng.integers(0, 256, (384, 128)).astype(np.float32)
    xq[200, 5] = 0.25                                                # tile 1 of 3
    xq[383, 0] = 1000.0                                              # tile 2: large but integer -- outside the bound
    active, got16, got32 = _search_both(pf, xb, xq, 50)
    assert active
    assert (got16[1] == got32[1]).all() and (got16[0].view(np.uint32) == got32[0].view(np.uint32)).all()
    Dr, Ir = oracle.flat_l2_search(xb, xq[:128], 50)
    assert (got16[1][:128] == Ir).all() and (got16[0][:128] == Dr).all()


@pytest.mark.parametrize("law", ["uint8", "gauss"])
def test_bf16_paths_with_nonfinite_and_huge_queries(law):
    """NaN, +Inf, -Inf and 1e30 inside query rows: the bf16 tiles (exact path over an 8-bit-valued base, conservative filter
    over a Gaussian one) must return what the fp32-operand tiles return, bit for bit, for those queries and for their finite
    neighbours in the same 128-query tile and in other tiles."""
    import prefhetch_amd as pf
    dev = _dev()
    rng = np.random.default_rng(4242)
    nb, nq, k = 70000, 300, 50
    if law == "uint8":
        xb = rng.integers(0, 256, (nb, 128)).astype(np.float32)
        xq = rng.integers(0, 256, (nq, 128)).astype(np.float32)
    else:
        xb = rng.standard_normal((nb, 128)).astype(np.float32)
        xq = rng.standard_normal((nq, 128)).astype(np.float32)
    xq[3, 5] = np.nan
    xq[3, 77] = -np.nan
    xq[40, 0] = np.inf
    xq[41, 127] = -np.inf
    xq[130, 64] = 1e30
    xq[131, 1] = -3e38
    xq[200, :] = np.nan
    xq[299, 10] = np.float32(np.uint32(0x7F800001).view(np.float32))        # a signalling NaN pattern
    flat = pf.FlatL2(xb, dev)
    assert flat.operands16() == (2 if law == "uint8" else 1)
    dq = torch.from_numpy(xq).to(dev)
    D16, I16 = flat.search(dq, k)
    flat.operands16(0)
    D32, I32 = flat.search(dq, k)
    torch.cuda.synchronize()
    assert torch.equal(I16, I32)
    assert torch.equal(D16.view(torch.int32), D32.view(torch.int32))
    # the finite queries also match the oracle
    fin = [i for i in range(nq) if np.isfinite(xq[i]).all() and np.abs(xq[i]).max() < 1e20][:6]
    Dr, Ir = oracle.flat_l2_search(xb, xq[fin], k)
    got_I, got_D = I32.cpu().numpy()[fin], D32.cpu().numpy()[fin]
    assert (got_I == Ir).all()
    if law == "uint8":
        assert (got_D == Dr).all()
    else:
        assert np.allclose(got_D, Dr, rtol=RTOL, atol=0)
