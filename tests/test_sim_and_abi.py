"""CPU tests: (1) the device NTT core executed on the host lane simulator against the oracle, bit-exact;
(2) host-side table construction; (3) the C-ABI library loads and exports every symbol the header declares
(no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from conftest import ROOT, edge_poly


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


SIM_CASES = [  # (logn, q, arithmetic) 0 = exact-FP64, 1 = u64 Harvey/Shoup, 2 = u64 lazy (q < 2^56)
    (10, 0x7E00001, 2), (12, 0x1FFFFE0001, 2), (13, 0xFFFFFFFC001, 2), (11, 0x3FFFFFFF000001, 2),
    (13, 0x7FFFFFFFE90001, 2), (14, 0x7FFFFFFFE90001, 2), (15, 0x7FFFFFFFE90001, 2), (15, 0xFFFFFFFFF70001, 2),
    (14, 0xFFFFFFFFFFC0001, 1), (15, 0x1FFFFFFFFFE10001, 1),     # 60- and 61-bit primes: Harvey butterflies only
    (10, 0x7E00001, 0), (10, 0x7E00001, 1),
    (12, 0xFFFFEE001, 0), (12, 0x1FFFFE0001, 1),
    (13, 0x7FFFFFD8001, 0), (13, 0xFFFFFEBC001, 0), (13, 0xFFFFFFFC001, 1),
    (11, 0x3FFFFFFF000001, 1), (13, 0x7FFFFFFFE90001, 1), (14, 0x7FFFFFFFE90001, 1),
    (15, 0x7FFFFFFFE90001, 1), (15, 0xFFFFFFFFF70001, 1), (15, 0x7FFFFDB0001, 0),    # last: exact-FP64 at N = 32768
]


@pytest.mark.parametrize("logn,q,arith", SIM_CASES)
def test_device_core_on_host_simulator(sim_lib, logn, q, arith):
    N = 1 << logn
    assert (q - 1) % (2 * N) == 0
    o = oracle.Oracle(N, [q])
    assert sim_lib.pf_sim_psi(logn, q) == o.psi(0)
    rng = np.random.default_rng(logn * 7 + arith)
    kinds = (0, 1) if logn >= 13 else (0, 1, 2, 3, 4)
    for kind in kinds:
        a = edge_poly(rng, N, q, kind)
        b = edge_poly(rng, N, q, 1 - kind if kind < 2 else 0)
        dst = np.empty(N, dtype=np.uint64)
        assert sim_lib.pf_sim_run(logn, q, arith, 0, 0, _p(a), _p(a), _p(dst)) == 0
        A = o.ntt_forward(a)
        assert (dst == A).all()
        assert sim_lib.pf_sim_run(logn, q, arith, 1, 0, _p(A), _p(A), _p(dst)) == 0
        assert (dst == a).all()
        bn = o.ntt_forward(b)
        for flags in range(8):
            src = A if flags & 2 else a
            acc0 = edge_poly(rng, N, q, 0)
            dst = acc0.copy()
            exp = o.ct_pt_mul(np.stack([src, src]).reshape(1, 2, 1, N), bn.reshape(1, 1, N), flags,
                              acc=np.stack([acc0, acc0]).reshape(1, 2, 1, N)).reshape(2, N)[0]
            assert sim_lib.pf_sim_run(logn, q, arith, 2, flags, _p(src), _p(bn), _p(dst)) == 0
            assert (dst == exp).all(), (flags, kind)
    assert sim_lib.pf_sim_range_violations() == 0           # the lazy butterflies stayed inside their analysed bounds


@pytest.mark.parametrize("q,arith", [(0x7FFFFFFFE90001, 2), (0xFFFFFFFFF70001, 2), (0x7FFFFDB0001, 0), (0x1FFFFFFFFFE10001, 1)])
def test_selectable_geometry_1024x32_at_n32768(sim_lib_1024x32, q, arith):
    """N = 32768 as 1024 threads x 32 coefficients (three full passes; both half-exchange selectors on wave-level thread-id
    bits, the dropped position bit in the middle of the index for the second exchange): forward, inverse and the fused
    product against the oracle, on every arithmetic family"""
    lib = sim_lib_1024x32
    N = 32768
    o = oracle.Oracle(N, [q])
    rng = np.random.default_rng(q & 0xFFFF)
    for kind in (0, 1):
        a, b = edge_poly(rng, N, q, kind), edge_poly(rng, N, q, 0)
        dst = np.empty(N, dtype=np.uint64)
        assert lib.pf_sim_run(15, q, arith, 0, 0, _p(a), _p(a), _p(dst)) == 0
        A = o.ntt_forward(a)
        assert (dst == A).all()
        assert lib.pf_sim_run(15, q, arith, 1, 0, _p(A), _p(A), _p(dst)) == 0
        assert (dst == a).all()
        bn = o.ntt_forward(b)
        for flags in (0, 3, 5):
            src = A if flags & 2 else a
            acc0 = edge_poly(rng, N, q, 0)
            dst = acc0.copy()
            exp = o.ct_pt_mul(np.stack([src, src]).reshape(1, 2, 1, N), bn.reshape(1, 1, N), flags, acc=np.stack([acc0, acc0]).reshape(1, 2, 1, N)).reshape(2, N)[0]
            assert lib.pf_sim_run(15, q, arith, 2, flags, _p(src), _p(bn), _p(dst)) == 0
            assert (dst == exp).all(), flags
    assert lib.pf_sim_range_violations() == 0


@pytest.mark.parametrize("logn,q", [(12, 0xFFFFEE001), (12, 0xFFFFC4001), (13, 0x7FFFFFD8001), (13, 0xFFFFFF6C001)])
def test_small_launch_geometry_16_per_thread(sim_lib_x16, logn, q):
    """N = 4096 / 8192 at 16 coefficients per thread (what pf_ct_pt_mul launches when the batch fills the device less than twice): forward,
    inverse and the fused product, exact-FP64 family, against the oracle.  N = 4096 is three full passes here -- the half-buffer exchange
    drops one of two role-swapping index bits from the LDS position, and the slot swizzle must not fold the dropped bit onto its partner
    (tools/lds_swizzle_search.py half_ok)."""
    lib = sim_lib_x16
    N = 1 << logn
    o = oracle.Oracle(N, [q])
    rng = np.random.default_rng(q & 0xFFFF)
    for kind in (0, 1):
        a, b = edge_poly(rng, N, q, kind), edge_poly(rng, N, q, 0)
        dst = np.empty(N, dtype=np.uint64)
        assert lib.pf_sim_run(logn, q, 0, 0, 0, _p(a), _p(a), _p(dst)) == 0
        A = o.ntt_forward(a)
        assert (dst == A).all()
        assert lib.pf_sim_run(logn, q, 0, 1, 0, _p(A), _p(A), _p(dst)) == 0
        assert (dst == a).all()
        bn = o.ntt_forward(b)
        for flags in (0, 1, 3, 5):
            src = A if flags & 2 else a
            acc0 = edge_poly(rng, N, q, 0)
            dst = acc0.copy()
            exp = o.ct_pt_mul(np.stack([src, src]).reshape(1, 2, 1, N), bn.reshape(1, 1, N), flags, acc=np.stack([acc0, acc0]).reshape(1, 2, 1, N)).reshape(2, N)[0]
            assert lib.pf_sim_run(logn, q, 0, 2, flags, _p(src), _p(bn), _p(dst)) == 0
            assert (dst == exp).all(), flags


@pytest.mark.parametrize("qs", [oracle.BFV_DEFAULT[32768][:2] + oracle.BFV_DEFAULT[32768][-1:],
                                [0x7FFFFFFF380001, 0x3FFFFFFF000001, 0xFFFFFFFFF70001]])          # mixed widths: a 54-bit modulus among 55/56-bit ones
def test_two_pass_key_switch_core_on_host_simulator(sim_lib, qs):
    """prefhetch_amd/csrc/ks_split.hpp (the two-pass digit transforms of key switching at N = 32768: pass A = stages 0..6 on
    128 x 64 tiles, pass B = stages 7..14 + 128-bit multiply-accumulate with the key, per half-wave) executed on the host, one OS
    thread per lane: acc[c][J] = sum_I NTT_J(target_I mod m_J) . ksk[I][c][J], bit for bit against the oracle; digits are taken
    WITHOUT reduction modulo m_J (the lazy butterflies' range analysis must hold: zero violations)."""
    N = 32768
    K, D = len(qs), len(qs) - 1
    assert all((q - 1) % (2 * N) == 0 for q in qs)
    rng = np.random.default_rng(K * 1000 + D)
    target = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:D]])
    target[0, :7] = qs[0] - 1
    target[1, -3:] = qs[1] - 1
    ksk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(D)])
    ksk[0, 0, :, :4] = (np.array(qs, dtype=np.uint64) - 1)[:, None]
    acc = np.zeros((2, K, N), dtype=np.uint64)
    mods = np.array(qs, dtype=np.uint64)
    assert sim_lib.pf_sim_ks_split(D, K, _p(mods), _p(target), _p(ksk), _p(acc), None) == 0
    for J, q in enumerate(qs):
        o = oracle.Oracle(N, [q])
        exp = [np.zeros(N, dtype=np.uint64), np.zeros(N, dtype=np.uint64)]
        for I in range(D):
            xt = o.ntt_forward(target[I] % np.uint64(q))
            for c in range(2):
                exp[c] = o.addsub(exp[c], o.dyadic_mul(xt, ksk[I, c, J]), 0)
        for c in range(2):
            assert (acc[c, J] == exp[c]).all(), (J, c)
    # the whole fused path: pass B continues with the first eight INVERSE stages, pass C finishes them, divides by the special
    # prime with rounding and adds into the ciphertext -- against the oracle's key switch (SEAL switch_key_inplace restated)
    ct = np.stack([rng.integers(0, q, (2, N), dtype=np.uint64) for q in qs[:D]], axis=1)            # [2][D][N]
    ct[0, 0, :3] = qs[0] - 1
    exp_ct = oracle.Oracle(N, qs).key_switch(target.reshape(1, D, N), ksk, ct.reshape(1, 2, D, N)).reshape(2, D, N)
    got = ct.copy()
    assert sim_lib.pf_sim_ks_split(D, K, _p(mods), _p(target), _p(ksk), _p(acc), _p(got)) == 0
    assert (got == exp_ct).all()
    assert sim_lib.pf_sim_range_violations() == 0


def test_simulator_golden_n1024(sim_lib, golden):
    for ci in (4, 5):
        g = lambda k: golden[f"c{ci}_{k}"]
        q = int(g("q"))
        dst = np.empty(1024, dtype=np.uint64)
        for arith in (0, 1):
            a = np.ascontiguousarray(g("a"))
            assert sim_lib.pf_sim_run(10, q, arith, 0, 0, _p(a), _p(a), _p(dst)) == 0
            assert (dst == g("ntt_a")).all()
            nb = np.ascontiguousarray(g("ntt_b"))
            assert sim_lib.pf_sim_run(10, q, arith, 2, 0, _p(a), _p(nb), _p(dst)) == 0
            assert (dst == g("a_times_b")).all()


def test_fp64_path_rejected_for_wide_primes(sim_lib):
    a = np.zeros(8192, dtype=np.uint64)
    assert sim_lib.pf_sim_run(13, 0x7FFFFFFFE90001, 0, 0, 0, _p(a), _p(a), _p(a)) == -3


def test_lazy_u64_path_rejected_for_wide_primes(sim_lib):
    a = np.zeros(16384, dtype=np.uint64)
    assert sim_lib.pf_sim_run(14, 0xFFFFFFFFFFC0001, 2, 0, 0, _p(a), _p(a), _p(a)) == -3


def test_library_loads_and_exports_header_symbols():
    import prefhetch_amd as pf
    header = open(os.path.join(ROOT, "include", "prefhetch_hip.h")).read()
    declared = set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", header))
    assert declared == set(pf.SYMBOLS), declared ^ set(pf.SYMBOLS)
    for name in declared:
        assert hasattr(pf.lib, name)
    assert pf.lib.pf_status_str(0) == b"ok"
    # error path without a device: status code, message, no exception, no abort
    n = C.c_int(-1)
    st = pf.lib.pf_device_count(C.byref(n))
    assert st in (0, -4)
    h = C.c_void_p()
    st = pf.lib.pf_ctx_create(C.byref(h), 0, 1000, 1, (C.c_uint64 * 1)(0x7E00001))
    assert st == -2 and b"power of two" in pf.lib.pf_last_error()


def test_python_api_refuses_cpu_tensors():
    import torch
    import prefhetch_amd as pf
    with pytest.raises(ValueError):
        pf.RnsContext(1024, [0x7E00001], device="cpu")
