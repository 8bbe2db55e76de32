import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "ntt_golden.npz"))


def _build_sim(name, extra):
    import ctypes as C
    so = os.path.join(ROOT, "tests", "cpp", name)
    src = os.path.join(ROOT, "tests", "cpp", "sim_ntt.cpp")
    deps = [src] + [os.path.join(ROOT, "prefhetch_amd", "csrc", f) for f in ("ntt_core.hpp", "tables.hpp", "lds_swizzle_tab.hpp", "ks_split.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-std=c++20", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-pthread", "-DPF_RANGE_CHECK", "-shared", "-fPIC"] + extra + [src, "-o", so])
    lib = C.CDLL(so)
    u64p = C.POINTER(C.c_uint64)
    lib.pf_sim_run.argtypes = [C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, u64p, u64p, u64p]
    lib.pf_sim_range_violations.restype = C.c_ulonglong
    lib.pf_sim_psi.restype = C.c_uint64
    lib.pf_sim_psi.argtypes = [C.c_int, C.c_uint64]
    lib.pf_sim_ks_split.argtypes = [C.c_int, C.c_int, u64p, u64p, u64p, u64p, u64p]
    return lib


@pytest.fixture(scope="session")
def sim_lib_1024x32():
    """The same core with the selectable N = 32768 geometry of 1024 threads x 32 coefficients (-DPF_LOGR_15=5)."""
    return _build_sim("libpf_sim_r5.so", ["-DPF_LOGR_15=5"])


@pytest.fixture(scope="session")
def sim_lib_x16():
    """The same core with 16 coefficients per thread at N = 4096 / 8192 (-DPF_LOGR_LARGE=4): the geometry k_ctpt takes for small launches."""
    return _build_sim("libpf_sim_r4.so", ["-DPF_LOGR_LARGE=4"])


@pytest.fixture(scope="session")
def sim_lib():
    """Host execution of the device NTT core (tests/cpp/sim_ntt.cpp), built on demand."""
    return _build_sim("libpf_sim.so", [])


def edge_poly(rng, N, q, kind):
    """kind 0 random, 1 all q-1, 2 all zero, 3 single q-1 spike, 4 all one."""
    if kind == 0:
        return rng.integers(0, q, N, dtype=np.uint64)
    if kind == 1:
        return np.full(N, q - 1, dtype=np.uint64)
    if kind == 2:
        return np.zeros(N, dtype=np.uint64)
    if kind == 3:
        a = np.zeros(N, dtype=np.uint64)
        a[int(rng.integers(0, N))] = q - 1
        return a
    return np.ones(N, dtype=np.uint64)
