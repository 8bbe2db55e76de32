"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
prefhetch_amd never does.  See the header of oracle/pf_oracle.c for what is restated and why
parity is unpinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpf_oracle.so")

# SEAL CoeffModulus::BFVDefault(N) primes (last prime of each set is the
# key-switching "special prime").  Minimal primitive 2N-th roots from the same table.
BFV_DEFAULT = {
    1024: [0x7E00001],
    2048: [0x3FFFFFFF000001],
    4096: [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001],
    8192: [0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001, 0xFFFFFEBC001],
    32768: [0x7FFFFFFFE90001, 0x7FFFFFFFBF0001, 0x7FFFFFFFBD0001, 0x7FFFFFFFBA0001, 0x7FFFFFFFAA0001,
            0x7FFFFFFFA50001, 0x7FFFFFFF9F0001, 0x7FFFFFFF7E0001, 0x7FFFFFFF770001, 0x7FFFFFFF380001,
            0x7FFFFFFF330001, 0x7FFFFFFF2D0001, 0x7FFFFFFF170001, 0x7FFFFFFF150001, 0x7FFFFFFEF00001,
            0xFFFFFFFFF70001],
}
MIN_PSI = {
    1024: [73993],
    4096: [24250113, 29008497, 8625844],
    8192: [1734247217, 304486499, 331339694, 9366611238, 632352760],
    32768: [1155186985540, 631260524634, 1526647220035, 455957817523, 1650884166641, 10316746886,
            768741990072, 3911086673862, 5947090524825, 47595902954, 2691682578057, 3903338373,
            235185854118, 1769787302793, 3151164484090, 724233080554],
}


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "pf_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libpf_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u64p, f32p, i64p = C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_int64)
        L.pfo_ctx_create.restype = C.c_void_p
        L.pfo_ctx_create.argtypes = [C.c_uint32, C.c_uint32, u64p]
        L.pfo_ctx_destroy.argtypes = [C.c_void_p]
        L.pfo_ctx_psi.restype = C.c_uint64
        L.pfo_ctx_psi.argtypes = [C.c_void_p, C.c_uint32]
        L.pfo_min_primitive_root.argtypes = [C.c_uint64, C.c_uint64, u64p]
        L.pfo_is_prime.argtypes = [C.c_uint64]
        L.pfo_ntt_forward.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_int]
        L.pfo_ntt_inverse.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_int]
        L.pfo_dyadic_mul.argtypes = [C.c_void_p, u64p, u64p, u64p, C.c_size_t, C.c_int]
        L.pfo_poly_addsub.argtypes = [C.c_void_p, u64p, u64p, u64p, C.c_size_t, C.c_int, C.c_int]
        L.pfo_ct_pt_mul.argtypes = [C.c_void_p, u64p, u64p, C.c_int, u64p, C.c_size_t, C.c_int, C.c_int]
        L.pfo_ivfpq_search_lists.restype = C.c_size_t
        L.pfo_ivfpq_search_lists.argtypes = [f32p, C.c_size_t, i64p, C.c_size_t, f32p, C.c_size_t, C.c_size_t, C.c_size_t, f32p,
                                             C.POINTER(C.c_uint8), i64p, u64p, f32p, i64p, u64p]
        L.pfo_key_switch.argtypes = [C.c_void_p, u64p, u64p, u64p, C.c_size_t, C.c_int]
        L.pfo_precise_search.argtypes = [f32p, f32p, i64p, C.c_size_t, C.c_size_t, C.c_size_t, f32p]
        L.pfo_gather_rows.argtypes = [f32p, i64p, C.c_size_t, C.c_size_t, f32p]
        L.pfo_flat_l2_search.argtypes = [f32p, C.c_size_t, C.c_size_t, f32p, C.c_size_t, C.c_size_t, f32p, i64p, C.c_int, C.c_int]
        L.pfo_flat_l2_search_f32.argtypes = [f32p, C.c_size_t, C.c_size_t, f32p, C.c_size_t, C.c_size_t, f32p, i64p, C.c_int]
        L.pfo_l2_reservoir_block.argtypes = [f32p, C.c_size_t, C.c_size_t, C.c_size_t, f32p, f32p, C.c_int64, C.c_size_t, C.c_size_t,
                                             f32p, i64p, C.POINTER(C.c_uint32), f32p, C.c_int, f32p, i64p, C.c_int]
        L.pfo_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


class Oracle:
    """Oracle A context for ring degree N and a list of RNS moduli."""

    ADD, SUB, NEG = 0, 1, 2
    ACCUMULATE, IN_NTT, OUT_NTT = 1, 2, 4

    def __init__(self, N, moduli):
        self.N, self.moduli, self.L = int(N), [int(q) for q in moduli], len(moduli)
        arr = np.array(self.moduli, dtype=np.uint64)
        self._h = lib().pfo_ctx_create(self.N, self.L, _p(arr, C.c_uint64))
        if not self._h:
            raise ValueError("oracle: invalid N/moduli (need prime q < 2^61 with q = 1 mod 2N)")

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:       # (module globals are gone when the interpreter is shutting down)
            lib().pfo_ctx_destroy(self._h)
            self._h = None

    def psi(self, l):
        return int(lib().pfo_ctx_psi(self._h, l))

    def _count(self, a):
        assert a.size % self.N == 0
        return a.size // self.N

    def ntt_forward(self, polys, threads=0):
        out = _u64(polys).copy()
        lib().pfo_ntt_forward(self._h, _p(out, C.c_uint64), self._count(out), threads)
        return out

    def ntt_inverse(self, polys, threads=0):
        out = _u64(polys).copy()
        lib().pfo_ntt_inverse(self._h, _p(out, C.c_uint64), self._count(out), threads)
        return out

    def dyadic_mul(self, a, b, threads=0):
        a, b = _u64(a), _u64(b)
        out = np.empty_like(a)
        lib().pfo_dyadic_mul(self._h, _p(a, C.c_uint64), _p(b, C.c_uint64), _p(out, C.c_uint64), self._count(a), threads)
        return out

    def addsub(self, a, b, op, threads=0):
        a = _u64(a)
        b = a if b is None else _u64(b)
        out = np.empty_like(a)
        lib().pfo_poly_addsub(self._h, _p(a, C.c_uint64), _p(b, C.c_uint64), _p(out, C.c_uint64), self._count(a), op, threads)
        return out

    def key_switch(self, target, ksk, ct, threads=0):
        """Context moduli = key moduli (special prime last).  target [B,L,N], ksk [L,2,L+1,N] (NTT form),
        ct [B,2,L,N] coefficient form; returns ct + switched polynomial."""
        target, ksk = _u64(target), _u64(ksk)
        out = _u64(ct).copy()
        L = self.L - 1
        B = target.size // (L * self.N)
        assert ksk.size == L * 2 * self.L * self.N and out.size == B * 2 * L * self.N
        lib().pfo_key_switch(self._h, _p(target, C.c_uint64), _p(ksk, C.c_uint64), _p(out, C.c_uint64), B, threads)
        return out

    def ct_pt_mul(self, ct, pt_ntt, flags=0, acc=None, threads=0):
        """ct [B,2,L,N], pt_ntt [B,L,N] or [1,L,N]/[L,N] (broadcast)."""
        ct, pt = _u64(ct), _u64(pt_ntt)
        B = ct.size // (2 * self.L * self.N)
        bcast = int(pt.size == self.L * self.N)
        out = _u64(acc).copy() if (flags & self.ACCUMULATE) else np.empty_like(ct)
        lib().pfo_ct_pt_mul(self._h, _p(ct, C.c_uint64), _p(pt, C.c_uint64), bcast, _p(out, C.c_uint64), B, flags, threads)
        return out


def precise_search(base, xq, ids):
    base = np.ascontiguousarray(base, np.float32)
    xq = np.ascontiguousarray(xq, np.float32)
    ids = np.ascontiguousarray(ids, np.int64)
    nq, c = ids.shape
    out = np.empty((nq, c), np.float32)
    lib().pfo_precise_search(_p(base, C.c_float), _p(xq, C.c_float), _p(ids, C.c_int64), nq, c, base.shape[1], _p(out, C.c_float))
    return out


def gather_rows(base, ids):
    base = np.ascontiguousarray(base, np.float32)
    ids = np.ascontiguousarray(ids, np.int64)
    out = np.empty(ids.shape + (base.shape[1],), np.float32)
    lib().pfo_gather_rows(_p(base, C.c_float), _p(ids, C.c_int64), ids.size, base.shape[1], _p(out, C.c_float))
    return out


def apply_galois(polys, galois_elt, moduli):
    """out(X) = in(X^g) mod (X^N + 1) per limb-polynomial (SEAL util::GaloisTool::apply_galois, coefficient form;
    restated from the definition): polys [..., L, N] canonical residues, limb = second-to-last axis."""
    polys = np.asarray(polys, np.uint64)
    N = polys.shape[-1]
    flat = polys.reshape(-1, len(moduli), N)
    out = np.zeros_like(flat)
    i = np.arange(N, dtype=np.int64)
    j = (i * int(galois_elt)) % (2 * N)
    dst = j % N
    neg = j >= N
    for l, q in enumerate(moduli):
        v = flat[:, l, :]
        w = np.where(neg[None, :] & (v != 0), np.uint64(q) - v, v)
        out[:, l, dst] = w
    return out.reshape(polys.shape)


def pack_rows(base, ids, N, moduli):
    """Plaintext packing of the encrypted precise search (the build's own encoding; the reference leaves this step as
    TODOs, include/client/client_lib.h:14,28-30): p(X) = sum_j sum_i x[ids[p][j]][i] X^(d*j - i) mod (X^N + 1), values
    rounded to integers, out[p][l][c] canonical mod moduli[l].  ids [n_polys][rows_per_poly]; an id outside the base is a
    zero row.  Written from the definition (exponent arithmetic with sign), not from the kernel's inverse mapping."""
    base = np.asarray(base, np.float32)
    ids = np.asarray(ids, np.int64)
    nb, d = base.shape
    n_polys, rows = ids.shape
    assert rows * d <= N
    out = np.zeros((n_polys, len(moduli), N), dtype=np.uint64)
    for p in range(n_polys):
        coeff = [0] * N
        for j in range(rows):
            rid = int(ids[p, j])
            if rid < 0 or rid >= nb:
                continue
            row = np.rint(base[rid]).astype(np.int64)
            for i in range(d):
                e = d * j - i
                sign = 1
                if e < 0:
                    e += N
                    sign = -1
                coeff[e] += sign * int(row[i])
        for l, q in enumerate(moduli):
            out[p, l] = np.array([c % q for c in coeff], dtype=np.uint64)
    return out


def flat_l2_search(xb, xq, k, mode=0, threads=0, f32=False):
    xb = np.ascontiguousarray(xb, np.float32)
    xq = np.ascontiguousarray(xq, np.float32)
    nq = xq.shape[0]
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    if f32:
        lib().pfo_flat_l2_search_f32(_p(xb, C.c_float), xb.shape[0], xb.shape[1], _p(xq, C.c_float), nq, k, _p(D, C.c_float), _p(I, C.c_int64), threads)
    else:
        lib().pfo_flat_l2_search(_p(xb, C.c_float), xb.shape[0], xb.shape[1], _p(xq, C.c_float), nq, k, _p(D, C.c_float), _p(I, C.c_int64), mode, threads)
    return D, I


def flat_l2_search_blas(xb, xq, k, threads=0, block=32768):
    """faiss IndexFlatL2::search as it runs for nq >= 20 (exhaustive_L2sqr_blas): blocks of the base matrix, inner
    products by BLAS sgemm (numpy's OpenBLAS, its own thread pool), dist = |x|^2 + |y|^2 - 2 x.y, per-query reservoir
    (ReservoirTopN) cut back by quickselect -- the CPU baseline of the pre-filter.  fp32 throughout, so distances carry
    the cancellation error of the expansion (exact on SIFT-like integer data up to 2^24)."""
    xb = np.ascontiguousarray(xb, np.float32)
    xq = np.ascontiguousarray(xq, np.float32)
    nq, nb = xq.shape[0], xb.shape[0]
    qn = np.einsum("ij,ij->i", xq, xq).astype(np.float32)
    cap = max(2 * k, 256)
    res_d = np.empty((nq, cap), np.float32)
    res_i = np.empty((nq, cap), np.int64)
    cnt = np.zeros(nq, np.uint32)
    thr = np.full(nq, np.inf, np.float32)
    D = np.empty((nq, k), np.float32)
    I = np.empty((nq, k), np.int64)
    ip = np.empty((nq, block), np.float32)
    if nb == 0:
        D[:] = np.inf
        I[:] = -1
        return D, I
    for lo in range(0, nb, block):
        blk = xb[lo:lo + block]
        n = blk.shape[0]
        bn = np.einsum("ij,ij->i", blk, blk).astype(np.float32)
        np.matmul(xq, blk.T, out=ip[:, :n])
        lib().pfo_l2_reservoir_block(_p(ip, C.c_float), nq, n, block, _p(qn, C.c_float), _p(bn, C.c_float), lo, k, cap,
                                     _p(res_d, C.c_float), _p(res_i, C.c_int64), _p(cnt, C.c_uint32), _p(thr, C.c_float),
                                     int(lo + n >= nb), _p(D, C.c_float), _p(I, C.c_int64), threads)
    return D, I


def ivfpq_search_lists(xq, probe, centroids, codebooks, codes, ids, list_off):
    """ADC scan of the given lists (IndexIVFPQ::search_encrypted restated).  Returns (D, I, list_sizes)."""
    xq = np.ascontiguousarray(xq, np.float32)
    probe = np.ascontiguousarray(probe, np.int64)
    centroids = np.ascontiguousarray(centroids, np.float32)
    codebooks = np.ascontiguousarray(codebooks, np.float32)
    codes = np.ascontiguousarray(codes, np.uint8)
    ids = np.ascontiguousarray(ids, np.int64)
    list_off = np.ascontiguousarray(list_off, np.uint64)
    nq, nprobe = probe.shape
    nlist, d = centroids.shape
    M = codebooks.shape[0]
    cap = int(sum(int(list_off[l + 1] - list_off[l]) for l in probe.ravel() if 0 <= l < nlist))
    D = np.empty(max(cap, 1), np.float32)
    I = np.empty(max(cap, 1), np.int64)
    sizes = np.zeros(nq, np.uint64)
    n = lib().pfo_ivfpq_search_lists(_p(xq, C.c_float), nq, _p(probe, C.c_int64), nprobe, _p(centroids, C.c_float), nlist, d, M,
                                     _p(codebooks, C.c_float), _p(codes, C.c_uint8), _p(ids, C.c_int64), _p(list_off, C.c_uint64),
                                     _p(D, C.c_float), _p(I, C.c_int64), _p(sizes, C.c_uint64))
    return D[:n], I[:n], sizes


def max_threads():
    return int(lib().pfo_max_threads())
