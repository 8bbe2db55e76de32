"""Host-side Python handles over the C ABI (torch supplies device memory and streams only).

RnsContext  -- one RNS ring (N, moduli): NTT, dyadic product, add/sub/negate, fused ct x pt.
               Stands in for the SEAL objects the reference links (CMakeLists.txt:33-38) at the
               Evaluator::multiply_plain / transform_to_ntt / add_inplace level.
FlatL2      -- faiss::IndexFlatL2 (reference include/server/server_lib.h:14) plus the gathered exact
               distances of Server::preciseSearch and the row gather of preciseVectorPIR.

uint64 residues travel in torch.int64 tensors (same bits); numpy uint64 <-> torch via .view(np.int64).
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib

ACCUMULATE, IN_NTT, OUT_NTT = 1, 2, 4


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _dev_index(device):
    d = torch.device(device)
    if d.type != "cuda":
        raise ValueError("prefhetch_amd runs on a HIP device only (torch device type 'cuda'); there is no CPU path")
    return d.index if d.index is not None else torch.cuda.current_device()


def _req(t, dtype, device_index, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device.index == device_index and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"{name}: need a contiguous {dtype} tensor on cuda:{device_index}")
    return C.c_void_p(t.data_ptr())


def to_device_u64(a, device):
    """numpy uint64 array -> torch.int64 tensor on `device` (bit-identical)."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).to(device)


def to_host_u64(t):
    return t.detach().cpu().numpy().view(np.uint64)


class RnsContext:
    def __init__(self, N, moduli, device="cuda:0"):
        self.device_index = _dev_index(device)
        self.device = torch.device("cuda", self.device_index)
        self.N, self.moduli, self.L = int(N), [int(q) for q in moduli], len(moduli)
        arr = (C.c_uint64 * self.L)(*self.moduli)
        h = C.c_void_p()
        check(lib.pf_ctx_create(C.byref(h), self.device_index, self.N, self.L, arr), "pf_ctx_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.pf_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def info(self):
        psi = (C.c_uint64 * self.L)()
        path = (C.c_int32 * self.L)()
        check(lib.pf_ctx_info(self._h, None, None, None, psi, path), "pf_ctx_info")
        return {"psi": list(psi), "arith_path": list(path)}

    def force_u64(self, mode=1):
        """0 automatic, 1 64-bit integer butterflies (lazy family where every q < 2^56), 2 general Harvey butterflies."""
        check(lib.pf_ctx_force_u64(self._h, int(mode)), "pf_ctx_force_u64")

    def _count(self, t):
        if t.numel() % self.N:
            raise ValueError("tensor size is not a multiple of N")
        return t.numel() // self.N

    def ntt_forward_(self, polys):
        p = _req(polys, torch.int64, self.device_index, "polys")
        check(lib.pf_ntt_forward(self._h, p, self._count(polys), _stream(self.device)), "pf_ntt_forward")
        return polys

    def ntt_forward(self, polys, out=None):
        """out-of-place forward transform (polys untouched)"""
        out = torch.empty_like(polys) if out is None else out
        ps, pd = _req(polys, torch.int64, self.device_index, "polys"), _req(out, torch.int64, self.device_index, "out")
        check(lib.pf_ntt_forward_to(self._h, ps, pd, self._count(polys), _stream(self.device)), "pf_ntt_forward_to")
        return out

    def ntt_inverse(self, polys, out=None):
        out = torch.empty_like(polys) if out is None else out
        ps, pd = _req(polys, torch.int64, self.device_index, "polys"), _req(out, torch.int64, self.device_index, "out")
        check(lib.pf_ntt_inverse_to(self._h, ps, pd, self._count(polys), _stream(self.device)), "pf_ntt_inverse_to")
        return out

    def ntt_inverse_(self, polys):
        p = _req(polys, torch.int64, self.device_index, "polys")
        check(lib.pf_ntt_inverse(self._h, p, self._count(polys), _stream(self.device)), "pf_ntt_inverse")
        return polys

    def _binary(self, fn, name, a, b, out):
        out = torch.empty_like(a) if out is None else out
        pa, pb, po = (_req(t, torch.int64, self.device_index, n) for t, n in ((a, "a"), (b, "b"), (out, "out")))
        if a.numel() != b.numel() or a.numel() != out.numel():
            raise ValueError("size mismatch")
        check(fn(self._h, pa, pb, po, self._count(a), _stream(self.device)), name)
        return out

    def dyadic_mul(self, a, b, out=None):
        return self._binary(lib.pf_dyadic_mul, "pf_dyadic_mul", a, b, out)

    def add(self, a, b, out=None):
        return self._binary(lib.pf_poly_add, "pf_poly_add", a, b, out)

    def sub(self, a, b, out=None):
        return self._binary(lib.pf_poly_sub, "pf_poly_sub", a, b, out)

    def negate(self, a, out=None):
        out = torch.empty_like(a) if out is None else out
        pa, po = _req(a, torch.int64, self.device_index, "a"), _req(out, torch.int64, self.device_index, "out")
        check(lib.pf_poly_negate(self._h, pa, po, self._count(a), _stream(self.device)), "pf_poly_negate")
        return out

    def ct_pt_mul(self, ct, pt_ntt, out=None, flags=0):
        """ct [B,2,L,N]; pt_ntt [B,L,N] or [1,L,N] (broadcast); returns/accumulates into out [B,2,L,N]."""
        per_ct = 2 * self.L * self.N
        if ct.numel() % per_ct:
            raise ValueError("ct size is not a multiple of 2*L*N")
        B = ct.numel() // per_ct
        pt_count = pt_ntt.numel() // (self.L * self.N)
        if pt_ntt.numel() != pt_count * self.L * self.N or pt_count not in (1, B):
            raise ValueError("pt_ntt must hold 1 or B plaintexts of L*N residues")
        if out is None:
            if flags & ACCUMULATE:
                raise ValueError("ACCUMULATE needs an `out` operand")
            out = torch.empty_like(ct)
        pc, pp, po = (_req(t, torch.int64, self.device_index, n) for t, n in ((ct, "ct"), (pt_ntt, "pt_ntt"), (out, "out")))
        check(lib.pf_ct_pt_mul(self._h, pc, pp, pt_count, po, B, int(flags), _stream(self.device)), "pf_ct_pt_mul")
        return out


    def ct_pt_mul_fanout(self, ct, pt_ntt, fanout, out=None, flags=0):
        """out[b] = ct[b // fanout] x pt_ntt[b]: ct [ceil(B/fanout),2,L,N], pt_ntt [B,L,N] -> out [B,2,L,N]."""
        if int(fanout) < 1:
            raise ValueError("fanout must be at least 1")
        B = pt_ntt.numel() // (self.L * self.N)
        if pt_ntt.numel() != B * self.L * self.N or ct.numel() != -(-B // fanout) * 2 * self.L * self.N:
            raise ValueError("ct must hold ceil(B / fanout) ciphertexts for the B plaintexts of pt_ntt")
        if out is None:
            if flags & ACCUMULATE:
                raise ValueError("ACCUMULATE needs an `out` operand")
            out = torch.empty((B, 2, self.L, self.N), dtype=torch.int64, device=self.device)
        pc, pp, po = (_req(t, torch.int64, self.device_index, n) for t, n in ((ct, "ct"), (pt_ntt, "pt_ntt"), (out, "out")))
        check(lib.pf_ct_pt_mul_fanout(self._h, pc, pp, po, B, int(fanout), int(flags), _stream(self.device)), "pf_ct_pt_mul_fanout")
        return out

    def apply_galois(self, polys, galois_elt, out=None):
        """out(X) = polys(X^galois_elt) mod (X^N + 1) on coefficient-form limb-polynomials [..., L, N]."""
        if polys.numel() % (self.L * self.N):
            raise ValueError("size is not a multiple of L*N")
        if out is None:
            out = torch.empty_like(polys)
        pi, po = _req(polys, torch.int64, self.device_index, "polys"), _req(out, torch.int64, self.device_index, "out")
        check(lib.pf_apply_galois(self._h, pi, po, polys.numel() // self.N, int(galois_elt), _stream(self.device)), "pf_apply_galois")
        return out

    def mul_monomial(self, polys, exponent, out=None):
        """out = polys * X^exponent mod (X^N + 1), exponent in [0, 2N), coefficient-form limb-polynomials [..., L, N]."""
        if polys.numel() % (self.L * self.N):
            raise ValueError("size is not a multiple of L*N")
        if out is None:
            out = torch.empty_like(polys)
        pi, po = _req(polys, torch.int64, self.device_index, "polys"), _req(out, torch.int64, self.device_index, "out")
        check(lib.pf_poly_mul_monomial(self._h, pi, po, polys.numel() // self.N, int(exponent), _stream(self.device)), "pf_poly_mul_monomial")
        return out

    def addsub_monomial(self, a, b, exponent, sum_out=None, diff_out=None):
        """(a + b, (a - b) * X^exponent) in one pass; coefficient-form limb-polynomials [..., L, N]."""
        if a.numel() % (self.L * self.N) or a.numel() != b.numel():
            raise ValueError("sizes must agree and be a multiple of L*N")
        sum_out = torch.empty_like(a) if sum_out is None else sum_out
        diff_out = torch.empty_like(a) if diff_out is None else diff_out
        pa, pb, ps, pd = (_req(t, torch.int64, self.device_index, n) for t, n in ((a, "a"), (b, "b"), (sum_out, "sum"), (diff_out, "diff")))
        check(lib.pf_poly_addsub_monomial(self._h, pa, pb, ps, pd, a.numel() // self.N, int(exponent), _stream(self.device)), "pf_poly_addsub_monomial")
        return sum_out, diff_out

    def apply_galois_ct(self, ct, galois_elt):
        """The permutation over ciphertexts [B,2,L,N], laid out for the key switch that follows: returns (ct_out, target) with
        ct_out[b,0] = tau(ct[b,0]), ct_out[b,1] = 0, target[b] = tau(ct[b,1]) ([B,L,N])."""
        if ct.numel() % (2 * self.L * self.N):
            raise ValueError("size is not a multiple of 2*L*N")
        B = ct.numel() // (2 * self.L * self.N)
        out = torch.empty_like(ct)
        target = torch.empty((B, self.L, self.N), dtype=torch.int64, device=self.device)
        pi, po, pt = (_req(t, torch.int64, self.device_index, n) for t, n in ((ct, "ct"), (out, "out"), (target, "target")))
        check(lib.pf_apply_galois_ct(self._h, pi, po, pt, B, int(galois_elt), _stream(self.device)), "pf_apply_galois_ct")
        return out, target

    def pack_rows(self, flat, ids, out=None, ntt=False):
        """Plaintext polynomials of the encrypted precise search: ids [n_polys, rows_per_poly] (int64, device) rows of
        the FlatL2 index `flat` -> [n_polys, L, N] coefficient-form residues (follow with ntt_forward_), or, with
        ntt=True, directly their NTT form (packing fused into the forward transform)."""
        if ids.dim() != 2:
            raise ValueError("ids must be [n_polys, rows_per_poly]")
        n_polys, rows = ids.shape
        if out is None:
            out = torch.empty((n_polys, self.L, self.N), dtype=torch.int64, device=self.device)
        pi, po = _req(ids, torch.int64, self.device_index, "ids"), _req(out, torch.int64, self.device_index, "out")
        if out.numel() != n_polys * self.L * self.N:
            raise ValueError("out must hold n_polys * L * N residues")
        fn = lib.pf_pack_rows_ntt if ntt else lib.pf_pack_rows
        check(fn(self._h, flat._h, pi, n_polys, rows, po, _stream(self.device)), "pf_pack_rows_ntt" if ntt else "pf_pack_rows")
        return out

    def ct_pt_dot(self, ct_ntt, pt_ntt, chunk):
        """out[g] = sum over the plaintexts p of chunk g of ct_ntt[p mod n_ct] . pt_ntt[p], all in NTT form: ct_ntt [n_ct,2,L,N],
        pt_ntt [n_pt,L,N] -> [ceil(n_pt / chunk),2,L,N]; chunk divides n_ct."""
        n_ct, n_pt = ct_ntt.numel() // (2 * self.L * self.N), pt_ntt.numel() // (self.L * self.N)
        out = torch.empty((-(-n_pt // int(chunk)), 2, self.L, self.N), dtype=torch.int64, device=self.device)
        pc, pp, po = (_req(t, torch.int64, self.device_index, n) for t, n in ((ct_ntt, "ct_ntt"), (pt_ntt, "pt_ntt"), (out, "out")))
        check(lib.pf_ct_pt_dot(self._h, pc, n_ct, pp, n_pt, int(chunk), po, _stream(self.device)), "pf_ct_pt_dot")
        return out

    def ct_rows_mul(self, ct_ntt, flat, ids, fanout, out=None):
        """out[b] = ct_ntt[b // fanout] x pack(ids[b]) in one kernel (pack_rows(ntt=True) + ct_pt_mul_fanout(IN_NTT), bit for
        bit, without the plaintexts ever existing in memory): ct_ntt [ceil(B/fanout),2,L,N] NTT form, ids [B, rows_per_poly]
        -> out [B,2,L,N] coefficient form."""
        if ids.dim() != 2:
            raise ValueError("ids must be [B, rows_per_poly]")
        if int(fanout) < 1:
            raise ValueError("fanout must be at least 1")
        B, rows = ids.shape
        if ct_ntt.numel() != -(-B // int(fanout)) * 2 * self.L * self.N:
            raise ValueError("ct_ntt must hold ceil(B / fanout) ciphertexts")
        if out is None:
            out = torch.empty((B, 2, self.L, self.N), dtype=torch.int64, device=self.device)
        if out.numel() != B * 2 * self.L * self.N:
            raise ValueError("out must hold B ciphertexts")
        pc, pi, po = (_req(t, torch.int64, self.device_index, n) for t, n in ((ct_ntt, "ct_ntt"), (ids, "ids"), (out, "out")))
        check(lib.pf_ct_rows_mul(self._h, pc, flat._h, pi, B, rows, int(fanout), po, _stream(self.device)), "pf_ct_rows_mul")
        return out

    def key_switch_reserve(self, B):
        """Allocates the key-switching workspace for calls of up to B polynomials, so that key_switch_ never allocates."""
        check(lib.pf_key_switch_reserve(self._h, int(B)), "pf_key_switch_reserve")

    def key_switch_(self, target, ksk, ct):
        """This context holds the key moduli (special prime last).  target [B,D,N], ksk [D,2,D+1,N] (NTT form),
        ct [B,2,D,N]: the switched polynomial is added into ct in place."""
        D = self.L - 1
        if target.numel() % (D * self.N):
            raise ValueError("target size is not a multiple of D*N")
        B = target.numel() // (D * self.N)
        if ksk.numel() != D * 2 * self.L * self.N or ct.numel() != B * 2 * D * self.N:
            raise ValueError("ksk must be [D,2,D+1,N] and ct [B,2,D,N]")
        pt, pk, pc = (_req(t, torch.int64, self.device_index, n) for t, n in ((target, "target"), (ksk, "ksk"), (ct, "ct")))
        check(lib.pf_key_switch(self._h, pt, pk, pc, B, _stream(self.device)), "pf_key_switch")
        return ct


class FlatL2:
    def __init__(self, xb, device="cuda:0"):
        """xb: [nb, d] float32 numpy array or torch tensor (host or device); copied into HBM."""
        self.device_index = _dev_index(device)
        self.device = torch.device("cuda", self.device_index)
        if isinstance(xb, np.ndarray):
            xb = torch.from_numpy(np.ascontiguousarray(xb, dtype=np.float32))
        if xb.dtype != torch.float32 or xb.dim() != 2:
            raise ValueError("xb must be [nb, d] float32")
        xb = xb.contiguous()
        self.nb, self.d = int(xb.shape[0]), int(xb.shape[1])
        h = C.c_void_p()
        check(lib.pf_flat_create(C.byref(h), self.device_index, C.c_void_p(xb.data_ptr()), self.nb, self.d), "pf_flat_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.pf_flat_destroy(self._h)
            self._h = None

    __del__ = close

    def operands16(self, mode=-1):
        """bf16 tiles of the batch pre-filter (pf_flat_exact16): mode 1 on (default), 0 fp32 operands always, -1 query.
        Returns 2 when on with an exactly representable base (the tiles evaluate the distance test itself), 1 when on as a
        conservative filter (an inexact base, or rows longer than 256 values: survivors re-evaluated by the fp32 chain), 0 when off."""
        a = C.c_int()
        check(lib.pf_flat_exact16(self._h, int(mode), C.byref(a)), "pf_flat_exact16")
        return int(a.value)

    def operands8(self, mode=-1):
        """int8 tiles of the batch pre-filter (pf_flat_operands8): mode 1 on (default), 0 off, -1 query.  True when the base is 8-bit data
        (every value an integer in [0, 255], d up to 128 and a multiple of 32 or not a multiple of 16 -- then padded to one), its int8 image exists and the path is on."""
        a = C.c_int()
        check(lib.pf_flat_operands8(self._h, int(mode), C.byref(a)), "pf_flat_operands8")
        return bool(a.value)

    def exact16(self, mode=-1):
        """True when the bf16 tiles are on AND the base passed the on-device exactness check (operands16() == 2)."""
        return self.operands16(mode) == 2

    def reserve(self, nq_max, k_max):
        check(lib.pf_flat_reserve(self._h, nq_max, k_max), "pf_flat_reserve")

    def search(self, xq, k):
        pq = _req(xq, torch.float32, self.device_index, "xq")
        if xq.dim() != 2 or xq.shape[1] != self.d:
            raise ValueError("xq must be [nq, d]")
        nq = xq.shape[0]
        D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        check(lib.pf_flat_search(self._h, pq, nq, k, C.c_void_p(D.data_ptr()), C.c_void_p(I.data_ptr()), _stream(self.device)), "pf_flat_search")
        return D, I

    def search_packed(self, xq, k, out=None):
        """The same search, emitting the exchange record of the multi-GPU gather instead of (D, I): int32 [nq, k, 3] =
        (id low word, id high word, distance bits), written by the selection kernel itself.  `out` may be a view into a
        larger gathered buffer (this rank's block), so that the all-gather runs in place."""
        pq = _req(xq, torch.float32, self.device_index, "xq")
        if xq.dim() != 2 or xq.shape[1] != self.d:
            raise ValueError("xq must be [nq, d]")
        nq = xq.shape[0]
        if out is None:
            out = torch.empty((nq, k, 3), dtype=torch.int32, device=self.device)
        po = _req(out, torch.int32, self.device_index, "out")
        if out.numel() != nq * k * 3:
            raise ValueError("out must hold nq * k * 3 int32 words")
        check(lib.pf_flat_search_packed(self._h, pq, nq, k, None, None, po, _stream(self.device)), "pf_flat_search_packed")
        return out

    def l2_gathered(self, xq, ids):
        pq = _req(xq, torch.float32, self.device_index, "xq")
        pi = _req(ids, torch.int64, self.device_index, "ids")
        nq, c = ids.shape
        if xq.shape != (nq, self.d):
            raise ValueError("xq must be [nq, d] with nq = ids.shape[0]")
        D = torch.empty((nq, c), dtype=torch.float32, device=self.device)
        check(lib.pf_l2_gathered(self._h, pq, pi, nq, c, C.c_void_p(D.data_ptr()), _stream(self.device)), "pf_l2_gathered")
        return D

    def gather_rows(self, ids):
        pi = _req(ids, torch.int64, self.device_index, "ids")
        out = torch.empty(tuple(ids.shape) + (self.d,), dtype=torch.float32, device=self.device)
        check(lib.pf_gather_rows(self._h, pi, ids.numel(), C.c_void_p(out.data_ptr()), _stream(self.device)), "pf_gather_rows")
        return out


class DeviceGroup:
    """pf_multi_*: ONE process driving G devices, one host thread + stream per device, batch sharded contiguously, RNS
    tables and base matrix replicated, one all-gather (RCCL, or direct copies when a device is listed twice) of the
    packed top-k blocks.  SURVEY.md section 8(e); the reference is single-device (src/server/server_lib.cpp:48-53)."""
    AUTO, RCCL, PEER_COPY = 0, 1, 2

    def __init__(self, devices, exchange=0):
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        check(lib.pf_multi_create(C.byref(h), arr, len(self.devices), int(exchange)), "pf_multi_create")
        self._h = h
        ex = C.c_int()
        check(lib.pf_multi_info(self._h, None, None, C.byref(ex)), "pf_multi_info")
        self.exchange = "rccl" if ex.value == self.RCCL else "peer_copy"
        self.G = len(self.devices)

    def close(self):
        if getattr(self, "_h", None):
            lib.pf_multi_destroy(self._h)
            self._h = None

    __del__ = close

    def ring(self, N, moduli):
        self.N, self.L = int(N), len(moduli)
        arr = (C.c_uint64 * len(moduli))(*[int(q) for q in moduli])
        check(lib.pf_multi_ring(self._h, int(N), len(moduli), arr), "pf_multi_ring")

    def flat(self, xb_host):
        xb = np.ascontiguousarray(xb_host, dtype=np.float32)
        self.nb, self.d = xb.shape
        check(lib.pf_multi_flat(self._h, xb.ctypes.data_as(C.c_void_p), xb.shape[0], xb.shape[1]), "pf_multi_flat")

    def reserve(self, nq_local_max, k_max):
        check(lib.pf_multi_reserve(self._h, int(nq_local_max), int(k_max)), "pf_multi_reserve")

    def stream(self, rank):
        """member `rank`'s hipStream_t as a torch ExternalStream (for events and for ordering caller-side work)"""
        s = C.c_void_p()
        check(lib.pf_multi_member(self._h, rank, None, C.byref(s), None, None), "pf_multi_member")
        return torch.cuda.ExternalStream(s.value, device=torch.device("cuda", self.devices[rank]))

    def _ptrs(self, tensors, dtype, name):
        if len(tensors) != self.G:
            raise ValueError(f"{name}: need one tensor per member")
        return (C.c_void_p * self.G)(*[_req(t, dtype, self.devices[r], f"{name}[{r}]").value for r, t in enumerate(tensors)])

    def flat_search(self, xq, k, gathered):
        """xq[r]: [nq_local, d] float32 on member r's device; gathered[r]: int32 [G * nq_local, k, 3] on member r's device."""
        nq_local = xq[0].shape[0]
        if any(t.shape != (nq_local, self.d) for t in xq) or any(t.numel() != self.G * nq_local * k * 3 for t in gathered):
            raise ValueError("flat_search: shards must be [nq_local, d] and gathered buffers [G * nq_local, k, 3]")
        check(lib.pf_multi_flat_search(self._h, self._ptrs(xq, torch.float32, "xq"), nq_local, k, self._ptrs(gathered, torch.int32, "gathered")),
              "pf_multi_flat_search")

    def ct_pt_mul(self, ct, pt, out, flags=0):
        per_ct = 2 * self.L * self.N
        B_local = ct[0].numel() // per_ct
        pt_count = pt[0].numel() // (self.L * self.N)
        if pt_count not in (1, B_local) or any(t.numel() != B_local * per_ct for t in list(ct) + list(out)):
            raise ValueError("ct_pt_mul: every member needs [B_local, 2, L, N] ct / out and 1 or B_local plaintexts")
        check(lib.pf_multi_ct_pt_mul(self._h, self._ptrs(ct, torch.int64, "ct"), self._ptrs(pt, torch.int64, "pt"), pt_count,
                                     self._ptrs(out, torch.int64, "out"), B_local, int(flags)), "pf_multi_ct_pt_mul")

    def synchronize(self):
        check(lib.pf_multi_synchronize(self._h), "pf_multi_synchronize")

    def flat_search_host(self, xq_host, k):
        """host [nq, d] float32 -> host (D [nq, k] float32, I [nq, k] int64): shard, upload, search, ONE all-gather, download."""
        xq = np.ascontiguousarray(xq_host, dtype=np.float32)
        nq = xq.shape[0]
        D = np.empty((nq, k), np.float32)
        I = np.empty((nq, k), np.int64)
        check(lib.pf_multi_flat_search_host(self._h, xq.ctypes.data_as(C.c_void_p), nq, k, D.ctypes.data_as(C.c_void_p),
                                            I.ctypes.data_as(C.c_void_p)), "pf_multi_flat_search_host")
        return D, I


class IvfPq:
    """faiss::IndexIVFPQ (8-bit sub-quantizers, by_residual) with trained tables: the index behind Server::coarseSearch
    (reference src/server/server_lib.cpp:33-36,111-138)."""

    def __init__(self, centroids, codebooks, device="cuda:0"):
        self.device_index = _dev_index(device)
        self.device = torch.device("cuda", self.device_index)
        centroids = np.ascontiguousarray(centroids, np.float32)
        codebooks = np.ascontiguousarray(codebooks, np.float32)
        self.nlist, self.d = centroids.shape
        self.M = codebooks.shape[0]
        if codebooks.shape != (self.M, 256, self.d // self.M):
            raise ValueError("codebooks must be [M, 256, d/M]")
        h = C.c_void_p()
        check(lib.pf_ivfpq_create(C.byref(h), self.device_index, self.d, self.nlist, self.M, centroids.ctypes.data_as(C.c_void_p),
                                  codebooks.ctypes.data_as(C.c_void_p)), "pf_ivfpq_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.pf_ivfpq_destroy(self._h)
            self._h = None

    __del__ = close

    def add_encoded(self, list_ids, codes, ids):
        list_ids = np.ascontiguousarray(list_ids, np.int64)
        codes = np.ascontiguousarray(codes, np.uint8)
        ids = np.ascontiguousarray(ids, np.int64)
        if codes.shape != (list_ids.size, self.M) or ids.size != list_ids.size:
            raise ValueError("codes must be [n, M]; ids and list_ids [n]")
        check(lib.pf_ivfpq_add_encoded(self._h, list_ids.size, list_ids.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p),
                                       ids.ctypes.data_as(C.c_void_p)), "pf_ivfpq_add_encoded")

    def list_sizes(self):
        sizes = np.zeros(self.nlist, np.uint64)
        check(lib.pf_ivfpq_info(self._h, None, None, None, None, sizes.ctypes.data_as(C.c_void_p)), "pf_ivfpq_info")
        return sizes

    def search_lists(self, xq, probe):
        """xq [nq,d] float32 on the device, probe [nq,nprobe] int64 numpy (host).  Returns (D, I, list_sizes)."""
        pq = _req(xq, torch.float32, self.device_index, "xq")
        probe = np.ascontiguousarray(probe, np.int64)
        nq, nprobe = probe.shape
        sizes = np.asarray(self.list_sizes(), dtype=np.uint64)
        flat = probe.ravel()
        cap = int(sizes[flat[(flat >= 0) & (flat < self.nlist)]].sum())
        D = torch.empty(max(cap, 1), dtype=torch.float32, device=self.device)
        I = torch.empty(max(cap, 1), dtype=torch.int64, device=self.device)
        per_q = np.zeros(nq, np.uint64)
        check(lib.pf_ivfpq_search_lists(self._h, pq, probe.ctypes.data_as(C.c_void_p), nq, nprobe, C.c_void_p(D.data_ptr()),
                                        C.c_void_p(I.data_ptr()), cap, per_q.ctypes.data_as(C.c_void_p), _stream(self.device)),
              "pf_ivfpq_search_lists")
        return D[:cap], I[:cap], per_q
