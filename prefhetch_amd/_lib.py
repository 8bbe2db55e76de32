"""ctypes binding of libprefhetch_hip.so (the C ABI declared in include/prefhetch_hip.h).

There is no fallback: if the HIP library is missing or does not load, importing fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PREFHETCH_HIP_LIB overrides the library path (kernel experiments); it must still be a HIP build of the C ABI
LIB_PATH = os.environ.get("PREFHETCH_HIP_LIB") or os.path.join(_HERE, "lib", "libprefhetch_hip.so")

# every symbol include/prefhetch_hip.h declares
SYMBOLS = [
    "pf_status_str", "pf_last_error", "pf_build_flags", "pf_device_count",
    "pf_malloc", "pf_free", "pf_memcpy_h2d", "pf_memcpy_d2h", "pf_memcpy_d2d", "pf_stream_synchronize",
    "pf_ctx_create", "pf_ctx_destroy", "pf_ctx_info", "pf_ctx_force_u64",
    "pf_ntt_forward", "pf_ntt_inverse", "pf_ntt_forward_to", "pf_ntt_inverse_to", "pf_dyadic_mul", "pf_poly_add", "pf_poly_sub", "pf_poly_negate",
    "pf_ct_pt_mul", "pf_ct_pt_mul_fanout", "pf_apply_galois", "pf_apply_galois_ct", "pf_poly_mul_monomial", "pf_poly_addsub_monomial", "pf_key_switch", "pf_key_switch_reserve", "pf_pack_rows", "pf_pack_rows_ntt", "pf_ct_rows_mul", "pf_ct_pt_dot",
    "pf_flat_create", "pf_flat_destroy", "pf_flat_info", "pf_flat_search", "pf_l2_gathered", "pf_gather_rows",
    "pf_flat_reserve", "pf_flat_search_packed", "pf_flat_exact16", "pf_flat_operands8",
    "pf_multi_create", "pf_multi_destroy", "pf_multi_info", "pf_multi_ring", "pf_multi_flat", "pf_multi_reserve", "pf_multi_member",
    "pf_multi_flat_search", "pf_multi_ct_pt_mul", "pf_multi_synchronize", "pf_multi_flat_search_host",
    "pf_ivfpq_create", "pf_ivfpq_destroy", "pf_ivfpq_add_encoded", "pf_ivfpq_info", "pf_ivfpq_get_list", "pf_ivfpq_search_lists",
]


class PfError(RuntimeError):
    def __init__(self, status, where, detail):
        super().__init__(f"{where}: status {status} ({detail})")
        self.status = status


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"prefhetch_amd: {LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C prefhetch_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    if not hasattr(lib, "pf_build_flags") and os.environ.get("PREFHETCH_HIP_LIB"):
        # an older library named on purpose (A/B timing against an earlier round): it predates the build-flag record
        return _bind(lib, skip=("pf_build_flags",))
    lib.pf_build_flags.restype = C.c_char_p
    lib.pf_build_flags.argtypes = []
    flags = (lib.pf_build_flags() or b"").decode()
    if flags.startswith("experiment:") and not os.environ.get("PREFHETCH_HIP_LIB"):
        # a timing-only ablation / debug build (pf_common.hpp) may only be loaded when it was asked for by path
        raise ImportError(f"prefhetch_amd: {LIB_PATH} is an experiment build ({flags}); it returns wrong results. "
                          "Rebuild with a plain `make -C prefhetch_amd/csrc`, or name it in PREFHETCH_HIP_LIB on purpose.")
    return _bind(lib)


def _bind(lib, skip=()):
    vp, sz, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
    lib.pf_status_str.restype = C.c_char_p
    lib.pf_status_str.argtypes = [C.c_int32]
    lib.pf_last_error.restype = C.c_char_p
    lib.pf_last_error.argtypes = []
    lib.pf_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.pf_malloc.argtypes = [i32, C.POINTER(vp), sz]
    lib.pf_free.argtypes = [i32, vp]
    lib.pf_memcpy_h2d.argtypes = [i32, vp, vp, sz, vp]
    lib.pf_memcpy_d2h.argtypes = [i32, vp, vp, sz, vp]
    lib.pf_memcpy_d2d.argtypes = [i32, vp, vp, sz, vp]
    lib.pf_stream_synchronize.argtypes = [i32, vp]
    lib.pf_ctx_create.argtypes = [C.POINTER(vp), i32, u32, u32, C.POINTER(C.c_uint64)]
    lib.pf_ctx_destroy.argtypes = [vp]
    lib.pf_ctx_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    lib.pf_ctx_force_u64.argtypes = [vp, i32]
    lib.pf_ntt_forward.argtypes = [vp, vp, sz, vp]
    lib.pf_ntt_inverse.argtypes = [vp, vp, sz, vp]
    lib.pf_ntt_forward_to.argtypes = [vp, vp, vp, sz, vp]
    lib.pf_ntt_inverse_to.argtypes = [vp, vp, vp, sz, vp]
    for name in ("pf_dyadic_mul", "pf_poly_add", "pf_poly_sub"):
        getattr(lib, name).argtypes = [vp, vp, vp, vp, sz, vp]
    lib.pf_poly_negate.argtypes = [vp, vp, vp, sz, vp]
    lib.pf_ct_pt_mul.argtypes = [vp, vp, vp, sz, vp, sz, i32, vp]
    lib.pf_key_switch.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.pf_key_switch_reserve.argtypes = [vp, sz]
    lib.pf_pack_rows.argtypes = [vp, vp, vp, sz, u32, vp, vp]
    lib.pf_apply_galois.argtypes = [vp, vp, vp, sz, u32, vp]
    lib.pf_apply_galois_ct.argtypes = [vp, vp, vp, vp, sz, u32, vp]
    lib.pf_poly_mul_monomial.argtypes = [vp, vp, vp, sz, u32, vp]
    lib.pf_poly_addsub_monomial.argtypes = [vp, vp, vp, vp, vp, sz, u32, vp]
    lib.pf_pack_rows_ntt.argtypes = [vp, vp, vp, sz, u32, vp, vp]
    lib.pf_ct_pt_mul_fanout.argtypes = [vp, vp, vp, vp, sz, u32, i32, vp]
    lib.pf_ct_rows_mul.argtypes = [vp, vp, vp, vp, sz, u32, u32, vp, vp]
    lib.pf_ct_pt_dot.argtypes = [vp, vp, sz, vp, sz, sz, vp, vp]
    lib.pf_flat_create.argtypes = [C.POINTER(vp), i32, vp, sz, u32]
    lib.pf_flat_destroy.argtypes = [vp]
    lib.pf_flat_info.argtypes = [vp, C.POINTER(sz), C.POINTER(u32)]
    lib.pf_flat_search.argtypes = [vp, vp, sz, u32, vp, vp, vp]
    lib.pf_l2_gathered.argtypes = [vp, vp, vp, sz, u32, vp, vp]
    lib.pf_gather_rows.argtypes = [vp, vp, sz, vp, vp]
    lib.pf_flat_reserve.argtypes = [vp, sz, u32]
    lib.pf_flat_exact16.argtypes = [vp, i32, C.POINTER(C.c_int)]
    lib.pf_flat_operands8.argtypes = [vp, i32, C.POINTER(C.c_int)]
    lib.pf_flat_search_packed.argtypes = [vp, vp, sz, u32, vp, vp, vp, vp]
    lib.pf_multi_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), i32, i32]
    lib.pf_multi_destroy.argtypes = [vp]
    lib.pf_multi_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.pf_multi_ring.argtypes = [vp, u32, u32, C.POINTER(C.c_uint64)]
    lib.pf_multi_flat.argtypes = [vp, vp, sz, u32]
    lib.pf_multi_reserve.argtypes = [vp, sz, u32]
    lib.pf_multi_member.argtypes = [vp, i32, C.POINTER(C.c_int), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.pf_multi_flat_search.argtypes = [vp, C.POINTER(vp), sz, u32, C.POINTER(vp)]
    lib.pf_multi_ct_pt_mul.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), sz, C.POINTER(vp), sz, i32]
    lib.pf_multi_synchronize.argtypes = [vp]
    lib.pf_multi_flat_search_host.argtypes = [vp, vp, sz, u32, vp, vp]
    lib.pf_ivfpq_create.argtypes = [C.POINTER(vp), i32, u32, u32, u32, vp, vp]
    lib.pf_ivfpq_destroy.argtypes = [vp]
    lib.pf_ivfpq_add_encoded.argtypes = [vp, sz, vp, vp, vp]
    lib.pf_ivfpq_info.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(sz), vp]
    lib.pf_ivfpq_get_list.argtypes = [vp, u32, vp, vp]
    lib.pf_ivfpq_search_lists.argtypes = [vp, vp, vp, sz, u32, vp, vp, sz, vp, vp]
    for name in SYMBOLS:
        if name in skip:
            continue
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        if name not in ("pf_status_str", "pf_last_error", "pf_build_flags"):
            fn.restype = C.c_int32
    return lib


lib = _load()


def check(status, where):
    if status != 0:
        raise PfError(status, where, (lib.pf_last_error() or b"").decode() or lib.pf_status_str(status).decode())
