"""prefhetch_amd -- MI355X (gfx950) implementation of the PreFHEtch server-side encrypted-query hot path.

The product is libprefhetch_hip.so (C ABI: include/prefhetch_hip.h) and the C++ `Server` class above
it (include/server/server_lib.h).  This package is the Python-side harness over the same C ABI;
importing it loads the HIP library and fails loudly if that is missing -- there is no CPU fallback.
"""
from ._lib import LIB_PATH, PfError, SYMBOLS, lib  # noqa: F401
from .api import ACCUMULATE, IN_NTT, OUT_NTT, DeviceGroup, FlatL2, IvfPq, RnsContext, to_device_u64, to_host_u64  # noqa: F401
