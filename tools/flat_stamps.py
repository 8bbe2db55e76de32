#!/usr/bin/env python3
"""Phase timeline of the bf16 tile walk (experiments): needs a library built with -DPF_FLAT_STAMPS, named by PREFHETCH_HIP_LIB.
Prints, per phase, the mean / max shader cycles over the stamped waves and tiles of the 524288-column chunk."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402
from prefhetch_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
if os.environ.get("PF_RK_LAW") == "gauss":                          # N(0,1): the tiles as a conservative filter, survivors by the fp32 chain
    xb = torch.randn((1_000_000, 128), generator=g, device=dev)
    xq = torch.randn((1024, 128), generator=g, device=dev)
else:
    xb = torch.randint(0, 256, (1_000_000, 128), generator=g, device=dev, dtype=torch.int32).float()
    xq = torch.randint(0, 256, (1024, 128), generator=g, device=dev, dtype=torch.int32).float()
idx = pf.FlatL2(xb, dev)
for _ in range(3):
    idx.search(xq, 200)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
WGS, TILES, K = 32, 16, 6
buf = np.zeros(WGS * 4 * TILES * K, dtype=np.uint64)
rc = lib.pf_flat_debug_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
assert rc == 0, rc
s = buf.reshape(WGS, 4, TILES, K).astype(np.int64)
ok = s[..., 0] > 0
nt = int(ok[0, 0].sum())
print("tiles stamped per workgroup:", nt)
s = s[:, :, :nt]
names = ["LDS-DMA requests for tile t+1", "addresses, zero acc", "fragment reads + matrix instructions", "sign sweep", "vmcnt(0) + barrier"]
for k in range(K - 1):
    d = s[..., k + 1] - s[..., k]
    print("%-40s mean %8.0f  p50 %8.0f  max %8.0f" % (names[k], d.mean(), np.median(d), d.max()))
gap = s[:, :, 1:, 0] - s[:, :, :-1, K - 1]
print("%-40s mean %8.0f  max %8.0f" % ("end of barrier -> next tile", gap.mean(), gap.max()))
per = s[:, :, 1:, 0] - s[:, :, :-1, 0]
print("%-40s mean %8.0f  p50 %8.0f max %8.0f" % ("tile period", per.mean(), np.median(per), per.max()))
print("per tile index (mean over waves): period", np.round(per.mean(axis=(0, 1))).astype(int).tolist())
for k in range(K - 1):
    print("  phase %d by tile:" % k, np.round((s[..., k + 1] - s[..., k]).mean(axis=(0, 1))).astype(int).tolist())
print("by wave (mean): ", [[int((s[:, w, :, k + 1] - s[:, w, :, k]).mean()) for k in range(K - 1)] for w in range(4)])

fb = np.zeros(WGS * 4 * 8 * 8, dtype=np.uint64)
if hasattr(lib, "pf_flat_debug_flush_stamps") and lib.pf_flat_debug_flush_stamps(fb.ctypes.data_as(C.c_void_p), C.c_size_t(fb.size)) == 0:
    fall = fb.reshape(WGS * 4, 8, 8).astype(np.int64)
    fn = ["entry -> first barrier (__syncthreads_or)", "slot reservation + decode of the verdict words", "barrier", "list -> row loads issued", "row reservations (global atomics) + barrier",
          "dot products, keys written", "last barrier, second __syncthreads_or, exit"]
    for no in (0, 1, 2):
        f = fall[:, no]
        f = f[f[:, 0] > 0]
        if not len(f):
            continue
        print("flush %d of the walk, stamped waves (cycles):" % no)
        for k in range(7):
            d = f[:, k + 1] - f[:, k]
            print("  %-52s mean %7.0f  p50 %7.0f  max %7.0f" % (fn[k], d.mean(), np.median(d), d.max()))
        print("  %-52s mean %7.0f" % ("whole flush", (f[:, 7] - f[:, 0]).mean()))
