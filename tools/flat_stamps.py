#!/usr/bin/env python3
"""Phase timeline of the pre-filter's tile walk -- the streamed int8 walk on 8-bit data, the bf16 tiles otherwise or with PF_FLAT_I8_OLD=1 -- (experiments): needs a library built with -DPF_FLAT_STAMPS, named by PREFHETCH_HIP_LIB.
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
walk8 = idx.operands8() == 1 and not os.environ.get("PF_FLAT_I8_OLD")
if walk8:
    # the streamed 8-bit walk (flat_tile8.hpp): every SECOND step of a wave is stamped, a step = 128 queries x 32 columns = 16 x v_mfma_i32_16x16x64_i8 at D = 128
    print("walk: streamed int8 (one wave = 128 queries x 32 columns per step; stamped: every second step)")
    names = ["(nothing)", "half 0: 8 matrix instr. of block 0 + sweep of the step before's block 1 + requests for the next step",
             "half 1: 8 matrix instr. of block 1 + sweep of block 0", "ring append (ballot, LDS write)", "ring-full check / flush"]
else:
    print("walk: bf16 tiles (128 x 128 per workgroup; 36 x v_mfma_f32_32x32x16_bf16 per wave and tile at d = 128)")
    names = ["LDS-DMA requests for tile t+1", "addresses, zero acc", "fragment reads + matrix instructions", "sign sweep", "vmcnt(0) + barrier"]
for k in range(K - 1):
    d = s[..., k + 1] - s[..., k]
    print("%-40s mean %8.0f  p50 %8.0f  max %8.0f" % (names[k][:110], d.mean(), np.median(d), d.max()))
gap = s[:, :, 1:, 0] - s[:, :, :-1, K - 1]
print("%-40s mean %8.0f  max %8.0f" % ("last stamp -> next stamped %s" % ("step (= the unstamped odd step)" if walk8 else "tile"), gap.mean(), gap.max()))
per = s[:, :, 1:, 0] - s[:, :, :-1, 0]
print("%-40s mean %8.0f  p50 %8.0f max %8.0f" % ("period (%s)" % ("TWO steps" if walk8 else "one tile"), per.mean(), np.median(per), per.max()))
print("per stamped index (mean over waves): period", np.round(per.mean(axis=(0, 1))).astype(int).tolist())
for k in range(K - 1):
    print("  phase %d by index:" % k, np.round((s[..., k + 1] - s[..., k]).mean(axis=(0, 1))).astype(int).tolist())
print("by wave (mean): ", [[int((s[:, w, :, k + 1] - s[:, w, :, k]).mean()) for k in range(K - 1)] for w in range(4)])

fb = np.zeros(WGS * 4 * 8 * 8, dtype=np.uint64)
if hasattr(lib, "pf_flat_debug_flush_stamps") and lib.pf_flat_debug_flush_stamps(fb.ctypes.data_as(C.c_void_p), C.c_size_t(fb.size)) == 0:
    fall = fb.reshape(WGS * 4, 8, 8).astype(np.int64)
    if walk8:
        # stamps 0 entry | 1 first decode round done | 2 first rows + row reservations requested | 3 exit; 6: records in the ring, 7: entries of the first list
        fn, last = ["entry -> ring decoded into the list (first round)", "first rows + reservation atomics requested", "rows back, dot products, keys written (all rounds)"], 3
    else:
        fn, last = ["entry -> first barrier (__syncthreads_or)", "slot reservation + decode of the verdict words", "barrier", "list -> row loads issued",
                    "row reservations (global atomics) + barrier", "dot products, keys written"], 6
    for no in (0, 1, 2):
        f = fall[:, no]
        f = f[(f[:, 0] > 0) & (f[:, last] > 0)]
        if not len(f):
            continue
        print("flush %d of the walk, %d stamped waves (cycles):" % (no, len(f)))
        for k in range(last):
            d = f[:, k + 1] - f[:, k]
            print("  %-52s mean %7.0f  p50 %7.0f  max %7.0f" % (fn[k], d.mean(), np.median(d), d.max()))
        print("  %-52s mean %7.0f" % ("whole flush", (f[:, last] - f[:, 0]).mean()))
        if walk8:
            print("  %-52s mean %7.1f  max %d" % ("records in the ring", f[:, 6].mean(), f[:, 6].max()))
            print("  %-52s mean %7.1f  max %d" % ("survivors in the first list", f[:, 7].mean(), f[:, 7].max()))
