#!/usr/bin/env python3
"""Config 5 (N = 32768, 15 limbs, batch 256): forward / inverse transform and fused ct x pt.
usage: tools/time_ns.py [reps]   (PF_NS_SPLIT=0 / PF_NS_ROUND=<polys> are read when the context is created)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402
QS = [0x7FFFFFFFE90001, 0x7FFFFFFFBF0001, 0x7FFFFFFFBD0001, 0x7FFFFFFFBA0001, 0x7FFFFFFFAA0001, 0x7FFFFFFFA50001, 0x7FFFFFFF9F0001, 0x7FFFFFFF7E0001,
      0x7FFFFFFF770001, 0x7FFFFFFF380001, 0x7FFFFFFF330001, 0x7FFFFFFF2D0001, 0x7FFFFFFF170001, 0x7FFFFFFF150001, 0x7FFFFFFEF00001]
N, B = 32768, int(os.environ.get("PF_NS_BATCH", "256"))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
ct = torch.stack([torch.randint(0, q, (B, 2, N), generator=g, device=dev, dtype=torch.int64) for q in QS], dim=2).contiguous()
pt = torch.stack([torch.randint(0, q, (B, N), generator=g, device=dev, dtype=torch.int64) for q in QS], dim=1).contiguous()
res = torch.empty_like(ct)
ctx = pf.RnsContext(N, QS, dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
def timed(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
L = len(QS)
for name, fn, nbytes in (("ct_x_pt", lambda: ctx.ct_pt_mul(ct, pt, out=res), 40 * L * N * B), ("forward", lambda: ctx.ntt_forward_(res), 32 * L * N * B),
                         ("inverse", lambda: ctx.ntt_inverse_(res), 32 * L * N * B)):
    ms = timed(fn)
    print("%-8s %.3f ms  %.3f of 8 TB/s   split=%s round=%s" % (name, ms, nbytes / (ms * 1e-3) / 8e12, os.environ.get("PF_NS_SPLIT", "1"), os.environ.get("PF_NS_ROUND", "960 (default)")))
