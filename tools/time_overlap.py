#!/usr/bin/env python3
"""Experiment: the two stages of a config-3 step (pre-filter; fused ct x pt) issued on ONE stream against TWO streams (batch i's products
concurrent with batch i+1's pre-filter).  usage: tools/time_overlap.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402
dev = torch.device("cuda", 0)
N, B = 8192, 1024
MODULI = [0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001]   # SEAL BFVDefault(8192) data primes
g = torch.Generator(device=dev).manual_seed(1)
xb = torch.randint(0, 256, (1_000_000, 128), generator=g, device=dev, dtype=torch.int32).float()
xq = torch.randint(0, 256, (B, 128), generator=g, device=dev, dtype=torch.int32).float()
idx = pf.FlatL2(xb, dev)
idx.reserve(B, 200)
ctx = pf.RnsContext(N, MODULI, dev)
ct = torch.stack([torch.randint(0, q, (B, 2, N), generator=g, device=dev, dtype=torch.int64) for q in MODULI], dim=2).contiguous()
pt = torch.stack([torch.randint(0, q, (B, N), generator=g, device=dev, dtype=torch.int64) for q in MODULI], dim=1).contiguous()
out = torch.empty_like(ct)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
Dq = torch.empty((B, 200), device=dev); Iq = torch.empty((B, 200), device=dev, dtype=torch.int64)

def serial():
    for _ in range(reps):
        idx.search(xq, 200)
        ctx.ct_pt_mul(ct, pt, out=out)

sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def overlapped():
    for _ in range(reps):
        with torch.cuda.stream(sa):
            idx.search(xq, 200)
        with torch.cuda.stream(sb):
            ctx.ct_pt_mul(ct, pt, out=out)

for name, fn in (("one stream", serial), ("two streams", overlapped), ("one stream", serial), ("two streams", overlapped)):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    fn()
    sa.synchronize(); sb.synchronize()
    b.record(); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("%-12s %.4f ms per step (wall clock, %d steps)" % (name, (t1 - t0) * 1e3 / reps, reps), flush=True)
