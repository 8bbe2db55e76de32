#!/usr/bin/env python3
"""Times FlatL2.search on N(0,1) data (1M x 128, 1024 queries, k = 200): bf16 tiles as a filter vs fp32 operands."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
xb = torch.randn((1_000_000, 128), generator=g, device=dev)
xq = torch.randn((1024, 128), generator=g, device=dev)
idx = pf.FlatL2(xb, dev)
idx.reserve(1024, 200)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
res = {}
for mode in (1, 0):
    idx.operands16(mode)
    res[mode] = idx.search(xq, 200)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        idx.search(xq, 200)
    b.record(); torch.cuda.synchronize()
    print("gaussian: operands16=%d  %.4f ms per search" % (idx.operands16(), a.elapsed_time(b) / reps))
print("bit-identical:", bool((res[1][1] == res[0][1]).all() and (res[1][0].view(torch.int32) == res[0][0].view(torch.int32)).all()))
