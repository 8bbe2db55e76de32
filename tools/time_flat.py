#!/usr/bin/env python3
"""Times FlatL2.search on BASELINE config-3 data (8-bit-valued, 1M x 128, 1024 queries, k = 200).  usage: tools/time_flat.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
xb = torch.randint(0, 256, (1_000_000, 128), generator=g, device=dev, dtype=torch.int32).float()
xq = torch.randint(0, 256, (1024, 128), generator=g, device=dev, dtype=torch.int32).float()
idx = pf.FlatL2(xb, dev)
idx.reserve(1024, 200)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for mode in (1, 0):
    idx.exact16(mode)
    idx.search(xq, 200)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        idx.search(xq, 200)
    b.record(); torch.cuda.synchronize()
    print("lib=%s exact16=%d active=%s  %.4f ms per search" % (os.path.basename(pf.LIB_PATH), mode, idx.exact16(), a.elapsed_time(b) / reps))
