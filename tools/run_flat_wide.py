#!/usr/bin/env python3
"""A few searches at one row length above 256 (for rocprofv3): tools/run_flat_wide.py <d> <law: uint8 | gauss> [nb] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402
d, law = int(sys.argv[1]), sys.argv[2]
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(d)
if law == "uint8":
    xb = torch.randint(0, 256, (nb, d), generator=g, device=dev, dtype=torch.int32).float()
    xq = torch.randint(0, 256, (1024, d), generator=g, device=dev, dtype=torch.int32).float()
else:
    xb = torch.randn((nb, d), generator=g, device=dev)
    xq = torch.randn((1024, d), generator=g, device=dev)
idx = pf.FlatL2(xb, dev)
idx.reserve(1024, 200)
for _ in range(reps):
    idx.search(xq, 200)
torch.cuda.synchronize()
