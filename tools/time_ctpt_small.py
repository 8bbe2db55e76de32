#!/usr/bin/env python3
"""ct x pt at the small shapes (launches that do not fill the device twice): ms per launch for a range of batches at N = 4096 / 8192.
usage: python3 tools/time_ctpt_small.py [PF_CTPT_PAIR_OFF=1 in the environment for the one-polynomial-per-workgroup kernel]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402

Q = {4096: [0xFFFFEE001, 0xFFFFC4001], 8192: [0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001]}
dev = torch.device("cuda", 0)
for N, batches in ((4096, (64, 128, 256, 383, 512, 1024)), (8192, (32, 64, 128, 191))):
    qs = Q[N]
    ctx = pf.RnsContext(N, qs, dev)
    for B in batches:
        g = torch.Generator(device=dev).manual_seed(B)
        ct = torch.stack([torch.randint(0, q, (B, 2, N), generator=g, device=dev, dtype=torch.int64) for q in qs], dim=2).contiguous()
        pt = torch.stack([torch.randint(0, q, (B, N), generator=g, device=dev, dtype=torch.int64) for q in qs], dim=1).contiguous()
        res = torch.empty_like(ct)
        for _ in range(5):
            ctx.ct_pt_mul(ct, pt, out=res)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ctx.ct_pt_mul(ct, pt, out=res)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50)
        gb = 40 * len(qs) * N * B / best / 1e6
        print(f"N={N} L={len(qs)} B={B}: {best*1e3:.1f} us  {gb:.0f} GB/s ({gb/8000:.3f} of 8 TB/s)", flush=True)
