#!/usr/bin/env python3
"""Times FlatL2.search over 1M rows x d for several row lengths, 8-bit-valued and N(0,1) data, bf16 tiles against fp32 operands.
usage: tools/time_flat_d.py [d ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402
dev = torch.device("cuda", 0)
dims = [int(a) for a in sys.argv[1:]] or [128, 144, 192, 256]
for d in dims:
    for law in ("uint8", "gauss"):
        g = torch.Generator(device=dev).manual_seed(1)
        if law == "uint8":
            xb = torch.randint(0, 256, (1_000_000, d), generator=g, device=dev, dtype=torch.int32).float()
            xq = torch.randint(0, 256, (1024, d), generator=g, device=dev, dtype=torch.int32).float()
        else:
            xb = torch.randn((1_000_000, d), generator=g, device=dev)
            xq = torch.randn((1024, d), generator=g, device=dev)
        idx = pf.FlatL2(xb, dev)
        idx.reserve(1024, 200)
        out = []
        for mode in (1, 0):
            active = idx.operands16(mode)
            idx.search(xq, 200)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                idx.search(xq, 200)
            b.record(); torch.cuda.synchronize()
            out.append("operands16=%d %.4f ms" % (active, a.elapsed_time(b) / 10))
        print("d=%3d %-5s  %s" % (d, law, "  |  ".join(out)), flush=True)
        del idx, xb, xq
