#!/usr/bin/env python3
"""Pre-filter at row lengths above 256 (1024 queries, k = 200): the slab tiles over bf16 images + fp32 chain against fp32 operands throughout.
usage: tools/time_flat_wide.py [d ...]   (nb = 1M for d <= 512, 500k beyond: the fp32 matrix is 2 GB at 1M x 512)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402
dev = torch.device("cuda", 0)
dims = [int(a) for a in sys.argv[1:]] or [384, 512, 768, 1024]
for d in dims:
    nb = 1_000_000 if d <= 512 else 500_000
    for law in ("uint8", "gauss"):
        g = torch.Generator(device=dev).manual_seed(d)
        if law == "uint8":
            xb = torch.randint(0, 256, (nb, d), generator=g, device=dev, dtype=torch.int32).float()
            xq = torch.randint(0, 256, (1024, d), generator=g, device=dev, dtype=torch.int32).float()
        else:
            xb = torch.randn((nb, d), generator=g, device=dev)
            xq = torch.randn((1024, d), generator=g, device=dev)
        idx = pf.FlatL2(xb, dev)
        idx.reserve(1024, 200)
        out = []
        ref = None
        for mode in (1, 0):
            idx.operands16(mode)
            D, I = idx.search(xq, 200)
            torch.cuda.synchronize()
            if ref is None:
                ref = (D.clone(), I.clone())
            else:
                assert (I == ref[1]).all() and (D.view(torch.int32) == ref[0].view(torch.int32)).all(), "paths differ"
            reps = 5 if mode else 2
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                idx.search(xq, 200)
            b.record(); torch.cuda.synchronize()
            out.append("operands16=%d %.3f ms" % (idx.operands16(), a.elapsed_time(b) / reps))
        print("d=%4d nb=%d %-6s %s  (bit-identical)" % (d, nb, law, "  |  ".join(out)), flush=True)
        del idx, xb, xq
        torch.cuda.empty_cache()
