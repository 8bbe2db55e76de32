#!/usr/bin/env python3
"""Flat-L2 pre-filter over nq / k / data law (MFMA-bound at nq=1024, HBM-bound at small nq).
usage: python3 tools/sweep_flat.py [nb] > gpurun_out/sweep_flat.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(20250801 + 3)
out = []
for law in ("uint8-valued", "gaussian"):
    if law == "gaussian":
        xb = torch.randn((nb, 128), generator=g, device=dev)
    else:
        xb = torch.randint(0, 256, (nb, 128), generator=g, device=dev, dtype=torch.int32).float()
    idx = pf.FlatL2(xb, dev)
    for nq in (1, 8, 32, 100, 256, 1024):
        xq = torch.randn((nq, 128), generator=g, device=dev) if law == "gaussian" else \
            torch.randint(0, 256, (nq, 128), generator=g, device=dev, dtype=torch.int32).float()
        for k in (100, 200):
            idx.search(xq, k)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                idx.search(xq, k)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            out.append({"data": law, "nq": nq, "k": k, "ms": ms, "TFLOPs": 2.0 * nq * nb * 128 / ms / 1e9,
                        "frac_of_157.3TF": 2.0 * nq * nb * 128 / ms / 1e9 / 157.3,
                        "GBps_base_matrix": nb * 512 / ms / 1e6, "frac_of_8TBps": nb * 512 / ms / 1e6 / 8000})
            print("%-12s nq=%4d k=%3d  %.3f ms  %.1f TF (%.1f%%)  %.0f GB/s (%.1f%%)" % (
                law, nq, k, ms, out[-1]["TFLOPs"], 100 * out[-1]["frac_of_157.3TF"], out[-1]["GBps_base_matrix"], 100 * out[-1]["frac_of_8TBps"]), file=sys.stderr)
    del idx, xb
print(json.dumps({"nb": nb, "runs": out}, indent=1))
