#!/usr/bin/env python3
"""Throughput of the polynomial kernels on the BASELINE.json ring-dim / limb / batch shapes (configs 1, 2, 3, 5), with the
CPU oracle timed beside them on a bounded sample.  Writes one JSON document to stdout.
usage: python3 tools/sweep_shapes.py > gpurun_out/sweep.json"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402
import prefhetch_amd as pf  # noqa: E402

SHAPES = [  # (config, N, number of data limbs, batch)
    ("config 1: N=1024, 1 prime, batch 1", 1024, 1, 1),
    ("config 2: N=4096, 2 limbs, batch 256", 4096, 2, 256),
    ("config 3: N=8192, 4 limbs, batch 1024", 8192, 4, 1024),
    ("config 5: N=32768, 15 limbs, batch 256", 32768, 15, 256),
]
dev = torch.device("cuda", 0)
threads = min(16, os.cpu_count() or 1)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = []
for name, N, L, B in SHAPES:
    qs = oracle.BFV_DEFAULT[N][:L]
    g = torch.Generator(device=dev).manual_seed(N + B)
    ct = torch.stack([torch.randint(0, q, (B, 2, N), generator=g, device=dev, dtype=torch.int64) for q in qs], dim=2).contiguous()
    pt = torch.stack([torch.randint(0, q, (B, N), generator=g, device=dev, dtype=torch.int64) for q in qs], dim=1).contiguous()
    res = torch.empty_like(ct)
    ctx = pf.RnsContext(N, qs, dev)
    reps = 20 if N < 32768 else 5
    rec = {"shape": name, "N": N, "limbs": L, "batch": B, "arith_path": {0: "exact-FP64", 1: "u64 Harvey/Shoup", 2: "u64 lazy (q < 2^56)"}[ctx.info()["arith_path"][0]]}
    for kname, fn, bytes_per in (
        ("ct_x_pt_fused", lambda: ctx.ct_pt_mul(ct, pt, out=res), 40 * L * N * B),
        ("ntt_forward", lambda: ctx.ntt_forward_(res), 16 * N * 2 * L * B),
        ("ntt_inverse", lambda: ctx.ntt_inverse_(res), 16 * N * 2 * L * B),
        ("dyadic_mul", lambda: ctx.dyadic_mul(ct, res, out=res), 24 * N * 2 * L * B),
        ("poly_add", lambda: ctx.add(ct, res, out=res), 24 * N * 2 * L * B),
    ):
        ms = timed(fn, reps)
        rec[kname] = {"ms": ms, "algorithmic_GBps": bytes_per / ms / 1e6, "frac_of_8TBps": bytes_per / ms / 1e6 / 8000.0}
    rec["encrypted_queries_per_s_gpu"] = B / (rec["ct_x_pt_fused"]["ms"] * 1e-3)
    # CPU oracle on a bounded sample
    nb = min(B, max(1, 4096 // (L * (N // 1024))))
    o = oracle.Oracle(N, qs)
    hct, hpt = pf.to_host_u64(ct[:nb]), pf.to_host_u64(pt[:nb])
    o.ct_pt_mul(hct[:1], hpt[:1], threads=threads)
    t0 = time.perf_counter()
    o.ct_pt_mul(hct, hpt, threads=threads)
    dt = time.perf_counter() - t0
    rec["encrypted_queries_per_s_cpu_oracle"] = nb / dt
    rec["cpu_sample"] = f"{nb} ct x pt, OpenMP {threads} threads"
    rec["speedup"] = rec["encrypted_queries_per_s_gpu"] / rec["encrypted_queries_per_s_cpu_oracle"]
    out.append(rec)
    del ct, pt, res, ctx
    torch.cuda.empty_cache()
print(json.dumps({"device": torch.cuda.get_device_name(0), "shapes": out}, indent=1))
