#!/usr/bin/env python3
"""Throughput of the polynomial kernels on the BASELINE.json ring-dim / limb / batch shapes (configs 1, 2, 3, 5), with the
GPU only (the CPU baseline beside the headline number is bench.py's).  Writes one JSON document to stdout.
usage: python3 tools/sweep_shapes.py > gpurun_out/sweep.json"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402

SHAPES = [  # (config, N, number of data limbs, batch)
    ("config 1: N=1024, 1 prime, batch 1", 1024, 1, 1),
    ("config 2: N=4096, 2 limbs, batch 256", 4096, 2, 256),
    ("config 3: N=8192, 4 limbs, batch 1024", 8192, 4, 1024),
    ("config 5: N=32768, 15 limbs, batch 256", 32768, 15, 256),
]
# SEAL CoeffModulus::BFVDefault(N), data primes first
BFV_DEFAULT = {
    1024: [0x7E00001],
    4096: [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001],
    8192: [0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001, 0xFFFFFEBC001],
    32768: [0x7FFFFFFFE90001, 0x7FFFFFFFBF0001, 0x7FFFFFFFBD0001, 0x7FFFFFFFBA0001, 0x7FFFFFFFAA0001, 0x7FFFFFFFA50001,
            0x7FFFFFFF9F0001, 0x7FFFFFFF7E0001, 0x7FFFFFFF770001, 0x7FFFFFFF380001, 0x7FFFFFFF330001, 0x7FFFFFFF2D0001,
            0x7FFFFFFF170001, 0x7FFFFFFF150001, 0x7FFFFFFEF00001, 0xFFFFFFFFF70001],
}
dev = torch.device("cuda", 0)


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
    qs = BFV_DEFAULT[N][:L]
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
    out.append(rec)
    del ct, pt, res, ctx
    torch.cuda.empty_cache()
print(json.dumps({"device": torch.cuda.get_device_name(0), "shapes": out}, indent=1))
