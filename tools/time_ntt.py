#!/usr/bin/env python3
"""Times forward / inverse NTT and fused ct x pt for one ring degree on 55-bit primes (the 64-bit integer butterflies).
usage: python3 tools/time_ntt.py <logn> <limb-polys> [reps] [limbs]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402

logn, n, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 10
N = 1 << logn
ALLQ = [0x7FFFFFFFE90001, 0x7FFFFFFFBF0001, 0x7FFFFFFFBD0001, 0x7FFFFFFFBA0001, 0x7FFFFFFFAA0001, 0x7FFFFFFFA50001,
        0x7FFFFFFF9F0001, 0x7FFFFFFF7E0001, 0x7FFFFFFF770001, 0x7FFFFFFF380001, 0x7FFFFFFF330001, 0x7FFFFFFF2D0001,
        0x7FFFFFFF170001, 0x7FFFFFFF150001, 0x7FFFFFFEF00001, 0xFFFFFFFFF70001]      # all = 1 mod 2^16
Q = ALLQ[:int(sys.argv[4])] if len(sys.argv) > 4 else ALLQ[:1]
dev = torch.device("cuda", 0)
ctx = pf.RnsContext(N, Q, dev)
g = torch.Generator(device=dev).manual_seed(1)
n = n // (2 * len(Q)) * 2 * len(Q)
x = torch.randint(0, min(Q), (n, N), generator=g, device=dev, dtype=torch.int64)
pt = torch.randint(0, min(Q), (n // 2, N), generator=g, device=dev, dtype=torch.int64)
out = torch.empty_like(x)


def timed(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for name, fn, bytes_ in (("ntt_fwd", lambda: ctx.ntt_forward_(x), 16 * N * n), ("ntt_inv", lambda: ctx.ntt_inverse_(x), 16 * N * n),
                         ("ctpt", lambda: ctx.ct_pt_mul(x, pt, out=out), 20 * N * n)):
    ms = timed(fn)
    print("%s lib=%s logn=%d limbs=%d polys=%d %-8s %.4f ms  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
        ctx.info()["arith_path"][0], os.path.basename(pf.LIB_PATH), logn, len(Q), n, name, ms, bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 80))
