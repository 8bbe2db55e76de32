#!/usr/bin/env python3
"""Would running passes of consecutive key-switch rounds on two streams pay?  Measured without changing the product: the 256 polynomials of
BASELINE config 5 are switched as ONE call on one stream, and as TWO calls of 128 on two streams through two contexts (their own workspaces), so
that one half's pass A runs beside the other half's pass B wherever the hardware lets it.  Both figures are wall time of the whole batch.
usage: tools/time_ks_overlap.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402
DQ = [0x7FFFFFFFE90001, 0x7FFFFFFFBF0001, 0x7FFFFFFFBD0001, 0x7FFFFFFFBA0001, 0x7FFFFFFFAA0001, 0x7FFFFFFFA50001, 0x7FFFFFFF9F0001, 0x7FFFFFFF7E0001,
      0x7FFFFFFF770001, 0x7FFFFFFF380001, 0x7FFFFFFF330001, 0x7FFFFFFF2D0001, 0x7FFFFFFF170001, 0x7FFFFFFF150001, 0x7FFFFFFEF00001]
KQ = DQ + [0xFFFFFFFFF70001]
N, B, D = 32768, 256, len(DQ)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
target = torch.stack([torch.randint(0, q, (B, N), generator=g, device=dev, dtype=torch.int64) for q in DQ], dim=1).contiguous()
ksk = torch.stack([torch.stack([torch.stack([torch.randint(0, q, (N,), generator=g, device=dev, dtype=torch.int64) for q in KQ]) for _ in range(2)]) for _ in range(D)]).contiguous()
ct = torch.stack([torch.randint(0, q, (B, 2, N), generator=g, device=dev, dtype=torch.int64) for q in DQ], dim=2).contiguous()
ctxs = [pf.RnsContext(N, KQ, dev) for _ in range(2)]
ctxs[0].key_switch_reserve(B)
ctxs[1].key_switch_reserve(B // 2)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
h = B // 2


def one():
    ctxs[0].key_switch_(target, ksk, ct)


def two():
    cur = torch.cuda.current_stream(dev)
    for i, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            ctxs[i].key_switch_(target[i * h:(i + 1) * h], ksk, ct[i * h:(i + 1) * h])
    for s in streams:
        cur.wait_stream(s)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for _ in range(2):
    print("one call of 256 on one stream:            %.2f ms per 256" % timed(one), flush=True)
    print("two calls of 128 on two streams, at once: %.2f ms per 256" % timed(two), flush=True)
