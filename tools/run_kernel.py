#!/usr/bin/env python3
"""Runs ONE hot-path kernel a few times on synthetic BASELINE config-3 data (for rocprofv3 --pmc / --kernel-trace).
usage: python3 tools/run_kernel.py {ctpt|ntt_fwd|ntt_inv|dyadic|flat|encround|keyswitch} [reps] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402

MODULI = [0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001]
N = 8192
if os.environ.get("PF_CONFIG") == "5":       # BASELINE config 5 ring: N=32768, 15 data primes (55-bit)
    N = 32768
    MODULI = [0x7FFFFFFFE90001, 0x7FFFFFFFBF0001, 0x7FFFFFFFBD0001, 0x7FFFFFFFBA0001, 0x7FFFFFFFAA0001, 0x7FFFFFFFA50001,
              0x7FFFFFFF9F0001, 0x7FFFFFFF7E0001, 0x7FFFFFFF770001, 0x7FFFFFFF380001, 0x7FFFFFFF330001, 0x7FFFFFFF2D0001,
              0x7FFFFFFF170001, 0x7FFFFFFF150001, 0x7FFFFFFEF00001]
what = sys.argv[1] if len(sys.argv) > 1 else "ctpt"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
if what == "keyswitch":          # PF_CONFIG=5: N=32768, 15 data primes + special; batch = argv[3]
    assert os.environ.get("PF_CONFIG") == "5"
    KQ = MODULI + [0xFFFFFFFFF70001]
    D, K = len(MODULI), len(MODULI) + 1
    ctx = pf.RnsContext(N, KQ, dev)
    target = torch.stack([torch.randint(0, q, (B, N), generator=g, device=dev, dtype=torch.int64) for q in MODULI], dim=1).contiguous()
    ksk = torch.stack([torch.stack([torch.stack([torch.randint(0, q, (N,), generator=g, device=dev, dtype=torch.int64) for q in KQ]) for _ in range(2)]) for _ in range(D)]).contiguous()
    ct = torch.stack([torch.randint(0, q, (B, 2, N), generator=g, device=dev, dtype=torch.int64) for q in MODULI], dim=2).contiguous()
    ctx.key_switch_(target, ksk, ct)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ctx.key_switch_(target, ksk, ct)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("keyswitch: %.3f ms per batch of %d (%.1f us per switched polynomial), %d digit NTTs of N=%d" % (ms, B, 1e3 * ms / B, B * D * K, N))
    sys.exit(0)
if what == "encround":           # the server side of the encrypted precise search: NTT of the query ciphertexts + pf_ct_rows_mul
    fan, rows, K = 4, N // 128, 200
    xb = torch.randint(0, 256, (1_000_000, 128), generator=g, device=dev, dtype=torch.int32).float()
    flat = pf.FlatL2(xb, dev)
    ctx = pf.RnsContext(N, MODULI, dev)
    ct = torch.stack([torch.randint(0, q, (B, 2, N), generator=g, device=dev, dtype=torch.int64) for q in MODULI], dim=2).contiguous()
    ids = torch.full((B * fan, rows), -1, dtype=torch.int64, device=dev)
    ids.view(B, fan * rows)[:, :K] = torch.randint(0, 1_000_000, (B, K), generator=g, device=dev)
    ctn, res = torch.empty_like(ct), torch.empty((B * fan, 2, len(MODULI), N), dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(reps + 1):
        if r == 1:
            e0.record()
        ctx.ntt_forward(ct, out=ctn)
        ctx.ct_rows_mul(ctn, flat, ids, fan, out=res)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("encround: %.3f ms per %d queries x %d candidates (%d products): %.0f encrypted precise queries/s" % (ms, B, K, B * fan, B / ms * 1e3))
    sys.exit(0)
if what == "flat":
    if os.environ.get("PF_RK_LAW") == "gauss":                      # N(0,1) rows: the bf16 tiles as a conservative filter
        xb = torch.randn((1_000_000, 128), generator=g, device=dev)
        xq = torch.randn((B, 128), generator=g, device=dev)
    else:
        xb = torch.randint(0, 256, (1_000_000, 128), generator=g, device=dev, dtype=torch.int32).float()
        xq = torch.randint(0, 256, (B, 128), generator=g, device=dev, dtype=torch.int32).float()
    idx = pf.FlatL2(xb, dev)
    for _ in range(reps):
        idx.search(xq, 200)
else:
    ctx = pf.RnsContext(N, MODULI, dev)
    if len(sys.argv) > 4 and sys.argv[4] == "u64":
        ctx.force_u64(True)
    ct = torch.stack([torch.randint(0, q, (B, 2, N), generator=g, device=dev, dtype=torch.int64) for q in MODULI], dim=2).contiguous()
    pt = torch.stack([torch.randint(0, q, (B, N), generator=g, device=dev, dtype=torch.int64) for q in MODULI], dim=1).contiguous()
    out = torch.empty_like(ct)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.ct_pt_mul(ct, pt, out=out)
    e0.record()
    for _ in range(reps):
        if what == "ctpt":
            ctx.ct_pt_mul(ct, pt, out=out)
        elif what == "ntt_fwd":
            ctx.ntt_forward_(ct)
        elif what == "ntt_inv":
            ctx.ntt_inverse_(ct)
        elif what == "dyadic":
            ctx.dyadic_mul(ct, out, out=out)
if what != "flat":
    e1.record()
torch.cuda.synchronize()
if what != "flat":
    ms = e0.elapsed_time(e1) / reps
    per = {"ctpt": 40, "ntt_fwd": 32, "ntt_inv": 32, "dyadic": 48}[what] * len(MODULI) * N * B
    print("%s: %.4f ms/launch  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (what, ms, per / ms / 1e6, per / ms / 1e6 / 80))
print("done", what, reps, B)
