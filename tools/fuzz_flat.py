#!/usr/bin/env python3
"""Differential fuzz of FlatL2.search: bf16 tiles vs fp32 operands (bit for bit), and the oracle on integer-valued data.
usage: tools/fuzz_flat.py [iterations] [seed] [i8 | wide]   (wide: row lengths above 256 -- the slab tiles -- including odd ones; no oracle
comparison there on integer data beyond 2^24, where the decomposition itself rounds)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402
import prefhetch_amd as pf  # noqa: E402

dev = torch.device("cuda", 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(iters):
    d = int(rng.choice(np.arange(16, 257, 16)))
    only8 = len(sys.argv) > 3 and sys.argv[3] == "i8"          # third argument "i8": 8-bit data at the int8 tiles' row lengths only
    if only8:
        d = int(rng.choice([32, 64, 96, 128]))
    wide = len(sys.argv) > 3 and sys.argv[3] == "wide"
    if wide:
        d = int(rng.choice([257, 260, 300, 320, 384, 511, 512, 640, 777, 1024, 1536]))
    nb = int(rng.choice([rng.integers(1, 300), rng.integers(300, 9000), rng.integers(9000, 60000), rng.integers(60000, 400000)]))
    if wide:
        nb = int(rng.choice([rng.integers(1, 300), rng.integers(300, 9000), rng.integers(9000, 40000), rng.integers(40000, 120000)]))
    nq = int(rng.choice([rng.integers(1, 70), rng.integers(70, 300), rng.integers(300, 700)]))
    k = int(min(rng.choice([1, 7, 64, 200, 256, 257, 1024]), 1024))
    law = str(rng.choice(["int", "ties", "dups", "part8"])) if len(sys.argv) > 3 and sys.argv[3] == "i8" else str(rng.choice(["int", "ties", "gauss", "mixed", "dups", "neg"]))
    top = 256 if d <= 128 else 128       # (|x|^2 + |y|^2 < 2^24: the oracle's exact distance is what the fp32 formula returns)
    if wide:
        top = 64 if d <= 512 else 32
    if law == "int":
        xb, xq = rng.integers(0, top, (nb, d)), rng.integers(0, top, (nq, d))
    elif law == "ties":
        xb, xq = rng.integers(0, 3, (nb, d)), rng.integers(0, 3, (nq, d))
    elif law == "neg":
        xb, xq = rng.integers(-top, top + 1, (nb, d)), rng.integers(-top, top + 1, (nq, d))
    elif law == "part8":                                      # 8-bit base; some query tiles carry a value outside [0, 255] (bf16 tiles for those)
        xb, xq = rng.integers(0, 256, (nb, d)), rng.integers(0, 256, (nq, d)).astype(np.float64)
        for t in range(0, nq, 128):
            if rng.random() < 0.5:
                xq[t + int(rng.integers(0, min(128, nq - t))), int(rng.integers(0, d))] = float(rng.choice([-1.0, 256.0, -200.0]))
    elif law == "gauss":
        xb, xq = rng.standard_normal((nb, d)), rng.standard_normal((nq, d))
    elif law == "mixed":
        xb, xq = rng.standard_normal((nb, d)) * rng.choice([0.01, 1, 40], (nb, 1)), rng.standard_normal((nq, d)) * rng.choice([0.1, 1, 10], (nq, 1))
    else:
        xb = rng.integers(0, top, (max(nb // 50, 1), d))[rng.integers(0, max(nb // 50, 1), nb)]      # every row ~50 times
        xq = xb[rng.integers(0, nb, nq)]
    xb, xq = np.ascontiguousarray(xb, np.float32), np.ascontiguousarray(xq, np.float32)
    f = pf.FlatL2(xb, dev)
    q = torch.from_numpy(xq).to(dev)
    mode, m8 = f.operands16(), f.operands8()
    D1, I1 = f.search(q, k)
    ok = True
    if m8:                                                    # int8 tiles ran: the bf16 tiles on the same data too
        f.operands8(0)
        D2, I2 = f.search(q, k)
        ok = bool((I1 == I2).all() and (D1.view(torch.int32) == D2.view(torch.int32)).all())
    f.operands16(0)
    D0, I0 = f.search(q, k)
    ok = ok and bool((I1 == I0).all() and (D1.view(torch.int32) == D0.view(torch.int32)).all())
    if ok and law in ("int", "ties", "neg", "dups", "part8"):
        Dr, Ir = oracle.flat_l2_search(xb, xq[:2], k)
        ok = bool((I1[:2].cpu().numpy() == Ir).all() and (D1[:2].cpu().numpy() == Dr).all())
    print("%3d nb=%6d nq=%4d k=%4d d=%3d %-5s operands16=%d int8=%d %s" % (it, nb, nq, k, d, law, mode, m8, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
    del f
print("fuzz:", "all ok" if not bad else "%d MISMATCHES" % bad)
sys.exit(1 if bad else 0)
