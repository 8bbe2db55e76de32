#!/usr/bin/env python3
"""Differential fuzz of pf_key_switch at N = 32768 (the two-pass split + fused tail) against the oracle: random digit counts, random
NTT-friendly moduli of mixed widths below 2^56, random batch sizes around the workspace round.  usage: tools/fuzz_keyswitch.py [iterations] [seed]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402
import prefhetch_amd as pf  # noqa: E402

N = 32768
dev = torch.device("cuda", 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)


def is_prime(n):
    if n < 2: return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0: return n == p
    d, s = n - 1, 0
    while d % 2 == 0: d //= 2; s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1): continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1: break
        else:
            return False
    return True


def prime_below(bits):
    q = ((1 << bits) - int(rng.integers(0, 1 << 20)) * 2 * N) // (2 * N) * (2 * N) + 1
    while not is_prime(q):
        q -= 2 * N
    return q


bad = 0
for it in range(iters):
    D = int(rng.integers(1, 8))
    qs = []
    while len(qs) < D + 1:
        q = prime_below(int(rng.integers(46, 57)))
        if q not in qs and q < (1 << 56):
            qs.append(q)
    K = D + 1
    B = int(rng.choice([1, 2, 3, 33]))
    o = oracle.Oracle(N, qs)
    ctx = pf.RnsContext(N, qs, dev)
    assert ctx.info()["arith_path"][0] == 2, "lazy 64-bit family expected"
    distinct = min(B, 2)
    target = np.stack([rng.integers(0, q, (distinct, N), dtype=np.uint64) for q in qs[:D]], axis=1)
    ct = np.stack([rng.integers(0, q, (distinct, 2, N), dtype=np.uint64) for q in qs[:D]], axis=2)
    ksk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(D)])
    exp = o.key_switch(target, ksk, ct)
    sel = np.arange(B) % distinct
    d_t = pf.to_device_u64(target[sel], dev)
    d_c = pf.to_device_u64(ct[sel], dev)
    ctx.key_switch_(d_t, pf.to_device_u64(ksk, dev), d_c)
    ok = bool((pf.to_host_u64(d_c) == exp[sel]).all())
    print("%3d D=%d B=%2d bits=%s %s" % (it, D, B, [q.bit_length() for q in qs], "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
print("fuzz:", "all ok" if not bad else "%d MISMATCHES" % bad)
sys.exit(1 if bad else 0)
