#!/usr/bin/env python3
"""The bench step (pre-filter + fused ct x pt, BASELINE config 3) replayed from a hipGraph against the same step launched
eagerly: what the launch path costs.  usage: python3 tools/time_step_graph.py [reps=40]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402

MODULI = [0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001]
N, B, NB, K = 8192, 1024, 1_000_000, 200
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
xb = torch.randint(0, 256, (NB, 128), generator=g, device=dev, dtype=torch.int32).float()
xq = torch.randint(0, 256, (B, 128), generator=g, device=dev, dtype=torch.int32).float()
flat = pf.FlatL2(xb, dev)
flat.reserve(B, K)
ctx = pf.RnsContext(N, MODULI, dev)
ct = torch.stack([torch.randint(0, q, (B, 2, N), generator=g, device=dev, dtype=torch.int64) for q in MODULI], dim=2).contiguous()
pt = torch.stack([torch.randint(0, q, (B, N), generator=g, device=dev, dtype=torch.int64) for q in MODULI], dim=1).contiguous()
out = torch.empty_like(ct)


def step():
    D, I = flat.search(xq, K)
    ctx.ct_pt_mul(ct, pt, out=out)
    return D, I


def timed(fn):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    De, Ie = step()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    Dg, Ig = step()
graph.replay()
torch.cuda.synchronize()
same = bool(torch.equal(Ig, Ie) and torch.equal(Dg.view(torch.int32), De.view(torch.int32)))
ms_e1, ms_g1, ms_e2, ms_g2 = timed(step), timed(graph.replay), timed(step), timed(graph.replay)
print('{"eager_ms": [%.4f, %.4f], "graph_replay_ms": [%.4f, %.4f], "results_identical": %s}' % (ms_e1, ms_e2, ms_g1, ms_g2, "true" if same else "false"))
# per-replay host-side view of a fresh graph: how long until replays are at speed
graph2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph2):
    step()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
torch.cuda.synchronize()
evs[0].record()
for i in range(60):
    graph2.replay()
    evs[i + 1].record()
torch.cuda.synchronize()
print("first replays of a fresh graph, ms:", " ".join("%.2f" % evs[i].elapsed_time(evs[i + 1]) for i in range(60)))
