"""GPU tests of the multi-GPU layer on a ONE-GPU box: the packed exchange record written by the selection kernel, the
C-ABI device group (pf_multi_*: one process, one host thread + stream per member) with both members on cuda:0, RCCL
itself on a one-rank communicator, and bench.py's real N > 1 control flow (self-launched rank processes, gloo, all ranks
on cuda:0).  The 8-GPU run is the driver's; these make the path correct before it gets there."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _data(nb=30000, nq=24, d=128, seed=3):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (nb, d)).astype(np.float32), rng.integers(0, 256, (nq, d)).astype(np.float32)


def test_search_packed_is_the_record_of_search():
    import prefhetch_amd as pf
    from prefhetch_amd import dist as pfd
    xb, xq = _data()
    f = pf.FlatL2(xb, DEV)
    q = torch.from_numpy(xq).to(DEV)
    for k in (1, 100, 200):
        D, I = f.search(q, k)
        packed = f.search_packed(q, k)
        assert torch.equal(packed, pfd.pack_topk(D, I))
        Dr, Ir = oracle.flat_l2_search(xb, xq, k)
        D2, I2 = pfd.unpack_topk(packed)
        assert (I2.cpu().numpy() == Ir).all() and (D2.cpu().numpy() == Dr).all()
    # k > nb: padding is (+inf, -1) in the record too
    f2 = pf.FlatL2(xb[:50], DEV)
    D2, I2 = pfd.unpack_topk(f2.search_packed(q, 64))
    assert (I2[:, 50:] == -1).all() and torch.isinf(D2[:, 50:]).all()


@pytest.mark.parametrize("G", [2, 3])
def test_device_group_two_members_on_one_gpu(G):
    import prefhetch_amd as pf
    from prefhetch_amd import dist as pfd
    xb, xq = _data(nq=8 * G)
    grp = pf.DeviceGroup([0] * G)
    assert grp.exchange == "peer_copy"            # RCCL refuses a repeated device; the group says what it uses
    grp.flat(xb)
    nq_local, k = 8, 100
    grp.reserve(nq_local, k)
    shards = [torch.from_numpy(xq[r * nq_local:(r + 1) * nq_local]).to(DEV) for r in range(G)]
    gathered = [torch.zeros((G * nq_local, k, 3), dtype=torch.int32, device=DEV) for _ in range(G)]
    for _ in range(3):                             # repeated calls reuse events and streams
        grp.flat_search(shards, k, gathered)
    grp.synchronize()
    Dr, Ir = oracle.flat_l2_search(xb, xq, k)
    for g in gathered:                             # every member holds the whole result
        D, I = pfd.unpack_topk(g)
        assert (I.cpu().numpy() == Ir).all() and (D.cpu().numpy() == Dr).all()
    # host entry point, batch not divisible by the group size, k > 64
    nq = 8 * G - 1
    D, I = grp.flat_search_host(xq[:nq], 130)
    Dr, Ir = oracle.flat_l2_search(xb, xq[:nq], 130)
    assert (I == Ir).all() and (D == Dr).all()
    # sharded ct x pt: no exchange, every member multiplies its own shard
    N, qs = 4096, oracle.BFV_DEFAULT[4096][:2]
    grp.ring(N, qs)
    rng = np.random.default_rng(5)
    o = oracle.Oracle(N, qs)
    cts, pts, outs, exps = [], [], [], []
    for r in range(G):
        ct = np.stack([rng.integers(0, q, (3, 2, N), dtype=np.uint64) for q in qs], axis=2)
        pt = np.stack([rng.integers(0, q, (3, N), dtype=np.uint64) for q in qs], axis=1)
        cts.append(pf.to_device_u64(ct, DEV)); pts.append(pf.to_device_u64(pt, DEV)); outs.append(torch.empty_like(cts[-1]))
        exps.append(o.ct_pt_mul(ct, pt))
    grp.ct_pt_mul(cts, pts, outs)
    grp.synchronize()
    for got, exp in zip(outs, exps):
        assert (pf.to_host_u64(got) == exp).all()
    grp.close()


def test_device_group_error_paths():
    import prefhetch_amd as pf
    with pytest.raises(pf.PfError):
        pf.DeviceGroup([0, 0], exchange=pf.DeviceGroup.RCCL)       # RCCL needs distinct devices
    with pytest.raises(pf.PfError):
        pf.DeviceGroup([0, 99])
    grp = pf.DeviceGroup([0, 0])
    with pytest.raises(pf.PfError):                                 # search before the base matrix exists
        grp.flat_search_host(np.zeros((2, 128), np.float32), 5)
    grp.close()


def test_one_member_group_runs_rccl():
    """ncclCommInitAll + the in-place ncclAllGather through the C ABI on the real device (a one-rank communicator is what a
    one-GPU box offers): librccl loads, the communicator comes up, the collective runs on the member's stream."""
    import prefhetch_amd as pf
    from prefhetch_amd import dist as pfd
    xb, xq = _data(nb=5000, nq=8)
    grp = pf.DeviceGroup([0], exchange=pf.DeviceGroup.RCCL)
    assert grp.exchange == "rccl"
    grp.flat(xb)
    gathered = [torch.zeros((8, 10, 3), dtype=torch.int32, device=DEV)]
    grp.flat_search([torch.from_numpy(xq).to(DEV)], 10, gathered)
    grp.synchronize()
    D, I = pfd.unpack_topk(gathered[0])
    Dr, Ir = oracle.flat_l2_search(xb, xq, 10)
    assert (I.cpu().numpy() == Ir).all() and (D.cpu().numpy() == Dr).all()
    grp.close()


def _run_bench(extra, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "64", "--nb", "20000",
           "--no-cpu-baseline", "--no-extras"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]            # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    """plain `python bench.py --gpus 2`: the parent starts two fresh rank processes (it has not touched the GPU), they run the
    real step -- packed in-place search, fused ct x pt, one all-gather -- and rank 0 prints one line with n_gpus = 2"""
    r = _run_bench(["--gpus", "2"], {"PF_BENCH_SINGLE_DEVICE": "1", "PF_BENCH_BACKEND": "gloo"})
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["verified"] is True, r
    assert r["verified_detail"]["gathered_block_of_rank_1_first_4_bit_exact"] is True
    assert r["stages_ms"]["gather"] > 0 and "REHEARSAL" in r["data"]


def test_bench_one_rank_through_rccl():
    """the N = 1 line with the process group forced on: RCCL (backend nccl) initialises and runs the in-place all-gather"""
    r = _run_bench(["--gpus", "1"], {"PF_BENCH_FORCE_DIST": "1"})
    assert r["n_gpus"] == 1 and r["verified"] is True and "nccl" in r["config"]["workload"], r


def test_bench_single_process_group():
    r = _run_bench(["--gpus", "2", "--single-process"], {"PF_BENCH_SINGLE_DEVICE": "1"})
    assert r["n_gpus"] == 2 and r["verified"] is True and "pf_multi" in r["config"]["parallelism"], r


def test_bench_default_line_is_unchanged_in_shape():
    r = _run_bench([], {})
    assert r["n_gpus"] == 1 and r["verified"] is True and r["roofline"]["frac"] > 0
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in r
