"""world_size-2 CPU tests (gloo) of the multi-GPU layer: sharding, packing and the single all-gather."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from prefhetch_amd import dist as pfd


def test_shard_range_covers_exactly():
    for n in (0, 1, 7, 8, 1024, 8191):
        for world in (1, 2, 3, 8):
            spans = [pfd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pack_roundtrip_bits():
    g = torch.Generator().manual_seed(0)
    D = torch.rand((5, 7), generator=g)
    D[0, 0] = float("inf")
    I = torch.randint(-1, 2**40, (5, 7), generator=g, dtype=torch.int64)
    I[0, 0] = -1
    D2, I2 = pfd.unpack_topk(pfd.pack_topk(D, I))
    assert torch.equal(D.view(torch.int32), D2.view(torch.int32)) and torch.equal(I, I2)


class _CpuIndex:
    """CPU stand-in with FlatL2's interface (this container has no GPU): exact brute force in float64.  Test-local; the
    real index under the same collective is exercised on the GPU box by tests/test_gpu_multi.py, which runs bench.py's
    own control flow with two ranks."""

    def __init__(self, xb):
        self.xb = xb

    def search(self, xq, k):
        d = ((xq[:, None, :].double() - self.xb[None].double()) ** 2).sum(-1)
        key = torch.argsort(d, dim=1, stable=True)[:, :k]
        return torch.gather(d, 1, key).float(), key

    def search_packed(self, xq, k, out=None):
        D, I = self.search(xq, k)
        out.copy_(pfd.pack_topk(D, I))
        return out


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        xb = torch.randint(0, 256, (500, 16), generator=g).float()
        xq = torch.randint(0, 256, (8, 16), generator=g).float()
        eng = pfd.ShardedPrefilter(_CpuIndex(xb))
        D, I = eng.search(xq, 5)
        Dref, Iref = _CpuIndex(xb).search(xq, 5)
        ok = torch.equal(I, Iref) and torch.equal(D, Dref)
        # gather_topk keeps rank order and needs exactly one collective buffer
        lo, hi = pfd.shard_range(8, rank, world)
        Dg, Ig, buf = pfd.gather_topk(Dref[lo:hi], Iref[lo:hi])
        ok = ok and torch.equal(Ig, Iref) and buf.shape == (world * (hi - lo), 5, 3)
        # the in-place form bench.py uses: the rank's block is written into the gathered buffer, one collective
        gathered = torch.zeros((8, 5, 3), dtype=torch.int32)
        pfd.search_and_gather(_CpuIndex(xb), xq[lo:hi].contiguous(), 5, gathered)
        D2, I2 = pfd.unpack_topk(gathered)
        ok = ok and torch.equal(I2, Iref) and torch.equal(D2, Dref)
        np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([int(ok)]))
    finally:
        dist.destroy_process_group()


def test_sharded_prefilter_two_ranks_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(int(np.load(tmp_path / f"ok_{r}.npy")[0]) == 1 for r in range(2))
