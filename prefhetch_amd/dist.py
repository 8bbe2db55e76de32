"""Multi-GPU layer: one process per GPU, torch.distributed over RCCL (backend "nccl") on xGMI.

The path shards by independent units: every encrypted query / ciphertext / flat-L2 query
is independent, so the batch dimension is split contiguously over ranks, the RNS tables, the plaintext DB and
the fp32 base matrix are replicated, and the ONLY exchange step is one all-gather of the per-rank top-k block.
Payload is ~2.4 KB per query (k=200), latency- not bandwidth-bound on 153 GB/s xGMI links, hence a single
collective per batch with indices and distances packed in one buffer -- never one collective per query.

The reference has no counterpart (single-process server, /root/reference/src/server/server_lib.cpp).
Everything here works on CPU tensors with the gloo backend too, which is how the N > 1 path is tested
without GPUs (tests/test_dist_gloo.py).
"""
import torch


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n units for `rank`; earlier ranks take the remainder."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_topk(D, I):
    """(D [nq,k] float32, I [nq,k] int64) -> one int32 buffer [nq,k,3]: id low word, id high word, distance bits."""
    nq, k = D.shape
    if I.shape != (nq, k) or D.dtype != torch.float32 or I.dtype != torch.int64:
        raise ValueError("pack_topk: need D [nq,k] float32 and I [nq,k] int64")
    return torch.cat([I.contiguous().view(torch.int32).view(nq, k, 2), D.contiguous().view(torch.int32).unsqueeze(-1)], dim=-1).contiguous()


def unpack_topk(packed):
    """Inverse of pack_topk for a buffer [..., k, 3]."""
    I = packed[..., :2].contiguous().view(torch.int64).squeeze(-1)
    D = packed[..., 2].contiguous().view(torch.float32)
    return D, I


def gather_topk(D, I, group=None, out=None):
    """All ranks contribute their [nq_local,k] block (equal nq_local on every rank); every rank receives the
    concatenation [world*nq_local, k] in rank order.  ONE all_gather_into_tensor."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    packed = pack_topk(D, I)
    if out is None:       # concatenation along dim 0 (the layout both RCCL and gloo accept)
        out = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=torch.int32, device=packed.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    Dg, Ig = unpack_topk(out)
    return Dg, Ig, out


class ShardedPrefilter:
    """Query-sharded IndexFlatL2 pre-filter: the base matrix is replicated on every rank's GPU; rank r searches
    queries shard_range(nq, r, world) and all ranks end up with the full (D, I)."""

    def __init__(self, flat, group=None):
        self.flat, self.group = flat, group

    def search(self, xq_all, k):
        import torch.distributed as dist
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        nq = xq_all.shape[0]
        if nq % world:
            raise ValueError("ShardedPrefilter.search: nq must be a multiple of the world size (pad the batch)")
        lo, hi = shard_range(nq, rank, world)
        D, I = self.flat.search(xq_all[lo:hi].contiguous(), k)
        Dg, Ig, _ = gather_topk(D, I, self.group)
        return Dg, Ig
