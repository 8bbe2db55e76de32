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
    """(D [nq,k] float32, I [nq,k] int64) -> one int32 buffer [nq,k,3]: id low word, id high word, distance bits.
    Host-side definition of the record; on the GPU path the selection kernel writes it directly (FlatL2.search_packed)."""
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


def search_and_gather(flat, xq_local, k, gathered, group=None):
    """The multi-GPU step of the pre-filter with no packing pass: this rank's selection kernel writes its block
    [nq_local, k, 3] IN PLACE at its offset of `gathered` [world * nq_local, k, 3] (int32, preallocated), then ONE
    all_gather_into_tensor completes every rank's copy (in place on RCCL; other backends get a private send buffer).
    Returns `gathered`; unpack_topk(gathered) gives (D, I) of all world * nq_local queries in rank order."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nq = xq_local.shape[0]
    if gathered.shape != (world * nq, k, 3) or gathered.dtype != torch.int32 or not gathered.is_contiguous():
        raise ValueError("search_and_gather: gathered must be a contiguous int32 [world * nq_local, k, 3]")
    block = gathered[rank * nq:(rank + 1) * nq]
    flat.search_packed(xq_local, k, out=block)
    send = block if dist.get_backend(group) == "nccl" else block.clone()
    dist.all_gather_into_tensor(gathered, send, group=group)
    return gathered


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
        gathered = torch.empty((nq, k, 3), dtype=torch.int32, device=xq_all.device)
        search_and_gather(self.flat, xq_all[lo:hi].contiguous(), k, gathered, self.group)
        return unpack_topk(gathered)
