"""Multi-GPU plumbing: one process per GPU, groups sharded by contiguous
group-id range (groups are independent — no term of quorum/majority.go or
tracker/tracker.go couples two groups), and the path's ONE exchange step: an
all-gather of the changed (group, commit) pairs so that every rank (and the
host loop that feeds raft.maybeCommit, raft.go:755-758) sees the full set.

The exchange itself is in the product: librafthip.so's rafthip_comm_* /
rafthip_commit_allgather run it over RCCL (include/rafthip.h); `make_comm` and
`commit_changed_allgather(comm)` are thin callers.  What stays here in Python is
the sharding arithmetic and a torch.distributed restatement of the exchange
(`allgather_changed`, and `commit_changed_allgather` given a bare engine) that the
CPU tests run over gloo to check that arithmetic without a GPU.
"""
import numpy as np

from .capi import PAIR_DTYPE, Comm, comm_unique_id


def shard_range(total_groups, rank, world):
    """Contiguous group-id range [first, first+count) owned by `rank`."""
    base, rem = divmod(total_groups, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def owner_of(group, total_groups, world):
    """Rank that owns a global group id (inverse of shard_range)."""
    base, rem = divmod(total_groups, world)
    cut = rem * (base + 1)
    if group < cut:
        return group // (base + 1)
    return rem + (group - cut) // base


def route_acks(recs, total_groups, world):
    """Split global-id ack records into per-rank arrays with shard-local ids."""
    base, rem = divmod(total_groups, world)
    g = recs["group"].astype(np.int64)
    cut = rem * (base + 1)
    owner = np.where(g < cut, g // (base + 1), rem + (g - cut) // max(base, 1))
    out = []
    for r in range(world):
        first, _ = shard_range(total_groups, r, world)
        sel = recs[owner == r].copy()
        sel["group"] -= np.uint32(first)
        out.append(sel)
    return out


def _allgather_pairs(buf, n, world, group):
    """buf: int64 tensor holding n (group, commit) pairs as 2n words, on the device
    the process group communicates on.  Count all-gather, then one padded payload
    all-gather; returns the concatenation over ranks as numpy PAIR_DTYPE."""
    import torch
    import torch.distributed as dist

    dev = buf.device
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    send = buf[: 2 * width] if buf.numel() >= 2 * width else torch.cat(
        [buf[: 2 * n], torch.zeros(2 * width - 2 * n, dtype=torch.int64, device=dev)])
    outs = [torch.empty(width * 2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(outs, send.contiguous(), group=group)
    parts = [o[: 2 * c].cpu().numpy().view(PAIR_DTYPE) for o, c in zip(outs, counts)]
    return np.concatenate(parts)


def allgather_changed(pairs, group_base=0, device=None, group=None):
    """All-gather changed pairs across ranks.

    pairs: numpy PAIR_DTYPE with shard-local group ids; group_base: first global
    group id of this rank's shard.  Returns the concatenation over ranks (global
    ids, rank order — hence sorted by group) as numpy PAIR_DTYPE.  Without an
    initialised process group this is the local set, rebased.
    """
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE).copy()
    local["group"] += np.uint64(group_base)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    dev = device if device is not None else torch.device("cpu")
    buf = torch.from_numpy(local.view(np.int64).reshape(-1).copy()).to(dev)
    return _allgather_pairs(buf, len(local), dist.get_world_size(group), group)


def make_comm(engine, group_base=0, max_pairs_per_rank=0, group=None):
    """The engine's communicator (rafthip_comm_create) over the ranks of the initialised
    torch.distributed job — or a one-rank communicator without one.  Rank 0 draws the id; the
    job's own process group only carries those 128 bytes."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    box = [comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    return Comm(engine, box[0], rank, world, group_base, max_pairs_per_rank)


def commit_changed_allgather(engine, group_base=0, device=None, group=None, max_pairs=None, advance=False):
    """The path's one exchange step.  Given a capi.Comm: rafthip_commit_allgather, the product
    path over RCCL.  Given a bare engine (CPU tests over gloo): the same exchange restated on
    torch.distributed — the engine's commit pass packs its pairs into a tensor on `device`,
    group ids are rebased there, count all-gather, padded payload all-gather.  Returns (all
    pairs as numpy PAIR_DTYPE sorted by global group id, this rank's pair count)."""
    if isinstance(engine, Comm):
        pairs, counts, _ = engine.commit_allgather(advance=advance, max_pairs=max_pairs)
        return pairs, int(counts[engine.rank])
    import torch
    import torch.distributed as dist

    cap = engine.G if max_pairs is None else max_pairs
    dev = device if device is not None else torch.device("cpu")
    buf = torch.empty(2 * max(cap, 1), dtype=torch.int64, device=dev)
    n_total = engine.commit_report_into(buf.data_ptr(), cap, advance=advance)
    n = min(n_total, cap)
    if n:
        buf[0:2 * n:2] += group_base
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return buf[: 2 * n].cpu().numpy().view(PAIR_DTYPE), n_total
    return _allgather_pairs(buf, n, dist.get_world_size(group), group), n_total


def allgather_commit_planes(plane, width, group=None):
    """The dense form of the exchange restated on torch.distributed (CPU tests over gloo; the product
    path is rafthip_commit_allgather_dense): every rank contributes its commit plane padded to `width`
    entries — the largest plane stride of any rank, agreed beforehand — and receives [world][width]."""
    import torch
    import torch.distributed as dist

    mine = np.zeros(width, dtype=np.uint64)
    mine[: len(plane)] = plane
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return mine.reshape(1, width)
    world = dist.get_world_size(group)
    outs = [torch.empty(width, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(outs, torch.from_numpy(mine.view(np.int64)), group=group)
    return np.stack([o.numpy().view(np.uint64) for o in outs])


def agree_width(stride, group=None):
    """The dense width: the largest plane stride over the ranks (what rafthip_comm_create agrees on)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(stride)
    t = torch.tensor([int(stride)], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())
