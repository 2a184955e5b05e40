"""C1 — the exchange step in the product (rafthip_comm_*, rafthip_commit_allgather,
include/rafthip.h) on a one-rank RCCL communicator: what comes out of the all-gather
must be rafthip_commit_changed / _advance's pairs with global group ids.  More ranks
need more GPUs than a test box has; the per-rank arithmetic (counts, slot offsets,
rank order = group order) is the same code, and the sharding maths is covered on CPU
over gloo (tests/test_multigpu_cpu.py)."""
import numpy as np
import pytest

import oracle
import raft_amd
from raft_amd import multigpu
from raft_amd.capi import PAIR_DTYPE, Comm, PinnedArray, comm_unique_id

pytestmark = pytest.mark.gpu


def _engine_pair(G, S, seed, uniform):
    rng = np.random.default_rng(seed)
    match = rng.integers(0, 1 << 40, size=(S, G), dtype=np.uint64)
    a, b = raft_amd.Engine(G, S, device=0), raft_amd.Engine(G, S, device=0)
    md = a.mask_dtype
    hi = (1 << S) - 1
    in_mask = rng.integers(1, hi, size=G, dtype=np.uint64, endpoint=True).astype(md)
    out_mask = np.where(rng.random(G) < 0.3, rng.integers(0, hi, size=G, dtype=np.uint64, endpoint=True), 0).astype(md)
    for e in (a, b):
        if uniform:
            e.set_uniform_config(uniform)
        else:
            e.upload_config(in_mask, out_mask)
        e.upload_match(match)
    a.shadow = dict(match=match, cfg=dict(uniform_voters=uniform) if uniform else dict(in_mask=in_mask, out_mask=out_mask))
    return a, b, rng


def _acks(rng, G, S, n, lo):
    recs = np.zeros(n, dtype=raft_amd.ACK_DTYPE)
    recs["group"] = rng.integers(0, G, size=n)
    recs["slot"] = rng.integers(0, S, size=n)
    recs["index"] = rng.integers(lo, lo + (1 << 20), size=n, dtype=np.uint64)
    return recs


@pytest.mark.parametrize("G,S,uniform", [(100_003, 5, 5), (70_001, 6, 0), (20_000, 20, 0)])
def test_one_rank_allgather_equals_commit_changed(G, S, uniform):
    a, b, rng = _engine_pair(G, S, 3 + S, uniform)
    base = 7_000_000_000  # global ids beyond 32 bits
    comm = Comm(a, comm_unique_id(), 0, 1, group_base=base)
    try:
        assert (comm.rank, comm.nranks) == (0, 1) and comm.max_pairs_per_rank >= G
        # exchanges with very different pair counts in a row: all groups, a few, none, many again —
        # the width guess is wrong in both directions
        prev = np.zeros(G, dtype=np.uint64)
        for k, n_acks in enumerate([0, 500, 0, G // 2, 3]):
            recs = _acks(rng, G, S, n_acks, (k + 1) << 41)
            a.scatter_acks(recs)
            b.scatter_acks(recs)
            want, n_want = b.commit_changed()
            got, counts, total = comm.commit_allgather()
            assert total == n_want == counts[0] and len(got) == n_want
            assert np.array_equal(got["group"], want["group"] + np.uint64(base))
            assert np.array_equal(got["commit"], want["commit"])
            assert np.array_equal(a.read_commit(), b.read_commit())
            # and against the oracle's changed set, not only the twin engine
            assert oracle.scatter_acks(a.shadow["match"], recs) == 0
            after = oracle.batch_committed(a.shadow["match"], **a.shadow["cfg"])
            ch = np.flatnonzero(after != prev)
            prev = after
            assert np.array_equal(got["group"], ch.astype(np.uint64) + np.uint64(base)) and np.array_equal(got["commit"], after[ch])
    finally:
        comm.close()
        a.close()
        b.close()


def test_one_rank_allgather_advance_device_and_pinned_dst():
    import torch
    G, S = 50_000, 7
    a, b, rng = _engine_pair(G, S, 17, 7)
    comm = multigpu.make_comm(a, group_base=123)  # no process group: a one-rank communicator
    try:
        tf = np.full(G, 1 << 41, dtype=np.uint64)
        tl = np.full(G, (1 << 41) + (1 << 19), dtype=np.uint64)  # about half of the acks land inside the term's run
        for e in (a, b):
            e.upload_commit(np.zeros(G, dtype=np.uint64))
            e.upload_term_range(tf, tl)
        for dst_kind in ("device", "pinned", "count-only"):
            recs = _acks(rng, G, S, 4 * G, 1 << 41)
            a.scatter_acks(recs)
            b.scatter_acks(recs)
            want, n_want = b.commit_advance()
            if dst_kind == "device":
                buf = torch.zeros(2 * G, dtype=torch.int64, device="cuda:0")
                _, counts, total = comm.commit_allgather(advance=True, max_pairs=G, dst=buf.data_ptr())
                got = buf[:2 * n_want].cpu().numpy().view(PAIR_DTYPE)
            elif dst_kind == "pinned":
                pin = PinnedArray(G, PAIR_DTYPE)
                _, counts, total = comm.commit_allgather(advance=True, max_pairs=G, dst=pin.array.ctypes.data)
                got = pin.array[:n_want].copy()
                pin.free()
            else:
                got, counts, total = comm.commit_allgather(advance=True, max_pairs=0)
                assert total == n_want and len(got) == 0
                assert np.array_equal(a.read_commit(), b.read_commit())
                continue
            assert total == n_want == counts[0]
            assert np.array_equal(got["group"], want["group"] + np.uint64(123))
            assert np.array_equal(got["commit"], want["commit"])
        pairs, n_local = multigpu.commit_changed_allgather(comm, advance=True)  # the thin caller bench.py uses
        assert n_local == len(pairs)
    finally:
        comm.close()
        a.close()
        b.close()


def test_per_rank_cap_clips_payload_not_count():
    G, S = 30_000, 3
    a, b, rng = _engine_pair(G, S, 29, 3)
    comm = Comm(a, comm_unique_id(), 0, 1, group_base=0, max_pairs_per_rank=1000)
    try:
        assert comm.max_pairs_per_rank == 1024  # rounded up to the payload granule
        want, n_want = b.commit_changed()
        got, counts, total = comm.commit_allgather()
        assert total == n_want == G and counts[0] == G
        assert len(got) == 1024 and np.array_equal(got["commit"], want["commit"][:1024])
    finally:
        comm.close()
        a.close()
        b.close()


@pytest.mark.parametrize("G,S,uniform", [(70_001, 5, 5), (9_000, 12, 0)])
def test_one_rank_dense_allgather_is_the_commit_plane(G, S, uniform):
    """The dense form of the exchange ("dense fallback"): commit pass + all-gather of the commit
    planes, into a pageable buffer, a device buffer and a page-locked one; with the term guard too."""
    import torch
    a, b, rng = _engine_pair(G, S, 41 + S, uniform)
    comm = Comm(a, comm_unique_id(), 0, 1, group_base=0)
    try:
        width = comm.dense_width()
        assert width >= G and width % 4096 == 0
        recs = _acks(rng, G, S, 3 * G, 1 << 41)
        a.scatter_acks(recs)
        b.scatter_acks(recs)
        b.commit()
        want = b.read_commit()
        planes, w = comm.commit_allgather_dense()
        assert w == width and planes.shape == (1, width)
        assert np.array_equal(planes[0, :G], want)
        assert oracle.scatter_acks(a.shadow["match"], recs) == 0
        assert np.array_equal(planes[0, :G], oracle.batch_committed(a.shadow["match"], **a.shadow["cfg"]))
        buf = torch.zeros(width, dtype=torch.int64, device="cuda:0")
        none, w = comm.commit_allgather_dense(dst=buf.data_ptr(), dst_elems=width)
        assert none is None and np.array_equal(buf[:G].cpu().numpy().view(np.uint64), want)
        pin = PinnedArray(width, np.uint64)
        comm.commit_allgather_dense(dst=pin.array.ctypes.data, dst_elems=width)
        assert np.array_equal(pin.array[:G], want)
        with pytest.raises(raft_amd.RafthipError):
            comm.commit_allgather_dense(dst=buf.data_ptr(), dst_elems=width - 1)
        # the term-guarded pass: the plane is raftLog.committed afterwards
        tf = np.full(G, 1 << 41, dtype=np.uint64)
        tl = np.full(G, (1 << 41) + (1 << 19), dtype=np.uint64)
        for e in (a, b):
            e.upload_commit(np.zeros(G, dtype=np.uint64))
            e.upload_term_range(tf, tl)
        b.commit_advance()
        planes, _ = comm.commit_allgather_dense(advance=True)
        assert np.array_equal(planes[0, :G], b.read_commit())
    finally:
        comm.close()
        a.close()
        b.close()
