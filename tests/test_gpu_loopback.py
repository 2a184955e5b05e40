"""C1 with more than one rank, on one GPU: the N > 1 branch of rafthip_commit_allgather_many /
rafthip_commit_allgather_dense_many executed through the library's loopback transport
(raft_amd/csrc/rafthip_internal.h: n engines on the same GPU are n ranks, the gather is device copies
ordered with events — RCCL refuses two ranks on one GPU).  Everything but the gather is the shipped
code: the reporting pass with group_base, the speculative payload width and its redo, k_pack_gathered's
rank offsets, clipping at the per-rank capacity, landing buffers, the dense form with ranks of
different plane strides, the capacity / width agreement at creation.

Expected values come from the ORACLE run on the concatenated global planes (consumer:
raft.maybeCommit, raft.go:755-758 — the changed groups of the whole node, sorted by group id), never
from a twin engine."""
import numpy as np
import pytest

import golden_util as gu
import raft_amd
from raft_amd import multigpu
from raft_amd.capi import PAIR_DTYPE, CommSet, PinnedArray

pytestmark = pytest.mark.gpu


def _bounds(sizes):
    first = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    return first[:-1], first[-1]


def _route(recs, firsts, sizes):
    """global-id ack records -> per-shard arrays with shard-local ids (ragged shards)."""
    g = recs["group"].astype(np.int64)
    out = []
    for f, s in zip(firsts, sizes):
        sel = recs[(g >= f) & (g < f + s)].copy()
        sel["group"] -= np.uint32(f)
        out.append(sel)
    return out


def _acks(rng, G, S, n, lo):
    recs = np.zeros(n, dtype=raft_amd.ACK_DTYPE)
    recs["group"] = rng.integers(0, G, size=n)
    recs["slot"] = rng.integers(0, S, size=n)
    recs["index"] = rng.integers(lo, lo + (1 << 20), size=n, dtype=np.uint64)
    return recs


def _make_ranks(sizes, S, match, uniform, m0=None, m1=None):
    firsts, _ = _bounds(sizes)
    engines = []
    for f, s in zip(firsts, sizes):
        e = raft_amd.Engine(int(s), S, device=0)
        if uniform:
            e.set_uniform_config(uniform)
        else:
            e.upload_config(np.ascontiguousarray(m0[f:f + s]), np.ascontiguousarray(m1[f:f + s]))
        e.upload_match(np.ascontiguousarray(match[:, f:f + s]))
        engines.append(e)
    return engines, firsts


def _close(comms, engines):
    comms.close()
    for e in engines:
        e.close()


@pytest.mark.parametrize("sizes,S,uniform", [
    ((70_001, 3, 130_000), 5, 5),                                   # n = 3, one tiny shard, K1
    ((9_000, 50_000), 6, 0),                                        # n = 2, joint masks, K2
    ((4_097, 20_000, 1, 8_191, 30_000, 2_048, 12_345, 60_000), 5, 5),  # n = 8
    ((3_000, 5_000, 2_500), 20, 0),                                 # n = 3, one group per lane (K4r), 64-group spans
])
def test_loopback_pairs_equal_the_oracle_on_the_global_planes(oracle, sizes, S, uniform):
    rng = np.random.default_rng(1000 + len(sizes) + S)
    sizes = np.array(sizes)
    firsts, G = _bounds(sizes)
    base = 5_000_000_000  # global ids beyond 32 bits
    match = rng.integers(0, 1 << 40, size=(S, G), dtype=np.uint64)
    md = raft_amd.mask_dtype(S)
    m0 = m1 = None
    if not uniform:
        hi = (1 << S) - 1
        m0 = rng.integers(1, hi, size=G, dtype=np.uint64, endpoint=True).astype(md)
        m1 = np.where(rng.random(G) < 0.3, rng.integers(0, hi, size=G, dtype=np.uint64, endpoint=True), 0).astype(md)
    engines, firsts = _make_ranks(sizes, S, match, uniform, m0, m1)
    comms = CommSet.loopback(engines, [base + f for f in firsts])
    try:
        n = len(sizes)
        assert comms.max_pairs_per_rank >= sizes.max() and all(c == comms.caps[0] for c in comms.caps)
        prev = np.zeros(G, dtype=np.uint64)  # what the engines' commit planes hold
        # churn jumps both ways: everything (first pass), nothing, a few, most, nothing, a handful
        for k, n_acks in enumerate([0, 0, 300, 3 * G, 0, 7]):
            recs = _acks(rng, G, S, n_acks, (k + 1) << 41)
            for e, part in zip(engines, _route(recs, firsts, sizes)):
                e.scatter_acks(part)
            assert oracle.scatter_acks(match, recs) == 0
            after = oracle.batch_committed(match, m0, m1, uniform_voters=uniform)
            ch = np.flatnonzero(after != prev)
            prev = after
            want_counts = [int(((ch >= f) & (ch < f + s)).sum()) for f, s in zip(firsts, sizes)]
            outs, counts, totals = comms.commit_allgather()
            for r in range(n):  # identical on every rank, sorted by global group id
                assert totals[r] == len(ch), (k, r)
                assert [int(c) for c in counts[r]] == want_counts, (k, r)
                assert np.array_equal(outs[r]["group"], ch.astype(np.uint64) + np.uint64(base)), (k, r)
                assert np.array_equal(outs[r]["commit"], after[ch]), (k, r)
            for e, f, s in zip(engines, firsts, sizes):
                assert np.array_equal(e.read_commit(), after[f:f + s])
    finally:
        _close(comms, engines)


def test_loopback_advance_into_device_and_pinned_buffers(oracle):
    """The term-guarded pass (raft.maybeCommit with raftLog.maybeCommit fused, log.go:447-456) through the exchange,
    gathered pairs landing in device memory on one rank, page-locked memory on another, nowhere on the third."""
    import torch
    sizes = np.array([40_000, 25_001, 60_000])
    S, V = 7, 7
    firsts, G = _bounds(sizes)
    rng = np.random.default_rng(77)
    match = rng.integers(0, 1 << 40, size=(S, G), dtype=np.uint64)
    engines, firsts = _make_ranks(sizes, S, match, V)
    tf = np.full(G, 1 << 41, dtype=np.uint64)
    tl = np.full(G, (1 << 41) + (1 << 19), dtype=np.uint64)  # about half of the acks land inside the term's run
    committed = np.zeros(G, dtype=np.uint64)
    for e, f, s in zip(engines, firsts, sizes):
        e.upload_commit(committed[f:f + s])
        e.upload_term_range(tf[f:f + s], tl[f:f + s])
    comms = CommSet.loopback(engines, firsts)
    try:
        for k in range(3):
            recs = _acks(rng, G, S, 4 * G, 1 << 41)
            for e, part in zip(engines, _route(recs, firsts, sizes)):
                e.scatter_acks(part)
            oracle.scatter_acks(match, recs)
            mci = oracle.batch_committed(match, uniform_voters=V)
            adv = (mci > committed) & (mci >= tf) & (mci <= tl)  # the guard of log.go:447-456 on the [term_first, term_last] summary
            committed = np.where(adv, mci, committed)
            ch = np.flatnonzero(adv)
            dev = torch.zeros(2 * G, dtype=torch.int64, device="cuda:0")
            pin = PinnedArray(G, PAIR_DTYPE)
            _, counts, totals = comms.commit_allgather(advance=True, max_pairs=[G, G, 0],
                                                       dsts=[dev.data_ptr(), pin.array.ctypes.data, None])
            assert totals == [len(ch)] * 3
            for got in (dev[:2 * len(ch)].cpu().numpy().view(PAIR_DTYPE), pin.array[:len(ch)]):
                assert np.array_equal(got["group"], ch.astype(np.uint64))
                assert np.array_equal(got["commit"], committed[ch])
            pin.free()
            for e, f, s in zip(engines, firsts, sizes):
                assert np.array_equal(e.read_commit(), committed[f:f + s])
    finally:
        _close(comms, engines)


def test_loopback_one_rank_over_the_capacity_clips_its_payload_not_its_count(oracle):
    sizes = np.array([5_000, 30_000, 9_000])
    S, V = 3, 3
    firsts, G = _bounds(sizes)
    rng = np.random.default_rng(5)
    match = rng.integers(1, 1 << 40, size=(S, G), dtype=np.uint64)
    engines, firsts = _make_ranks(sizes, S, match, V)
    comms = CommSet.loopback(engines, firsts, max_pairs_per_rank=[300, 1000, 500])
    try:
        cap = comms.max_pairs_per_rank
        assert cap == 1024 and comms.caps == [1024] * 3  # the largest any rank asked for, in granules of 256
        prev = np.zeros(G, dtype=np.uint64)
        # pass 1: every group changes (all three ranks over cap); pass 2: only rank 1 over cap; pass 3: nobody
        for k, per_rank in enumerate([None, (200, 4000, 900), (10, 1000, 0)]):
            if per_rank is not None:
                parts = []
                for f, s, m in zip(firsts, sizes, per_rank):
                    r = np.zeros(m, dtype=raft_amd.ACK_DTYPE)
                    r["group"] = f + rng.choice(s, size=m, replace=False)
                    r["slot"] = 0
                    r["index"] = (k + 1) << 42
                    parts.append(r)
                    # two of three voters at the new index make it the quorum index
                    r2 = r.copy()
                    r2["slot"] = 1
                    parts.append(r2)
                recs = np.concatenate(parts)
                for e, part in zip(engines, _route(recs, firsts, sizes)):
                    e.scatter_acks(part)
                oracle.scatter_acks(match, recs)
            after = oracle.batch_committed(match, uniform_voters=V)
            ch = np.flatnonzero(after != prev)
            prev = after
            per = [ch[(ch >= f) & (ch < f + s)] for f, s in zip(firsts, sizes)]
            want = np.concatenate([p[:cap] for p in per])
            outs, counts, totals = comms.commit_allgather()
            for r in range(3):
                assert [int(c) for c in counts[r]] == [len(p) for p in per] and totals[r] == len(ch)
                assert np.array_equal(outs[r]["group"], want.astype(np.uint64)), (k, r)
                assert np.array_equal(outs[r]["commit"], after[want]), (k, r)
            if per_rank is not None:
                assert [len(p) for p in per] == list(per_rank)
    finally:
        _close(comms, engines)


def test_loopback_dense_exchange_with_ranks_of_different_plane_strides(oracle):
    """SURVEY.md 8(e) "dense fallback": one all-gather of the commit planes, [rank][width], width = the largest
    plane stride — a rank with a shorter plane sends through its padded copy."""
    import torch
    sizes = np.array([5_000, 70_001, 4_096, 12_289])
    S = 6
    firsts, G = _bounds(sizes)
    rng = np.random.default_rng(9)
    match = rng.integers(0, 1 << 40, size=(S, G), dtype=np.uint64)
    hi = (1 << S) - 1
    m0 = rng.integers(1, hi, size=G, dtype=np.uint64, endpoint=True).astype(np.uint8)
    m1 = np.where(rng.random(G) < 0.3, rng.integers(0, hi, size=G, dtype=np.uint64, endpoint=True), 0).astype(np.uint8)
    engines, firsts = _make_ranks(sizes, S, match, 0, m0, m1)
    comms = CommSet.loopback(engines, firsts)
    try:
        n = len(sizes)
        width = comms.dense_width()
        assert width == 73_728  # 70 001 rounded up to the 4096-element plane granule
        recs = _acks(rng, G, S, 3 * G, 1 << 41)
        for e, part in zip(engines, _route(recs, firsts, sizes)):
            e.scatter_acks(part)
        oracle.scatter_acks(match, recs)
        want = oracle.batch_committed(match, m0, m1)
        planes, w = comms.commit_allgather_dense()
        assert w == width
        for r in range(n):
            assert planes[r].shape == (n, width)
            for k, (f, s) in enumerate(zip(firsts, sizes)):
                assert np.array_equal(planes[r][k, :s], want[f:f + s]), (r, k)
        # device / page-locked / pageable destinations side by side, after more acks
        recs = _acks(rng, G, S, G, 1 << 42)
        for e, part in zip(engines, _route(recs, firsts, sizes)):
            e.scatter_acks(part)
        oracle.scatter_acks(match, recs)
        want = oracle.batch_committed(match, m0, m1)
        dev = torch.zeros(n * width, dtype=torch.int64, device="cuda:0")
        pin = PinnedArray(n * width, np.uint64)
        page = [np.empty(n * width, dtype=np.uint64) for _ in range(2)]
        comms.commit_allgather_dense(dsts=[dev.data_ptr(), pin.array.ctypes.data, page[0].ctypes.data, page[1].ctypes.data],
                                     dst_elems=[n * width] * n)
        for got in (dev.cpu().numpy().view(np.uint64), pin.array, page[0], page[1]):
            got = got.reshape(n, width)
            for k, (f, s) in enumerate(zip(firsts, sizes)):
                assert np.array_equal(got[k, :s], want[f:f + s])
        pin.free()
        # the term-guarded pass: the planes are raftLog.committed afterwards
        tf = np.full(G, 1 << 43, dtype=np.uint64)
        tl = np.full(G, (1 << 43) + (1 << 19), dtype=np.uint64)
        for e, f, s in zip(engines, firsts, sizes):
            e.upload_term_range(tf[f:f + s], tl[f:f + s])
        recs = _acks(rng, G, S, 4 * G, 1 << 43)
        for e, part in zip(engines, _route(recs, firsts, sizes)):
            e.scatter_acks(part)
        oracle.scatter_acks(match, recs)
        mci = oracle.batch_committed(match, m0, m1)
        adv = (mci > want) & (mci >= tf) & (mci <= tl)
        committed = np.where(adv, mci, want)
        planes, _ = comms.commit_allgather_dense(advance=True)
        for r in range(n):
            for k, (f, s) in enumerate(zip(firsts, sizes)):
                assert np.array_equal(planes[r][k, :s], committed[f:f + s])
    finally:
        _close(comms, engines)


def _cfg3_masks(G, seed):
    r = gu.splitmix64_array(seed, G)
    kind = r % np.uint64(10)
    m0 = np.empty(G, dtype=np.uint8)
    m1 = np.empty(G, dtype=np.uint8)
    joint = kind < 8
    m0[joint], m1[joint] = 0b111110, 0b011111
    nj = kind == 8
    m0[nj], m1[nj] = 0b011111, 0
    asym = kind == 9
    m0[asym], m1[asym] = 0b011111, 0b000111
    return m0, m1


@pytest.mark.parametrize("world", [2, 8])
def test_cfg3_sharded_over_loopback_ranks_equals_one_engine_and_the_oracle(oracle, world):
    """BASELINE cfg3's shape (joint 5 in / 5 out / overlap 4, S = 6) at 2^21 groups, sharded with
    multigpu.shard_range exactly as bench.py --config cfg3 --gpus N shards it."""
    G, S = 1 << 21, 6
    match = gu.dist_a(0x5EED0003, S, G)
    m0, m1 = _cfg3_masks(G, 0x5EED0003)
    want = oracle.batch_committed(match, m0, m1, threads=0)
    with raft_amd.Engine(G, S) as one:
        one.upload_config(m0, m1)
        one.upload_match(match)
        pairs_one, n_one = one.commit_changed()
    sizes = np.array([multigpu.shard_range(G, r, world)[1] for r in range(world)])
    engines, firsts = _make_ranks(sizes, S, match, 0, m0, m1)
    assert [int(f) for f in firsts] == [multigpu.shard_range(G, r, world)[0] for r in range(world)]
    comms = CommSet.loopback(engines, firsts)
    try:
        outs, counts, totals = comms.commit_allgather()
        ch = np.flatnonzero(want != 0)
        for r in range(world):
            assert totals[r] == n_one == len(ch)
            assert np.array_equal(outs[r], pairs_one)
            assert np.array_equal(outs[r]["group"], ch.astype(np.uint64)) and np.array_equal(outs[r]["commit"], want[ch])
    finally:
        _close(comms, engines)


@pytest.mark.parametrize("world", [4])
def test_cfg5_sharded_over_loopback_ranks_equals_one_engine_and_the_oracle(oracle, world):
    """BASELINE cfg5's shape (7 voters + 2 learners, 4 acks per group, stale ones among them) at 2^20 groups:
    acks routed with multigpu.route_acks, scatter + commit per rank, one exchange."""
    G, S, V = 1 << 20, 9, 7
    match = gu.dist_b(0x5EED0005, S, G) & np.uint64((1 << 63) - 1)
    rng = np.random.default_rng(0x5EED0005)
    M = 4 * G
    recs = np.zeros(M, dtype=raft_amd.ACK_DTYPE)
    recs["group"] = rng.integers(0, G, size=M, dtype=np.uint32)
    recs["slot"] = rng.integers(0, S, size=M, dtype=np.uint32)
    old = match[recs["slot"], recs["group"]]
    recs["index"] = old + rng.integers(0, 64, size=M, dtype=np.uint64)
    stale = rng.random(M) < 0.05
    recs["index"][stale] = old[stale] >> np.uint64(1)
    before = oracle.batch_committed(match, uniform_voters=V, threads=0)
    sizes = np.array([multigpu.shard_range(G, r, world)[1] for r in range(world)])
    engines, firsts = _make_ranks(sizes, S, match, V)
    for e, f, s in zip(engines, firsts, sizes):
        e.upload_commit(before[f:f + s])
    comms = CommSet.loopback(engines, firsts)
    try:
        with raft_amd.Engine(G, S) as one:
            one.set_uniform_config(V)
            one.upload_match(match)
            one.upload_commit(before)
            one.scatter_acks(recs)
            pairs_one, n_one = one.commit_changed()
        for e, part in zip(engines, multigpu.route_acks(recs, G, world)):
            e.scatter_acks(part)
        oracle.scatter_acks(match, recs)
        after = oracle.batch_committed(match, uniform_voters=V, threads=0)
        ch = np.flatnonzero(after != before)
        outs, counts, totals = comms.commit_allgather()
        for r in range(world):
            assert totals[r] == n_one == len(ch)
            assert np.array_equal(outs[r], pairs_one)
            assert np.array_equal(outs[r]["group"], ch.astype(np.uint64)) and np.array_equal(outs[r]["commit"], after[ch])
        for e, f, s in zip(engines, firsts, sizes):
            assert np.array_equal(e.read_match(), match[:, f:f + s])
    finally:
        _close(comms, engines)


def test_loopback_refuses_a_partial_set_of_ranks():
    engines = [raft_amd.Engine(1000, 3, device=0) for _ in range(3)]
    for e in engines:
        e.set_uniform_config(3)
    comms = CommSet.loopback(engines, [0, 1000, 2000])
    try:
        import ctypes as C
        L = raft_amd.load_library()
        arr = (C.c_void_p * 2)(comms._hs[0], comms._hs[1])
        totals = (C.c_uint64 * 2)()
        rc = L.rafthip_commit_allgather_many(arr, 2, 0, None, None, None, totals)
        assert rc == -22 and b"loopback" in L.rafthip_last_error()
    finally:
        _close(comms, engines)
