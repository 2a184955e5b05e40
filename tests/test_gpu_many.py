"""rafthip_commit_many: ProgressTracker.Committed (tracker/tracker.go:179-181) for the shards of
several engines in one launch — bit-exact against the oracle and against rafthip_commit per engine."""
import numpy as np
import pytest

import oracle
import raft_amd
from raft_amd.capi import EngineSet

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V,S,shared_stream", [(5, 5, True), (3, 3, False), (7, 9, True), (1, 4, False)])
def test_commit_many_equals_oracle(V, S, shared_stream):
    import torch
    rng = np.random.default_rng(40 + V)
    sizes = [1, 2047, 2048, 2049, 100_000, 4097, 65_536]  # ragged shards: partial tiles, one group, whole tiles
    stream = torch.cuda.Stream(device=0) if shared_stream else None
    engines, mats = [], []
    for G in sizes:
        e = raft_amd.Engine(G, S, device=0, stream=stream.cuda_stream if stream else None)
        m = rng.integers(0, 1 << 63, size=(S, G), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(S, G), dtype=np.uint64)
        e.set_uniform_config(V)
        e.upload_match(m)
        engines.append(e)
        mats.append(m)
    try:
        es = EngineSet(engines)
        es.commit()
        for e, m in zip(engines, mats):
            assert np.array_equal(e.read_commit(), oracle.batch_committed(m, uniform_voters=V))
        # new acks on some engines (each on its own stream when not shared), then only the first three in one launch
        for e, m in zip(engines[:3], mats[:3]):
            m[:] = rng.integers(0, 1 << 40, size=m.shape, dtype=np.uint64)
            e.upload_match(m)
        before = [e.read_commit() for e in engines]
        es.commit(3)
        for i, (e, m) in enumerate(zip(engines, mats)):
            want = oracle.batch_committed(m, uniform_voters=V) if i < 3 else before[i]
            assert np.array_equal(e.read_commit(), want)
        assert engines[0].stats()["commit_launches"] == 2
    finally:
        for e in engines:
            e.close()


def test_commit_many_more_shards_than_one_launch_holds_and_mixed_configs():
    rng = np.random.default_rng(5)
    engines, mats = [], []
    for k in range(37):  # more than kManyMax = 32 shards: two launches
        G = int(rng.integers(1, 9000))
        e = raft_amd.Engine(G, 5, device=0)
        m = rng.integers(0, 1 << 50, size=(5, G), dtype=np.uint64)
        e.set_uniform_config(5)
        e.upload_match(m)
        engines.append(e)
        mats.append(m)
    try:
        EngineSet(engines).commit()
        for e, m in zip(engines, mats):
            assert np.array_equal(e.read_commit(), oracle.batch_committed(m, uniform_voters=5))
        # a set that one launch cannot serve (different voter counts, a masked config): served engine by engine
        engines[1].set_uniform_config(3)
        G2 = engines[2].G
        in_mask = rng.integers(1, 32, size=G2).astype(np.uint8)
        out_mask = rng.integers(0, 32, size=G2).astype(np.uint8)
        engines[2].upload_config(in_mask, out_mask)
        EngineSet(engines[:4]).commit()
        assert np.array_equal(engines[1].read_commit(), oracle.batch_committed(mats[1], uniform_voters=3))
        assert np.array_equal(engines[2].read_commit(), oracle.batch_committed(mats[2], in_mask, out_mask))
        assert np.array_equal(engines[3].read_commit(), oracle.batch_committed(mats[3], uniform_voters=5))
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("advance", [False, True])
def test_commit_report_many_equals_the_single_engine_calls(advance):
    """rafthip_commit_report_many (one commit launch + one pack launch + one wait for all shards) against
    rafthip_commit_changed / rafthip_commit_advance on twin engines, pass after pass: ragged shards, pageable and
    device destinations, clipped buffers, a pass in which nothing changes."""
    import torch
    rng = np.random.default_rng(77 + advance)
    sizes = [1, 2047, 2049, 150_000, 4097, 70_000]
    V, S = 5, 5
    many, twins, mats = [], [], []
    for G in sizes:
        m = rng.integers(1, 1 << 30, size=(S, G), dtype=np.uint64)
        pair = []
        for _ in range(2):
            e = raft_amd.Engine(G, S, device=0)
            e.set_uniform_config(V)
            e.upload_match(m)
            if advance:
                e.upload_commit(np.zeros(G, dtype=np.uint64))
                e.upload_term_range(np.full(G, 1 << 29, dtype=np.uint64), np.full(G, 1 << 40, dtype=np.uint64))
            pair.append(e)
        many.append(pair[0])
        twins.append(pair[1])
        mats.append(m)
    # the oracle's view of every shard: what its commit plane holds, and the pairs a pass must report
    held = [np.zeros(G, dtype=np.uint64) for G in sizes]

    def oracle_pairs(i):
        mci = oracle.batch_committed(mats[i], uniform_voters=V)
        if advance:  # raftLog.maybeCommit on the [term_first, term_last] summary (log.go:447-456)
            moved = (mci > held[i]) & (mci >= np.uint64(1 << 29)) & (mci <= np.uint64(1 << 40))
            held[i] = np.where(moved, mci, held[i])
        else:
            moved = mci != held[i]
            held[i] = mci
        ch = np.flatnonzero(moved)
        return ch.astype(np.uint64), held[i][ch]

    try:
        es = EngineSet(many)
        for step in range(4):
            if step == 2:
                pass  # nothing new: no pair anywhere
            elif step:
                for m, a, b in zip(mats, many, twins):
                    bump = rng.random(m.shape) < 0.05
                    m[bump] += np.uint64(1 << 29)
                    a.upload_match(m)
                    b.upload_match(m)
            room = [max(1, G // 3) for G in sizes] if step == 3 else None  # clipped on the last pass
            outs, counts = es.commit_report(advance=advance, max_pairs=room)
            for i, (b, G) in enumerate(zip(twins, sizes)):
                want, n = (b.commit_advance if advance else b.commit_changed)(max_pairs=room[i] if room else None)
                assert counts[i] == n, (step, i)
                assert np.array_equal(outs[i], want), (step, i)
                assert np.array_equal(many[i].read_commit(), b.read_commit()), (step, i)
                og, oc = oracle_pairs(i)  # and against the oracle, not only the twin
                k = len(outs[i])
                assert counts[i] == len(og) and k == (min(len(og), room[i]) if room else len(og)), (step, i)
                assert np.array_equal(outs[i]["group"], og[:k]) and np.array_equal(outs[i]["commit"], oc[:k]), (step, i)
                assert np.array_equal(many[i].read_commit(), held[i]), (step, i)
            if step == 2:
                assert counts == [0] * len(sizes)
        # device destinations, count-only for one shard
        for m, a, b in zip(mats, many, twins):
            m += np.uint64(1 << 31)
            a.upload_match(m)
            b.upload_match(m)
        bufs = [torch.zeros(2 * G, dtype=torch.int64, device="cuda:0") for G in sizes]
        room = list(sizes)
        room[1] = 0
        dsts = [buf.data_ptr() if r else None for buf, r in zip(bufs, room)]
        none, counts = es.commit_report(advance=advance, max_pairs=room, dsts=dsts)
        assert none is None
        for i, b in enumerate(twins):
            want, n = (b.commit_advance if advance else b.commit_changed)()
            og, oc = oracle_pairs(i)
            assert counts[i] == n == len(og)
            if room[i]:
                got = bufs[i][:2 * n].cpu().numpy().view(raft_amd.PAIR_DTYPE)
                assert np.array_equal(got, want)
                assert np.array_equal(got["group"], og) and np.array_equal(got["commit"], oc)
    finally:
        for e in many + twins:
            e.close()


def test_commit_report_many_mixed_configs_fall_back_and_duplicates_are_refused():
    rng = np.random.default_rng(3)
    G = 5000
    m = rng.integers(1, 1 << 30, size=(6, G), dtype=np.uint64)
    a = raft_amd.Engine(G, 6, device=0)
    b = raft_amd.Engine(G, 6, device=0)
    try:
        a.set_uniform_config(5)
        m0 = np.full(G, 0b111111, dtype=np.uint8)
        m1 = np.where(rng.random(G) < 0.5, 0b000111, 0).astype(np.uint8)
        b.upload_config(m0, m1)
        a.upload_match(m)
        b.upload_match(m)
        outs, counts = EngineSet([a, b]).commit_report()
        wa = oracle.batch_committed(m, uniform_voters=5)
        wb = oracle.batch_committed(m, in_mask=m0, out_mask=m1)
        assert counts == [int((wa != 0).sum()), int((wb != 0).sum())]
        assert np.array_equal(outs[0]["commit"], wa[wa != 0]) and np.array_equal(outs[1]["commit"], wb[wb != 0])
        with pytest.raises(raft_amd.RafthipError):
            EngineSet([a, a]).commit_report()
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("S", [6, 3, 8])
def test_commit_many_and_report_many_with_masked_shards(S):
    """A host in the middle of a conf change: some shards carry per-group joint masks (quorum/joint.go:49-56),
    others still the uniform config.  Same S <= 8 everywhere -> one K2m launch; results against the oracle, shard by
    shard, for the plain pass, the changed report and the term-guarded report."""
    rng = np.random.default_rng(600 + S)
    sizes = [1, 2047, 2049, 90_000, 4097, 33_000]
    hi = (1 << S) - 1
    engines, mats, cfgs = [], [], []
    for k, G in enumerate(sizes):
        e = raft_amd.Engine(G, S, device=0)
        m = rng.integers(1, 1 << 40, size=(S, G), dtype=np.uint64)
        if k % 3 == 2:  # a shard that has not entered a joint config
            v = min(S, 5)
            e.set_uniform_config(v)
            cfgs.append(dict(uniform_voters=v))
        else:
            m0 = rng.integers(1, hi, size=G, dtype=np.uint64, endpoint=True).astype(np.uint8)
            m1 = np.where(rng.random(G) < (0.0 if k == 1 else 0.5), rng.integers(0, hi, size=G, dtype=np.uint64, endpoint=True), 0).astype(np.uint8)
            e.upload_config(m0, m1)
            cfgs.append(dict(in_mask=m0, out_mask=m1))
        e.upload_match(m)
        engines.append(e)
        mats.append(m)
    try:
        es = EngineSet(engines)
        es.commit()
        held = []
        for e, m, c in zip(engines, mats, cfgs):
            want = oracle.batch_committed(m, **c)
            assert np.array_equal(e.read_commit(), want)
            held.append(want)
        assert all(e.stats()["commit_launches"] == 1 for e in engines)
        # changed report after new acks
        for e, m in zip(engines, mats):
            bump = rng.random(m.shape) < 0.1
            m[bump] += np.uint64(1 << 41)
            e.upload_match(m)
        outs, counts = es.commit_report()
        for i, (e, m, c) in enumerate(zip(engines, mats, cfgs)):
            want = oracle.batch_committed(m, **c)
            ch = np.flatnonzero(want != held[i])
            assert counts[i] == len(ch), i
            assert np.array_equal(outs[i]["group"], ch.astype(np.uint64)) and np.array_equal(outs[i]["commit"], want[ch]), i
            assert np.array_equal(e.read_commit(), want)
            held[i] = want
        # the term-guarded report: only groups whose new index lies inside [term_first, term_last] move
        for e, m in zip(engines, mats):
            e.upload_term_range(np.full(e.G, 1 << 42, dtype=np.uint64), np.full(e.G, (1 << 42) + (1 << 41), dtype=np.uint64))
            bump = rng.random(m.shape) < 0.3
            m[bump] += np.uint64(1 << 42)
            e.upload_match(m)
        outs, counts = es.commit_report(advance=True)
        for i, (e, m, c) in enumerate(zip(engines, mats, cfgs)):
            mci = oracle.batch_committed(m, **c)
            adv = (mci > held[i]) & (mci >= np.uint64(1 << 42)) & (mci <= np.uint64((1 << 42) + (1 << 41)))
            ch = np.flatnonzero(adv)
            assert counts[i] == len(ch), i
            assert np.array_equal(outs[i]["group"], ch.astype(np.uint64)) and np.array_equal(outs[i]["commit"], mci[ch]), i
            assert np.array_equal(e.read_commit(), np.where(adv, mci, held[i]))
    finally:
        for e in engines:
            e.close()
