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
