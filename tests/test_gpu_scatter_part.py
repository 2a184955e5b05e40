"""K5p — the partitioned form of rafthip_scatter_acks (raft_amd/csrc/scatter_part.h): Progress.MaybeUpdate's
Match part for a batch of acks (tracker/progress.go:168-177 via raft.go:1477), brought into tile order by two
LDS-staged partition passes and applied tile by tile in LDS.  Same result as the atomic form, bit for bit, and
both against the oracle applying the acks one by one: duplicates, stale acks, records without a Progress row,
ragged group counts, every tile width, batches smaller than a chunk, a batch piled onto one tile (the device-side
fallback to atomics), host and device records."""
import numpy as np
import pytest

import oracle
import raft_amd
from raft_amd.capi import FLAG_SCATTER_ATOMIC, FLAG_SCATTER_PARTITION

pytestmark = pytest.mark.gpu


def _recs(rng, G, S, n, match, bad=0.0, dup=0.0):
    recs = np.zeros(n, dtype=raft_amd.ACK_DTYPE)
    recs["group"] = rng.integers(0, G, size=n, dtype=np.uint32)
    recs["slot"] = rng.integers(0, S, size=n, dtype=np.uint32)
    old = match[recs["slot"], recs["group"]]
    recs["index"] = old + rng.integers(0, 64, size=n, dtype=np.uint64)
    stale = rng.random(n) < 0.05
    recs["index"][stale] = old[stale] >> np.uint64(1)
    if dup and n > 10:  # many records on few cells
        k = int(n * dup)
        src = rng.integers(0, n, size=k)
        dst = rng.integers(0, n, size=k)
        recs["group"][dst], recs["slot"][dst] = recs["group"][src], recs["slot"][src]
        recs["index"][dst] = rng.integers(0, 1 << 62, size=k, dtype=np.uint64)
    if bad:
        b = rng.random(n) < bad
        recs["group"][b] = rng.choice(np.array([G, G + 1, 0xFFFFFFFF], dtype=np.uint32), size=int(b.sum()))
        b2 = rng.random(n) < bad
        recs["slot"][b2] = rng.choice(np.array([S, 64, 0xFFFFFFFF], dtype=np.uint32), size=int(b2.sum()))
    return recs


def _check(G, S, batches, seed, device_records=False, bad=0.0, dup=0.0):
    rng = np.random.default_rng(seed)
    match = rng.integers(0, 1 << 62, size=(S, G), dtype=np.uint64)
    want = match.copy()
    with raft_amd.Engine(G, S, flags=FLAG_SCATTER_PARTITION) as p, raft_amd.Engine(G, S, flags=FLAG_SCATTER_ATOMIC) as a:
        p.upload_match(match)
        a.upload_match(match)
        skipped = 0
        for n in batches:
            recs = _recs(rng, G, S, n, want, bad=bad, dup=dup)
            if device_records:
                import torch
                d = torch.from_numpy(recs.view(np.uint8)).to("cuda:0")
                p.scatter_acks(d.data_ptr(), n)
                a.scatter_acks(d.data_ptr(), n)
                p.synchronize()
                a.synchronize()
            else:
                p.scatter_acks(recs)
                a.scatter_acks(recs)
            skipped += oracle.scatter_acks(want, recs)
            got = p.read_match()
            assert np.array_equal(got, want), (G, S, n)
            assert np.array_equal(a.read_match(), want)
        assert p.stats()["acks_rejected"] == skipped == a.stats()["acks_rejected"]
        assert p.stats()["acks_applied"] == sum(batches)


@pytest.mark.parametrize("G,S", [(1, 1), (1000, 5), (4097, 9), (100_003, 9), (70_000, 3), (33_000, 18), (9_000, 36), (5_000, 64)])
def test_partitioned_scatter_equals_the_oracle(G, S):
    # batches below one chunk, of exactly one chunk, of many chunks with a ragged tail
    _check(G, S, [1, 100, 4096, 4097, 50_000, 3 * G + 17], seed=G + S, dup=0.1)


def test_partitioned_scatter_many_buckets_and_device_records():
    # 300 001 groups at 1024 per tile = 293 tiles = 3 buckets; records already on the device
    _check(300_001, 7, [1_200_000, 5, 400_000], seed=9, device_records=True, dup=0.05)


def test_partitioned_scatter_drops_and_counts_records_without_a_progress_row():
    _check(50_000, 9, [200_000, 10, 4096], seed=10, bad=0.02)
    _check(50_000, 9, [100_000], seed=11, bad=0.02, device_records=True)


def test_partitioned_scatter_piled_onto_one_tile_takes_the_atomic_form():
    """More than 2^17 records in one tile: the scan's flag sends the batch through K5's atomics on the device."""
    G, S = 200_000, 5
    rng = np.random.default_rng(12)
    match = rng.integers(0, 1 << 40, size=(S, G), dtype=np.uint64)
    want = match.copy()
    n = 300_000
    recs = np.zeros(n, dtype=raft_amd.ACK_DTYPE)
    recs["group"] = rng.integers(3000, 3100, size=n)  # one tile of 1024 groups
    recs["slot"] = rng.integers(0, S, size=n)
    recs["index"] = rng.integers(0, 1 << 50, size=n, dtype=np.uint64)
    with raft_amd.Engine(G, S, flags=FLAG_SCATTER_PARTITION) as p:
        p.upload_match(match)
        p.scatter_acks(recs)
        oracle.scatter_acks(want, recs)
        assert np.array_equal(p.read_match(), want)
        # and the next, unskewed batch goes back to the partitioned form on clean scratch
        recs2 = _recs(rng, G, S, 500_000, want)
        p.scatter_acks(recs2)
        oracle.scatter_acks(want, recs2)
        assert np.array_equal(p.read_match(), want)


def test_scatter_flags_exclude_each_other():
    with pytest.raises(raft_amd.RafthipError):
        raft_amd.Engine(100, 3, flags=FLAG_SCATTER_ATOMIC | FLAG_SCATTER_PARTITION)


def test_default_engine_picks_the_partitioned_form_for_big_batches_only():
    """Auto mode: a 2^22-record batch over 2^20 x 9 cells (one ack per 2.25 cells) is partitioned, a small one is
    not; both equal the oracle (the choice is visible only in speed)."""
    G, S = 1 << 20, 9
    rng = np.random.default_rng(13)
    match = rng.integers(0, 1 << 62, size=(S, G), dtype=np.uint64)
    want = match.copy()
    import torch
    with raft_amd.Engine(G, S) as e:
        e.upload_match(match)
        for n in (1 << 22, 1000, 1 << 21):
            recs = _recs(rng, G, S, n, want)
            d = torch.from_numpy(recs.view(np.uint8)).to("cuda:0")
            e.scatter_acks(d.data_ptr(), n)
            e.synchronize()
            oracle.scatter_acks(want, recs)
            assert np.array_equal(e.read_match(), want)
