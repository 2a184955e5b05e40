"""The stateless entry points quorum.BatchCommittedIndex / BatchVoteResult bind to
(rafthip_batch_committed_index / _vote_result; reference: quorum/joint.go:49-75 looped over
groups): pageable host planes in, host results out, staged in chunks through page-locked slots.
Bit-exact against the oracle across chunk boundaries, at every mask width, from several threads."""
import threading

import numpy as np
import pytest

import oracle
import raft_amd

pytestmark = pytest.mark.gpu


def _masks(rng, S, G, joint_share=0.4):
    md = raft_amd.mask_dtype(S)
    hi = (1 << S) - 1
    in_mask = rng.integers(0, hi, size=G, dtype=np.uint64, endpoint=True).astype(md)
    out_mask = np.where(rng.random(G) < joint_share, rng.integers(0, hi, size=G, dtype=np.uint64, endpoint=True), 0).astype(md)
    return in_mask, out_mask


@pytest.mark.parametrize("S,G", [(5, 1), (5, 4095), (5, 65_536), (5, 65_537), (5, 1_048_576), (3, 700_001),
                                 (8, 300_000), (12, 150_001), (33, 40_000), (64, 9_000)])
def test_batch_committed_index_host_planes(S, G):
    rng = np.random.default_rng(S * 1000 + G % 997)
    match = rng.integers(0, 1 << 63, size=(S, G), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(S, G), dtype=np.uint64)
    in_mask, out_mask = _masks(rng, S, G)
    got = raft_amd.batch_committed_index(match, in_mask, out_mask)
    assert np.array_equal(got, oracle.batch_committed(match, in_mask, out_mask))
    got = raft_amd.batch_committed_index(match, in_mask)  # non-joint: the outgoing half is the zero config
    assert np.array_equal(got, oracle.batch_committed(match, in_mask, np.zeros_like(in_mask)))


@pytest.mark.parametrize("S,G", [(5, 1), (5, 1_000_003), (7, 524_288), (16, 200_001), (40, 50_000)])
def test_batch_vote_result_host_planes(S, G):
    rng = np.random.default_rng(S * 77 + G % 991)
    votes = rng.integers(0, 3, size=(S, G)).astype(np.uint8)
    in_mask, out_mask = _masks(rng, S, G)
    got = raft_amd.batch_vote_result(votes, in_mask, out_mask)
    assert np.array_equal(got, oracle.batch_tally_votes(votes, in_mask, out_mask)[0])
    got = raft_amd.batch_vote_result(votes, in_mask)
    assert np.array_equal(got, oracle.batch_tally_votes(votes, in_mask, np.zeros_like(in_mask))[0])


def test_batch_committed_index_into_a_result_buffer_the_caller_keeps():
    """A tick loop reuses its result slice: the same array filled by call after call, at sizes that take the
    pipelined path (helper threads kept between calls) and the in-line one."""
    rng = np.random.default_rng(77)
    for G in (300_000, 300_000, 5_000, 300_000):
        match = rng.integers(0, 1 << 62, size=(5, G), dtype=np.uint64)
        in_mask, out_mask = _masks(rng, 5, G)
        out = np.full(G, 0xDEAD, dtype=np.uint64)
        got = raft_amd.batch_committed_index(match, in_mask, out_mask, out=out)
        assert got is out
        assert np.array_equal(out, oracle.batch_committed(match, in_mask, out_mask))
    with pytest.raises(ValueError):
        raft_amd.batch_committed_index(match, in_mask, out_mask, out=np.zeros(G, dtype=np.uint32))
    with pytest.raises(ValueError):
        raft_amd.batch_committed_index(match, in_mask, out_mask, out=np.zeros(G + 1, dtype=np.uint64))


def test_batch_calls_from_several_threads_and_shrinking_sizes():
    """Each call takes a staging context from the pool; calls of different shapes follow one another
    (a smaller call finds a larger call's stale slots) and overlap across threads."""
    errs = []

    def worker(seed):
        try:
            rng = np.random.default_rng(seed)
            for S, G in [(5, 400_000), (5, 5_000), (9, 120_000), (2, 17)]:
                match = rng.integers(0, 1 << 62, size=(S, G), dtype=np.uint64)
                in_mask, out_mask = _masks(rng, S, G)
                got = raft_amd.batch_committed_index(match, in_mask, out_mask)
                if not np.array_equal(got, oracle.batch_committed(match, in_mask, out_mask)):
                    errs.append((seed, S, G))
        except Exception as ex:  # noqa: BLE001
            errs.append((seed, repr(ex)))

    ts = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_batch_committed_index_device_pointers_still_served():
    import torch
    rng = np.random.default_rng(9)
    S, G = 5, 70_000
    match = rng.integers(0, 1 << 62, size=(S, G), dtype=np.uint64)
    in_mask, out_mask = _masks(rng, S, G)
    d_match = torch.from_numpy(match.view(np.int64)).to("cuda:0")
    d_in, d_out = torch.from_numpy(in_mask).to("cuda:0"), torch.from_numpy(out_mask).to("cuda:0")
    res = torch.zeros(G, dtype=torch.int64, device="cuda:0")
    from raft_amd.capi import _check, _ptr, load_library
    _check(load_library().rafthip_batch_committed_index(0, G, S, _ptr(d_match), _ptr(d_in), _ptr(d_out), _ptr(res)))
    assert np.array_equal(res.cpu().numpy().view(np.uint64), oracle.batch_committed(match, in_mask, out_mask))
