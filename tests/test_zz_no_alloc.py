"""rafthip_commit_changed_static — the allocation-free variant of
rafthip_commit_changed (include/rafthip.h): same pairs, same commit plane, scratch
in the library's static device array.  Through the C ABI (ctypes), bit-exact
against rafthip_commit_changed on twin engines AND against the oracle's changed set
(oracle.batch_committed before / after the acks), two calls in a row on one GPU, the
second on a smaller input (the array keeps the first call's contents)."""
import numpy as np
import pytest

import oracle
import raft_amd
from raft_amd.capi import PAIR_DTYPE, PinnedArray, RafthipError

pytestmark = pytest.mark.gpu


def _twin(G, S, seed, joint):
    """Two engines with the same config and Match planes."""
    rng = np.random.default_rng(seed)
    match = rng.integers(0, 1 << 40, size=(S, G), dtype=np.uint64)
    a, b = raft_amd.Engine(G, S, device=0), raft_amd.Engine(G, S, device=0)
    md = a.mask_dtype
    hi = (1 << S) - 1
    in_mask = rng.integers(1, hi, size=G, dtype=np.uint64, endpoint=True).astype(md)
    out_mask = np.where(rng.random(G) < 0.4, rng.integers(0, hi, size=G, dtype=np.uint64, endpoint=True), 0).astype(md)
    for e in (a, b):
        if joint:
            e.upload_config(in_mask, out_mask)
        else:
            e.set_uniform_config(min(S, 5))
        e.upload_match(match)
    # the oracle's shadow of engine a: planes, config, and what its commit plane holds
    a.shadow = dict(match=match, prev=np.zeros(G, dtype=np.uint64),
                    cfg=dict(in_mask=in_mask, out_mask=out_mask) if joint else dict(uniform_voters=min(S, 5)))
    return a, b, match, rng


def _round(a, b, recs, dst_kind):
    """Scatter the same acks into both engines; a reports through the static entry point."""
    import torch
    a.scatter_acks(recs)
    b.scatter_acks(recs)
    want, n_want = b.commit_changed()
    if dst_kind == "pinned":
        pin = PinnedArray(a.G, PAIR_DTYPE)
        n = a.commit_changed_static(pin.array.ctypes.data, a.G)
        got = pin.array[:n].copy()
        pin.free()
    else:
        buf = torch.zeros(2 * a.G, dtype=torch.int64, device="cuda:0")
        n = a.commit_changed_static(buf.data_ptr(), a.G)
        got = buf[:2 * n].cpu().numpy().view(PAIR_DTYPE)
    assert n == n_want
    assert np.array_equal(got["group"], want["group"]) and np.array_equal(got["commit"], want["commit"])
    assert np.array_equal(a.read_commit(), b.read_commit())
    sh = a.shadow
    assert oracle.scatter_acks(sh["match"], recs) == 0
    after = oracle.batch_committed(sh["match"], **sh["cfg"])
    ch = np.flatnonzero(after != sh["prev"])
    sh["prev"] = after
    assert n == len(ch) and np.array_equal(got["group"], ch.astype(np.uint64)) and np.array_equal(got["commit"], after[ch])
    return n


def _acks(rng, G, S, n, lo):
    recs = np.zeros(n, dtype=raft_amd.ACK_DTYPE)
    recs["group"] = rng.integers(0, G, size=n)
    recs["slot"] = rng.integers(0, S, size=n)
    recs["index"] = rng.integers(lo, lo + (1 << 20), size=n, dtype=np.uint64)
    return recs


@pytest.mark.parametrize("S,joint", [(5, False), (6, True), (12, True), (40, True)])
def test_static_equals_commit_changed_twice_second_smaller(S, joint):
    G1, G2 = 300_001, 4_999  # the second call is the smaller one and finds the first call's scratch
    a1, b1, _, rng = _twin(G1, S, 11 + S, joint)
    a2, b2, _, _ = _twin(G2, S, 23 + S, joint)
    try:
        n1 = _round(a1, b1, _acks(rng, G1, S, G1 // 2, 1 << 41), "pinned")   # first pass: every group changes (commit was 0)
        assert n1 > G1 // 2
        n2 = _round(a2, b2, _acks(rng, G2, S, G2 // 3, 1 << 41), "device")
        assert 0 < n2 <= G2
        # a third and fourth pass on each: few groups change; nothing changes
        n3 = _round(a1, b1, _acks(rng, G1, S, 1000, 1 << 42), "device")
        assert 0 < n3 <= 1000
        assert _round(a2, b2, np.zeros(0, dtype=raft_amd.ACK_DTYPE), "pinned") == 0
    finally:
        for e in (a1, b1, a2, b2):
            e.close()


def test_static_truncates_and_counts_like_commit_changed():
    import torch
    G, S = 50_000, 5
    a, b, _, rng = _twin(G, S, 5, False)
    try:
        want, n_want = b.commit_changed()
        buf = torch.full((2 * 100 + 2,), -1, dtype=torch.int64, device="cuda:0")
        n = a.commit_changed_static(buf.data_ptr(), 100)
        got = buf.cpu().numpy()
        assert n == n_want == G
        assert np.array_equal(got[:200].view(PAIR_DTYPE)["commit"], want["commit"][:100])
        assert (got[200:] == -1).all()  # nothing written past max_pairs
        assert a.commit_changed_static(None, 0) == 0  # count only
    finally:
        a.close()
        b.close()


def test_static_rejects_what_it_cannot_serve():
    with raft_amd.Engine(1000, 5, device=0) as e:
        out = np.zeros(1000, dtype=PAIR_DTYPE)
        with pytest.raises(RafthipError, match="set a config first"):
            e.commit_changed_static(out.ctypes.data, 1000)
        e.set_uniform_config(3)
        with pytest.raises(RafthipError, match="page-locked"):
            e.commit_changed_static(out.ctypes.data, 1000)  # pageable host memory
    big = (1 << 24) + 1
    with raft_amd.Engine(big, 1, device=0) as e:
        e.set_uniform_config(1)
        with pytest.raises(RafthipError, match="at most"):
            e.commit_changed_static(None, 0)
