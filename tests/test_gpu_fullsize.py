"""GPU parity at the sizes BASELINE.json's configs name.

The oracle's batched views run at >10^7 groups/s on the host cores, so at
these sizes the whole output is still compared bit-exact, not only sampled;
size-independent properties (half swap, joint-with-self, monotonicity under
acks, idempotent re-commit) are checked on top.
"""
import numpy as np
import pytest

import golden_util as gu
import raft_amd

pytestmark = pytest.mark.gpu


def test_cfg2_1m_groups_3_voters(oracle):
    """1M groups x 3 voters batched CommittedIndex, dist A and dist B."""
    G, V = 1 << 20, 3
    for name, match in (("A", gu.dist_a(0x5EED0002, V, G)), ("B", gu.dist_b(0x5EED0002, V, G))):
        with raft_amd.Engine(G, V) as e:
            e.set_uniform_config(V)
            e.upload_match(match)
            e.commit()
            got = e.read_commit()
        assert np.array_equal(got, oracle.batch_committed(match, uniform_voters=V, threads=0)), name
        # stateless entry point: what quorum.BatchCommittedIndex binds to
        full = np.full(G, 0b111, dtype=np.uint8)
        assert np.array_equal(raft_amd.batch_committed_index(match, full), got), name


def cfg3_masks(G, seed):
    """80% joint 5 in / 5 out / overlap 4 (S=6), 10% non-joint V=5, 10% asymmetric 3 -> 5."""
    r = gu.splitmix64_array(seed, G)
    kind = r % np.uint64(10)
    m0 = np.empty(G, dtype=np.uint8)
    m1 = np.empty(G, dtype=np.uint8)
    joint = kind < 8
    m0[joint], m1[joint] = 0b111110, 0b011111   # slot 0 replaced by slot 5
    nj = kind == 8
    m0[nj], m1[nj] = 0b011111, 0
    asym = kind == 9
    m0[asym], m1[asym] = 0b011111, 0b000111     # outgoing 3 voters, incoming 5
    return m0, m1


def test_cfg3_16m_groups_joint(oracle):
    """16M groups x 5 voters with confchange-style joint configs (one GPU holds all
    16M; the multi-GPU form shards this same range, tests/test_multigpu_cpu.py)."""
    G, S = 1 << 24, 6
    match = gu.dist_a(0x5EED0003, S, G)
    m0, m1 = cfg3_masks(G, 0x5EED0003)
    want = oracle.batch_committed(match, m0, m1, threads=0)
    with raft_amd.Engine(G, S) as e:
        e.upload_config(m0, m1)
        e.upload_match(match)
        e.commit()
        got = e.read_commit()
        assert np.array_equal(got, want)
        # halves swap symmetrically (datadriven_test.go:219)
        e.upload_config(m1, m0)
        e.commit()
        assert np.array_equal(e.read_commit(), want)
        # joint-with-self == majority of the incoming half (datadriven_test.go:183)
        e.upload_config(m0, m0)
        e.commit()
        self_joint = e.read_commit()
        e.upload_config(m0, np.zeros_like(m0))
        e.commit()
        assert np.array_equal(self_joint, e.read_commit())
        # second pass reports no change
        _, n = e.commit_changed(max_pairs=0)
        assert n == 0
    del match
    # dist B at 4M groups: duplicates, zeros and >= 2^63 cells through the masked kernel
    G2 = 1 << 22
    match = gu.dist_b(0x5EED0003, S, G2)
    m0, m1 = cfg3_masks(G2, 7)
    with raft_amd.Engine(G2, S) as e:
        e.upload_config(m0, m1)
        e.upload_match(match)
        e.commit()
        assert np.array_equal(e.read_commit(), oracle.batch_committed(match, m0, m1, threads=0))


def test_cfg4_64m_groups_tallies(oracle):
    """64M groups, V=5: TallyVotes + QuorumActive + ReadIndex ack quorum."""
    G, V = 1 << 26, 5
    rng = np.random.default_rng(0x5EED0004)
    lut = np.array([2, 2, 1, 0, 0], dtype=np.uint8)  # 40% yes, 20% no, 40% missing
    votes = lut[rng.integers(0, 5, size=(V, G), dtype=np.uint8)]
    active = rng.integers(0, 32, size=G, dtype=np.uint8)
    acks = active[::-1].copy()
    with raft_amd.Engine(G, V) as e:
        e.set_uniform_config(V)
        e.upload_votes(votes)
        e.tally_votes()
        res, gr, rj = e.read_vote_result()
        ores, ogr, orj = oracle.batch_tally_votes(votes, uniform_voters=V)
        assert np.array_equal(res, ores) and np.array_equal(gr, ogr) and np.array_equal(rj, orj)
        del ores, ogr, orj
        # counts are consistent with the cells
        assert np.array_equal(gr, (votes == 2).sum(axis=0, dtype=np.uint8))
        assert np.bincount(res, minlength=4)[0] == 0
        del res, gr, rj, votes
        e.upload_active(active)
        e.quorum_active()
        assert np.array_equal(e.read_quorum_active(), oracle.batch_quorum_active(active, V, uniform_voters=V))
        e.upload_acks(acks)
        e.ack_quorum()
        assert np.array_equal(e.read_ack_quorum(), oracle.batch_ack_quorum(acks, V, uniform_voters=V))


def test_cfg5_scatter_then_commit_7_voters_2_learners(oracle):
    """7 voters + 2 learners: random MsgAppResp stream -> atomicMax -> commit pass."""
    G, S, V = 1 << 22, 9, 7
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
    del old, stale
    with raft_amd.Engine(G, S) as e:
        e.set_uniform_config(V)
        e.upload_match(match)
        e.commit()
        before = e.read_commit()
        assert np.array_equal(before, oracle.batch_committed(match, uniform_voters=V, threads=0))
        e.scatter_acks(recs)
        pairs, n = e.commit_changed()
        want_match = match.copy()
        assert oracle.scatter_acks(want_match, recs) == 0
        assert np.array_equal(e.read_match(), want_match)
        after = oracle.batch_committed(want_match, uniform_voters=V, threads=0)
        assert np.array_equal(e.read_commit(), after)
        # note: the first commit() seeded the commit plane, so pairs = groups whose index moved
        ch = np.flatnonzero(after != before)
        assert n == len(ch) and np.array_equal(pairs["group"], ch.astype(np.uint64))
        assert np.array_equal(pairs["commit"], after[ch])
        assert (after >= before).all()  # Match only grows, so the commit index cannot drop
        # learner rows never count: raising only learner cells changes nothing
        lr = np.zeros(1 << 16, dtype=raft_amd.ACK_DTYPE)
        lr["group"] = rng.integers(0, G, size=len(lr), dtype=np.uint32)
        lr["slot"] = rng.integers(V, S, size=len(lr), dtype=np.uint32)
        lr["index"] = np.uint64(gu.U64MAX)
        e.scatter_acks(lr)
        _, n = e.commit_changed(max_pairs=0)
        assert n == 0


def test_cfg5_at_the_config_size_16m_groups_64m_acks(oracle):
    """BASELINE cfg5 as SURVEY.md 8(d) sizes it on one GPU: 2^24 groups x (7 voters + 2 learners), M = 4 G = 2^26
    acks (about 5 % stale, duplicate cells) -> rafthip_scatter_acks -> commit pass; whole Match planes, whole commit
    plane and every changed pair against the oracle.  This is the size at which the planes (1.2 GB) leave the TLB
    reach and at which the default engine takes the partitioned scatter (K5p); the records are device-resident, as
    a host that decodes on the GPU or receives over GPU-direct holds them.  A second engine pinned to the atomic
    form must leave the same planes."""
    import torch
    from raft_amd.capi import FLAG_SCATTER_ATOMIC
    G, S, V = 1 << 24, 9, 7
    match = gu.dist_b(0x5EED0005, S, G) & np.uint64((1 << 63) - 1)
    rng = np.random.default_rng(0x5EED0055)
    M = 4 * G
    recs = np.zeros(M, dtype=raft_amd.ACK_DTYPE)
    recs["group"] = rng.integers(0, G, size=M, dtype=np.uint32)
    recs["slot"] = rng.integers(0, S, size=M, dtype=np.uint32)
    old = match[recs["slot"], recs["group"]]
    recs["index"] = old + rng.integers(0, 64, size=M, dtype=np.uint64)
    stale = rng.random(M) < 0.05
    recs["index"][stale] = old[stale] >> np.uint64(1)
    del old, stale
    before = oracle.batch_committed(match, uniform_voters=V, threads=0)
    match0 = match.copy()
    d_recs = torch.from_numpy(recs.view(np.uint8)).to("cuda:0")
    with raft_amd.Engine(G, S) as e:
        e.set_uniform_config(V)
        e.upload_match(match)
        e.upload_commit(before)
        e.scatter_acks(d_recs.data_ptr(), M)
        pairs, n = e.commit_changed()
        assert oracle.scatter_acks(match, recs) == 0  # match is now the oracle's planes after the acks
        got = e.read_match()
        assert np.array_equal(got, match)
        del got
        after = oracle.batch_committed(match, uniform_voters=V, threads=0)
        assert np.array_equal(e.read_commit(), after)
        ch = np.flatnonzero(after != before)
        assert n == len(ch) and n > G // 4
        assert np.array_equal(pairs["group"], ch.astype(np.uint64)) and np.array_equal(pairs["commit"], after[ch])
        assert (after >= before).all()
        assert e.stats()["acks_rejected"] == 0
    del pairs, ch, before
    with raft_amd.Engine(G, S, flags=FLAG_SCATTER_ATOMIC) as a:  # the other form of the same call
        a.set_uniform_config(V)
        a.upload_match(match0)
        del match0
        a.scatter_acks(d_recs.data_ptr(), M)
        a.commit()
        assert np.array_equal(a.read_commit(), after)
        assert np.array_equal(a.read_match(), match)


def test_cfg5_heartbeats_to_the_wire(oracle):
    """cfg5 shape (2^22 groups x 7 voters + 2 learners): the 3.4e7 serialized MsgHeartbeat of one bcastHeartbeat,
    whole output byte for byte against the oracle, plus what holds at any size: the Commit each message carries
    never exceeds the peer's Match nor the group's commit index, and every message parses back to its (To, From)."""
    G, S, V = 1 << 22, 9, 7
    match = gu.dist_b(0x5EED0005, S, G) & np.uint64((1 << 63) - 1)
    rng = np.random.default_rng(0x5EED0105)
    ids = np.ascontiguousarray((np.arange(S, dtype=np.uint64)[:, None] + np.uint64(1)) * np.uint64(0x10001)
                               + rng.integers(0, 1 << 16, size=G, dtype=np.uint64)[None, :] * np.uint64(1 << 20))
    term = rng.integers(1, 1 << 20, size=G).astype(np.uint64)
    ctx = rng.integers(1, 1 << 63, size=G, dtype=np.uint64)
    ctx_len = np.where(rng.random(G) < 0.5, 8, 0).astype(np.uint8)
    self_mask = (np.uint16(1) << rng.integers(0, V, size=G).astype(np.uint16)).astype(np.uint16)
    self_mask[rng.random(G) < 0.05] = 0  # not led here
    with raft_amd.Engine(G, S) as e:
        e.set_uniform_config(V)
        e.upload_match(match)
        e.commit()
        commit = e.read_commit()
        e.upload_ids(ids)
        e.upload_term(term)
        e.upload_read_ctx(ctx, ctx_len)
        e.upload_self(self_mask)
        got, got_len = e.emit_heartbeats()
    led = self_mask != 0
    assert (got_len[:, ~led] == 0).all() and (got_len[:, led] > 0).sum() == (S - 1) * int(led.sum())
    want, want_len = oracle.bcast_heartbeat(match, commit, ids, term, ctx, ctx_len, self_mask)
    assert np.array_equal(got_len, want_len)
    assert np.array_equal(got, want)
    del want
    # a sample parsed back by Unmarshal: To / From / Term as uploaded, Commit = min(Match, committed)
    for k in rng.integers(0, S * G, size=300):
        s, g = divmod(int(k), G)
        if got_len[s, g] == 0:
            continue
        rc, f = oracle.message_unmarshal(got[s, g, :got_len[s, g]].tobytes())
        me = int(self_mask[g]).bit_length() - 1
        assert rc == 0 and (f["type"], f["to"], f["from"], f["term"]) == (8, int(ids[s, g]), int(ids[me, g]), int(term[g]))
        assert f["commit"] == min(int(match[s, g]), int(commit[g]))
        assert f["context"] == (int(ctx[g]).to_bytes(8, "little") if ctx_len[g] else None)
