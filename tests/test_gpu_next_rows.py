"""GPU parity for the steps either side of the quorum arithmetic:
raft.maybeCommit with the term guard fused into the commit pass, stepLeader's
per-message bookkeeping (ingest), the whole MsgCheckQuorum step, and the
per-peer heartbeat commit.  All through the C ABI, bit-exact vs the oracle."""
import numpy as np
import pytest

import golden_util as gu
import raft_amd
from test_gpu_parity import rand_masks

pytestmark = pytest.mark.gpu

TABLES = gu.load_tables()


def test_commit_advance_testcommit_table():
    """TestCommit (raft_test.go:1178-1225) end to end on the GPU: Match -> quorum
    index -> term guard -> committed."""
    for row in TABLES["TestCommit"]["rows"]:
        n = len(row["matches"])
        terms = np.array([t for _, t in row["logs"]], dtype=np.uint64)
        off = np.array([0, len(terms)], dtype=np.uint64)
        one = np.ones(1, dtype=np.uint64)
        tf, tl = gu.term_range(off, terms, one, np.zeros(1, dtype=np.uint64),
                               np.array([row["smTerm"]], dtype=np.uint64))
        with raft_amd.Engine(1, n) as e:
            e.set_uniform_config(n)
            e.upload_match(np.array(row["matches"], dtype=np.uint64).reshape(n, 1))
            e.upload_term_range(tf, tl)
            pairs, k = e.commit_advance()
            assert int(e.read_commit()[0]) == row["w"], row
            assert k == (1 if row["w"] else 0)
            if k:
                assert pairs["group"][0] == 0 and pairs["commit"][0] == row["w"]


@pytest.mark.parametrize("S,mode", [(5, "uniform"), (3, "uniform"), (6, "masked"), (8, "masked"), (12, "wide"), (40, "wide"),
                                    (12, "bitonic"), (40, "bitonic")])
def test_commit_advance_random_logs(oracle, S, mode):
    rng = np.random.default_rng(700 + S)
    G = 30011 if S <= 8 else 3001
    logs = gu.random_logs(rng, G)
    off, terms, first, snap = logs
    lens = (off[1:] - off[:-1]).astype(np.int64)
    last = first.astype(np.int64) + lens - 1
    # r.Term: mostly a term present in the log, sometimes beyond it, sometimes 0
    term = np.where(lens > 0, terms[np.minimum(off[:-1].astype(np.int64) + np.maximum(lens - 1, 0), len(terms) - 1)], snap)
    term = term - (rng.random(G) < 0.2).astype(np.uint64) * (term > 0)
    term = np.where(rng.random(G) < 0.05, term + np.uint64(1), term)
    term = np.where(rng.random(G) < 0.03, np.uint64(0), term).astype(np.uint64)
    committed = np.maximum(first.astype(np.int64) - 1 + rng.integers(0, 4, size=G), 0).astype(np.uint64)
    # Match around the log: below first, inside, a little beyond lastIndex
    match = (np.maximum(first.astype(np.int64)[None, :] - 2 + rng.integers(0, 16, size=(S, G)), 0)).astype(np.uint64)
    match = np.minimum(match, (last[None, :] + 2).astype(np.uint64))
    match = np.ascontiguousarray(match)
    tf, tl = gu.term_range(off, terms, first, snap, term)
    with raft_amd.Engine(G, S, flags=raft_amd.capi.FLAG_WIDE_BITONIC if mode == "bitonic" else 0) as e:
        if mode == "uniform":
            e.set_uniform_config(S)
            kw = dict(uniform_voters=S)
        else:
            m0, m1 = rand_masks(rng, S, G)
            e.upload_config(m0, m1)
            kw = dict(in_mask=m0, out_mask=m1)
        e.upload_match(match)
        e.upload_commit(committed)
        e.upload_term_range(tf, tl)
        pairs, n = e.commit_advance()
        want, adv = oracle.batch_maybe_commit(match, logs, term, committed, **kw)
        assert np.array_equal(e.read_commit(), want)
        idx = np.flatnonzero(adv)
        assert n == len(idx) and np.array_equal(pairs["group"], idx.astype(np.uint64))
        assert np.array_equal(pairs["commit"], want[idx])
        assert 0 < len(idx) < G
        # a second pass over the same state advances nothing (commitTo never decreases)
        _, n = e.commit_advance(max_pairs=0)
        assert n == 0
        assert e.stats()["advanced_reported"] == len(idx)


def test_committed_entry_in_current_term(oracle):
    """raft.go:2001-2005 against the literal term lookup on explicit logs."""
    rng = np.random.default_rng(77)
    G = 5000
    logs = gu.random_logs(rng, G)
    off, terms, first, snap = logs
    term = rng.integers(0, 6, size=G).astype(np.uint64)
    committed = np.maximum(first.astype(np.int64) - 2 + rng.integers(0, 16, size=G), 0).astype(np.uint64)
    tf, tl = gu.term_range(off, terms, first, snap, term)
    want = np.array([oracle.committed_entry_in_current_term(int(first[g]), int(snap[g]),
                                                             terms[int(off[g]):int(off[g + 1])], int(committed[g]),
                                                             int(term[g])) and int(term[g]) != 0
                     for g in range(G)], dtype=np.uint8)
    with raft_amd.Engine(G, 3) as e:
        e.upload_commit(committed)
        e.upload_term_range(tf, tl)
        got = e.committed_in_term()
        assert np.array_equal(got, want) and 0 < want.sum() < G
        assert np.array_equal(e.committed_in_term(100, 50), want[100:150])


def rand_msgs(rng, N, G, S, match=None):
    msgs = np.zeros(N, dtype=raft_amd.MSG_DTYPE)
    msgs["group"] = rng.integers(0, G, size=N)
    msgs["slot"] = rng.integers(0, S, size=N)
    msgs["type"] = rng.choice(np.array([4, 9, 0xF0], dtype=np.uint8), size=N, p=[0.6, 0.3, 0.1])
    msgs["reject"] = rng.random(N) < 0.1
    msgs["flags"] = rng.random(N) < 0.5
    base = match[msgs["slot"], msgs["group"]] if match is not None else np.uint64(0)
    msgs["index"] = base + rng.integers(0, 64, size=N).astype(np.uint64)
    return msgs


@pytest.mark.parametrize("S", [3, 7, 9, 16, 33, 64])
def test_ingest_msgs(oracle, S):
    rng = np.random.default_rng(800 + S)
    G = 5003
    N = 20 * G
    mdt = gu.mask_dtype(S)
    match = rng.integers(0, 1 << 40, size=(S, G), dtype=np.uint64)
    msgs = rand_msgs(rng, N, G, S, match)
    # malformed and foreign records: dropped / ignored, never faulted on
    msgs["group"][:4] = [G, G + 1, 0xFFFFFFFF, G]
    msgs["type"][:4] = [4, 9, 4, 0xF0]
    msgs["slot"][4:6] = [S if S < 255 else 255, 255]
    msgs["type"][4:6] = [4, 9]
    msgs["type"][6:9] = [3, 6, 18]
    a0, _ = rand_masks(rng, S, G, joint_frac=0, empty_frac=0.5)
    k0 = np.zeros(G, dtype=mdt)
    tl0 = rng.integers(0, 1 << 40, size=G, dtype=np.uint64)
    want_match, want_a, want_k, want_tl = match.copy(), a0.copy(), k0.copy(), tl0.copy()
    dropped, ignored = oracle.step_leader_msgs(want_match, want_a, want_k, want_tl, msgs)
    with raft_amd.Engine(G, S) as e:
        e.upload_match(match)
        e.upload_active(a0)
        e.upload_acks(k0)
        e.upload_term_range(np.zeros(G, dtype=np.uint64), tl0)
        e.ingest_msgs(msgs)
        assert np.array_equal(e.read_match(), want_match)
        assert np.array_equal(e.read_active(), want_a)
        assert np.array_equal(e.read_acks(), want_k)
        assert np.array_equal(e.read_term_range()[1], want_tl)
        st = e.stats()
        assert (st["msgs_dropped"], st["msgs_ignored"], st["msgs_ingested"]) == (dropped, ignored, N)
        assert dropped == 6 and ignored == 3


@pytest.mark.parametrize("S", [3, 5, 8, 12, 64])
def test_check_quorum(oracle, S):
    rng = np.random.default_rng(900 + S)
    G = 40009 if S <= 8 else 4001
    mdt = gu.mask_dtype(S)
    m0, m1 = rand_masks(rng, S, G)
    learners = (~(m0 | m1)) & mdt((1 << S) - 1 if S < 64 else gu.U64MAX)
    active, _ = rand_masks(rng, S, G, joint_frac=0, empty_frac=0.1)
    self_mask = (np.uint64(1) << rng.integers(0, S, size=G).astype(np.uint64)).astype(mdt)
    self_mask[rng.random(G) < 0.05] = 0  # a leader that is no longer in the config
    with raft_amd.Engine(G, S) as e:
        e.upload_config(m0, m1, learners)
        e.upload_active(active)
        e.upload_self(self_mask)
        e.check_quorum()
        want_a = active.copy()
        want = oracle.batch_check_quorum(want_a, S, self_mask, m0, m1, learners)
        assert np.array_equal(e.read_quorum_active(), want)
        assert np.array_equal(e.read_active(), want_a)
        assert np.array_equal(want_a, active & self_mask)
        # uniform config path
        V = max(1, S - 1)
        e.set_uniform_config(V)
        e.upload_active(active)
        e.check_quorum()
        want_a = active.copy()
        want = oracle.batch_check_quorum(want_a, S, self_mask, uniform_voters=V)
        assert np.array_equal(e.read_quorum_active(), want)
        assert np.array_equal(e.read_active(), want_a)


@pytest.mark.parametrize("S,G", [(1, 1), (5, 4097), (9, 10000), (64, 513)])
def test_heartbeat_commit(oracle, S, G):
    rng = np.random.default_rng(1000 + S)
    match = rng.integers(0, 1 << 63, size=(S, G), dtype=np.uint64) * np.uint64(2)
    com = rng.integers(0, 1 << 63, size=G, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    with raft_amd.Engine(G, S) as e:
        e.upload_match(match)
        e.upload_commit(com)
        got = e.heartbeat_commit()
        assert np.array_equal(got, oracle.batch_heartbeat_commit(match, com))
        lo = G // 3
        part = e.heartbeat_commit(first=lo, count=G - lo)
        assert np.array_equal(part, got[:, lo:])


def test_leader_tick_pipeline_full_size(oracle):
    """The whole leader-side tick at cfg5 scale (2^22 groups, 7 voters + 2 learners):
    ingest 16M responses -> commit with term guard -> heartbeat commits -> CheckQuorum,
    every plane compared with stepping the same messages through the oracle."""
    G, S, V = 1 << 22, 9, 7
    rng = np.random.default_rng(0x5EED0006)
    match = (gu.dist_b(0x5EED0006, S, G) & np.uint64((1 << 40) - 1)) + np.uint64(64)
    lastidx = match.max(axis=0) + np.uint64(32)
    term_first = lastidx - np.minimum(rng.integers(0, 256, size=G).astype(np.uint64), lastidx)
    committed = np.sort(match[:V], axis=0)[V - (V // 2 + 1)]  # where a previous pass left it
    committed = committed - rng.integers(0, 3, size=G).astype(np.uint64)
    N = 4 * G
    msgs = rand_msgs(rng, N, G, S, match)
    app = msgs["type"] == 0xF0
    msgs["index"][app] = lastidx[msgs["group"][app]] + rng.integers(0, 8, size=int(app.sum())).astype(np.uint64)
    self_mask = np.ones(G, dtype=np.uint16)  # the leader sits in slot 0
    zero = np.zeros(G, dtype=np.uint16)
    with raft_amd.Engine(G, S) as e:
        e.set_uniform_config(V)
        e.upload_match(match)
        e.upload_commit(committed)
        e.upload_term_range(term_first, lastidx)
        e.upload_active(self_mask)
        e.upload_acks(zero)
        e.upload_self(self_mask)
        e.ingest_msgs(msgs)
        pairs, n = e.commit_advance()
        hb = e.heartbeat_commit()
        e.check_quorum()
        e.ack_quorum()
        # oracle: step the same messages one by one, then the per-group steps
        wm, wa, wk, wl = match.copy(), self_mask.copy(), zero.copy(), lastidx.copy()
        assert oracle.step_leader_msgs(wm, wa, wk, wl, msgs) == (0, 0)
        mci = oracle.batch_committed(wm, uniform_voters=V, threads=0)
        adv = (mci > committed) & (mci >= term_first) & (mci <= wl)  # cross-checked vs explicit logs in the small tests
        want_c = np.where(adv, mci, committed)
        assert np.array_equal(e.read_match(), wm)
        assert np.array_equal(e.read_commit(), want_c)
        idx = np.flatnonzero(adv)
        assert n == len(idx) and np.array_equal(pairs["group"], idx.astype(np.uint64))
        assert np.array_equal(pairs["commit"], mci[idx])
        assert np.array_equal(hb, oracle.batch_heartbeat_commit(wm, want_c))
        assert np.array_equal(e.read_ack_quorum(), oracle.batch_ack_quorum(wk, S, uniform_voters=V))
        want_qa = oracle.batch_check_quorum(wa, S, self_mask, uniform_voters=V)
        assert np.array_equal(e.read_quorum_active(), want_qa)
        assert np.array_equal(e.read_active(), wa)
        assert 0.05 < want_qa.mean() < 1.0 and len(idx) > G // 10


def test_scenario_fixtures_on_the_gpu():
    """tests/golden/scenarios.json through the engine: one group per sequence, slot =
    position of the id among voters + learners."""
    import json
    import os
    with open(os.path.join(gu.HERE, "golden", "scenarios.json")) as fh:
        sc = json.load(fh)
    names = {"VotePending": 1, "VoteLost": 2, "VoteWon": 3}
    for seq in sc["vote_sequences"]:
        ids = seq["voters"]
        with raft_amd.Engine(1, len(ids)) as e:
            e.set_uniform_config(len(ids))
            cells = np.zeros((len(ids), 1), dtype=np.uint8)
            e.upload_votes(cells)
            for st in seq["steps"]:
                recs = np.zeros(len(st["record"]), dtype=raft_amd.MSG_DTYPE)  # RecordVote on the device
                for k, (i, v) in enumerate(st["record"]):
                    recs[k] = (0, ids.index(i), raft_amd.capi.MSG_VOTE_RESP, 0 if v else 1, 0, 0)
                e.record_votes(recs)
                e.tally_votes()
                res, gr, rj = e.read_vote_result()
                assert (int(res[0]), int(gr[0]), int(rj[0])) == (names[st["result"]], st["granted"], st["rejected"]), (seq["src"], st)
    for seq in sc["check_quorum_sequences"]:
        ids = seq["voters"]
        bit = lambda xs: np.array([sum(1 << ids.index(i) for i in xs)], dtype=np.uint8)
        with raft_amd.Engine(1, len(ids)) as e:
            e.set_uniform_config(len(ids))
            e.upload_self(bit([seq["self"]]))
            e.upload_active(bit([seq["self"]]))
            for st in seq["steps"]:
                if "active_before" in st:
                    e.upload_active(bit(st["active_before"]))
                msgs = np.zeros(len(st["heard_from"]), dtype=raft_amd.MSG_DTYPE)
                msgs["slot"] = [ids.index(i) for i in st["heard_from"]]
                msgs["type"] = raft_amd.MSG_HEARTBEAT_RESP
                e.ingest_msgs(msgs)
                e.check_quorum()
                assert bool(e.read_quorum_active()[0]) == st["quorum_active"], (seq["src"], st)
                assert int(e.read_active()[0]) == int(bit([seq["self"]])[0])
    for seq in sc["read_index_sequences"]:
        ids = seq["voters"] + seq.get("learners", [])
        with raft_amd.Engine(1, len(ids)) as e:
            e.set_uniform_config(len(seq["voters"]))
            e.upload_acks(np.zeros(1, dtype=np.uint8))
            for st in seq["steps"]:
                msgs = np.zeros(len(st["acks"]), dtype=raft_amd.MSG_DTYPE)
                msgs["slot"] = [ids.index(i) for i in st["acks"]]
                msgs["type"] = raft_amd.MSG_HEARTBEAT_RESP
                msgs["flags"] = raft_amd.MSGF_READ_CTX
                e.ingest_msgs(msgs)
                e.ack_quorum()
                assert int(e.read_ack_quorum()[0]) == names[st["result"]], (seq["src"], st)


def test_heartbeat_commit_cases_on_the_gpu():
    import json
    import os
    with open(os.path.join(gu.HERE, "golden", "scenarios.json")) as fh:
        cases = json.load(fh)["heartbeat_commit_cases"]
    for c in cases:
        n = len(c["ids"])
        with raft_amd.Engine(1, n) as e:
            e.upload_match(np.array(c["match"], dtype=np.uint64).reshape(n, 1))
            e.upload_commit(np.array([c["committed"]], dtype=np.uint64))
            got = e.heartbeat_commit()
            for peer, want in c["want"].items():
                assert int(got[c["ids"].index(int(peer)), 0]) == want, c["src"]


@pytest.mark.parametrize("S", [3, 8, 9, 33, 64])
def test_record_votes_first_vote_sticks(oracle, S):
    """rafthip_record_votes vs RecordVote stepped one response at a time: conflicting duplicates inside
    a batch, votes recorded before the batch, foreign records; then the tally."""
    rng = np.random.default_rng(1900 + S)
    G = 3001
    votes = (rng.integers(0, 3, size=(S, G)) * (rng.random((S, G)) < 0.2)).astype(np.uint8)
    m0, m1 = rand_masks(rng, S, G)
    with raft_amd.Engine(G, S) as e:
        e.upload_config(m0, m1)
        e.upload_votes(votes)
        want = votes.copy()
        for rnd in range(3):
            N = 40000
            msgs = np.zeros(N, dtype=raft_amd.MSG_DTYPE)
            msgs["group"] = rng.integers(0, G // (1 + 20 * rnd) + 3, size=N)  # later rounds pile onto few groups
            msgs["slot"] = rng.integers(0, min(S + 1, 255), size=N)
            msgs["type"] = rng.choice(np.array([6, 18, 4, 9], dtype=np.uint8), size=N, p=[0.5, 0.3, 0.1, 0.1])
            msgs["reject"] = rng.integers(0, 2, size=N)
            e.record_votes(msgs)
            dropped = oracle.record_votes(want, msgs)
            assert dropped > 0
        e.tally_votes()
        res, gr, rj = e.read_vote_result()
        ores, ogr, orj = oracle.batch_tally_votes(want, in_mask=m0, out_mask=m1)
        assert np.array_equal(res, ores) and np.array_equal(gr, ogr) and np.array_equal(rj, orj)
        got = e.read_votes()
        assert np.array_equal(got, want)
        e.reset_votes()
        e.record_votes(msgs[:10])  # after a reset every cell is open again
        w2 = np.zeros((S, G), dtype=np.uint8)
        oracle.record_votes(w2, msgs[:10])
        e.tally_votes()
        assert np.array_equal(e.read_vote_result()[0], oracle.batch_tally_votes(w2, in_mask=m0, out_mask=m1)[0])
