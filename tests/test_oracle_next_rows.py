"""CPU pins for the oracle's restatements of the steps either side of the
quorum arithmetic: raftLog.maybeCommit on an explicit log,
stepLeader's per-message bookkeeping, MsgCheckQuorum, sendHeartbeat's commit."""
import numpy as np

import golden_util as gu

TABLES = gu.load_tables()


def test_testcommit_table_with_explicit_logs(oracle):
    """TestCommit, raft_test.go:1178-1225: matches -> Committed() -> raftLog.maybeCommit
    on the row's own log (storage starts at index 1, dummy entry term 0)."""
    for row in TABLES["TestCommit"]["rows"]:
        n = len(row["matches"])
        ids = list(range(1, n + 1))
        mci = oracle.majority_committed_index(ids, dict(zip(ids, row["matches"])))
        assert [i for i, _ in row["logs"]] == list(range(1, len(row["logs"]) + 1))
        terms = [t for _, t in row["logs"]]
        adv, com = oracle.raftlog_maybe_commit(1, 0, terms, 0, mci, row["smTerm"])
        assert com == row["w"] and adv == (row["w"] != 0), row


def test_maybe_commit_guard_edges(oracle):
    terms = [1, 1, 2, 2, 3]          # indexes 5..9, entry 4 compacted with term 1
    f = lambda c, i, t: oracle.raftlog_maybe_commit(5, 1, terms, c, i, t)
    assert f(5, 7, 2) == (True, 7)
    assert f(7, 7, 2) == (False, 7)      # not beyond committed
    assert f(5, 6, 2) == (False, 5)      # entry 6 is term 1
    assert f(5, 10, 3) == (False, 5)     # beyond lastIndex: ErrUnavailable -> 0
    assert f(0, 4, 1) == (True, 4)       # the compacted boundary entry keeps its term
    assert f(0, 3, 1) == (False, 0)      # ErrCompacted -> 0
    assert f(5, 9, 0) == (False, 5)      # term 0 never matches (log.go:448-450)
    assert oracle.raftlog_maybe_commit(1, 0, [], 0, 0, 0) == (False, 0)


def test_term_range_equals_literal_term_lookup(oracle):
    """The (term_first, term_last) summary the engine consumes decides exactly as
    raftLog.term does, for every index around the log."""
    rng = np.random.default_rng(11)
    G = 400
    logs = gu.random_logs(rng, G)
    off, terms, first, snap = logs
    term = rng.integers(0, 6, size=G).astype(np.uint64)
    tf, tl = gu.term_range(off, terms, first, snap, term)
    for g in range(G):
        t = terms[int(off[g]):int(off[g + 1])]
        for i in range(0, int(first[g]) + len(t) + 3):
            for c in (0, max(0, i - 1)):
                adv, _ = oracle.raftlog_maybe_commit(int(first[g]), int(snap[g]), t, c, i, int(term[g]))
                assert adv == (i > c and int(tf[g]) <= i <= int(tl[g])), (g, i, c)


def test_step_leader_msgs_order_independent(oracle):
    rng = np.random.default_rng(5)
    S, G, N = 5, 50, 4000
    msgs = np.zeros(N, dtype=oracle.MSG_DTYPE)
    msgs["group"] = rng.integers(0, G + 2, size=N)
    msgs["slot"] = rng.integers(0, S + 1, size=N)
    msgs["type"] = rng.choice([4, 9, 0xF0, 3, 6], size=N, p=[0.5, 0.3, 0.1, 0.05, 0.05])
    msgs["reject"] = rng.random(N) < 0.1
    msgs["flags"] = rng.integers(0, 2, size=N)
    msgs["index"] = rng.integers(0, 1000, size=N)
    outs = []
    for perm in (np.arange(N), rng.permutation(N)):
        match = np.zeros((S, G), dtype=np.uint64)
        active = np.zeros(G, dtype=np.uint8)
        acks = np.zeros(G, dtype=np.uint8)
        last = np.zeros(G, dtype=np.uint64)
        cnt = oracle.step_leader_msgs(match, active, acks, last, np.ascontiguousarray(msgs[perm]))
        outs.append((match, active, acks, last, cnt))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    match, active, acks, last, (dropped, ignored) = outs[0]
    ok = (msgs["group"] < G) & (msgs["slot"] < S)
    assert ignored == int(np.isin(msgs["type"], [3, 6]).sum())
    assert dropped == int((np.isin(msgs["type"], [4, 9]) & ~ok).sum() + ((msgs["type"] == 0xF0) & (msgs["group"] >= G)).sum())
    # a rejected MsgAppResp marks the peer active but leaves Match alone (raft.go:1337-1352)
    m = np.zeros(1, dtype=oracle.MSG_DTYPE)
    m["type"], m["reject"], m["index"] = 4, 1, 77
    match = np.zeros((1, 1), dtype=np.uint64)
    active = np.zeros(1, dtype=np.uint8)
    oracle.step_leader_msgs(match, active, np.zeros(1, dtype=np.uint8), np.zeros(1, dtype=np.uint64), m)
    assert match[0, 0] == 0 and active[0] == 1


def test_check_quorum_clears_all_but_self(oracle):
    """raft.go:1231-1243 and testdata/checkquorum.txt: the verdict uses the bits as
    they were; afterwards only the leader's own RecentActive survives."""
    active = np.array([0b111, 0b001, 0b110, 0b000], dtype=np.uint8)
    self_mask = np.array([0b001, 0b001, 0b001, 0b001], dtype=np.uint8)
    out = oracle.batch_check_quorum(active, 3, self_mask, uniform_voters=3)
    assert out.tolist() == [1, 0, 1, 0]
    assert active.tolist() == [0b001, 0b001, 0b000, 0b000]


def test_heartbeat_commit_is_min(oracle):
    match = np.array([[5, 0, 9], [7, 3, 1 << 63]], dtype=np.uint64)
    com = np.array([6, 2, 1 << 62], dtype=np.uint64)
    assert np.array_equal(oracle.batch_heartbeat_commit(match, com), np.minimum(match, com[None, :]))


def _scenarios():
    import json
    import os
    with open(os.path.join(gu.HERE, "golden", "scenarios.json")) as fh:
        return json.load(fh)


NAMES = {"VotePending": 1, "VoteLost": 2, "VoteWon": 3}


def test_scenario_fixtures_on_the_oracle(oracle):
    """tests/golden/scenarios.json: what the reference's state-machine tests state about
    TallyVotes, MsgCheckQuorum and the ReadIndex ack quorum, on the id-level oracle."""
    sc = _scenarios()
    for seq in sc["vote_sequences"]:
        votes = {}
        for st in seq["steps"]:
            for i, v in st["record"]:
                votes.setdefault(i, v)  # RecordVote: first vote sticks (tracker.go:260-265)
            assert oracle.majority_vote_result(seq["voters"], votes) == NAMES[st["result"]], (seq["src"], st)
            assert sum(1 for v in votes.values() if v) == st["granted"]
            assert sum(1 for v in votes.values() if not v) == st["rejected"]
    for seq in sc["check_quorum_sequences"]:
        active = set()
        for st in seq["steps"]:
            active |= set(st.get("active_before", [])) | set(st["heard_from"]) | {seq["self"]}
            votes = {i: (i in active) for i in seq["voters"]}
            assert (oracle.majority_vote_result(seq["voters"], votes) == 3) == st["quorum_active"], (seq["src"], st)
            active = {seq["self"]}  # raft.go:1238-1242
    for seq in sc["read_index_sequences"]:
        for st in seq["steps"]:
            acks = {i: True for i in st["acks"]}
            assert oracle.majority_vote_result(seq["voters"], acks) == NAMES[st["result"]], (seq["src"], st)


def test_heartbeat_commit_cases(oracle):
    """sendHeartbeat's Commit (raft.go:669-685) against what the reference's interaction tests print."""
    import json
    import os
    with open(os.path.join(gu.HERE, "golden", "scenarios.json")) as fh:
        cases = json.load(fh)["heartbeat_commit_cases"]
    for c in cases:
        match = np.array(c["match"], dtype=np.uint64).reshape(-1, 1)
        got = oracle.batch_heartbeat_commit(match, np.array([c["committed"]], dtype=np.uint64))
        for peer, want in c["want"].items():
            assert int(got[c["ids"].index(int(peer)), 0]) == want, c["src"]


def _vote_records(pairs, ids, group=0):
    msgs = np.zeros(len(pairs), dtype=oracle_msg_dtype())
    for k, (i, v) in enumerate(pairs):
        msgs[k] = (group, ids.index(i), 6, 0 if v else 1, 0, 0)
    return msgs


def oracle_msg_dtype():
    from oracle import pyoracle
    return pyoracle.MSG_DTYPE


def test_record_votes_on_the_scenarios_and_first_vote_sticks(oracle):
    """oracle.record_votes = RecordVote one response at a time (tracker/tracker.go:260-265): the vote
    sequences of the reference's tests, then conflicting duplicates and foreign records."""
    sc = _scenarios()
    for seq in sc["vote_sequences"]:
        ids = seq["voters"]
        cells = np.zeros((len(ids), 1), dtype=np.uint8)
        for st in seq["steps"]:
            assert oracle.record_votes(cells, _vote_records(st["record"], ids)) == 0
            res, gr, rj = oracle.batch_tally_votes(cells, uniform_voters=len(ids))
            assert (int(res[0]), int(gr[0]), int(rj[0])) == (NAMES[st["result"]], st["granted"], st["rejected"]), (seq["src"], st)
    cells = np.zeros((3, 2), dtype=np.uint8)
    cells[2, 1] = 1  # recorded before the batch: sticks
    msgs = np.zeros(7, dtype=oracle_msg_dtype())
    msgs["group"] = [1, 1, 1, 1, 0, 5, 1]
    msgs["slot"] = [0, 0, 2, 1, 3, 0, 1]
    msgs["type"] = [6, 18, 6, 4, 6, 6, 18]
    msgs["reject"] = [1, 0, 0, 0, 0, 0, 0]
    assert oracle.record_votes(cells, msgs) == 3  # an MsgAppResp, a slot without a row, a group out of range
    assert cells[:, 1].tolist() == [1, 2, 1] and not cells[:, 0].any()
