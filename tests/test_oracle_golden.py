"""Pins the CPU oracle against every golden vector the reference holds for the
hot path.  CPU only."""
import numpy as np
import pytest

import golden_util as gu

CASES = gu.load_cases()
TABLES = gu.load_tables()


def test_case_counts():
    # 16 + 50 + 22 + 39 = 127 stanzas in quorum/testdata
    assert len(CASES) == 127
    assert all(c["clean"] for c in CASES)


@pytest.mark.parametrize("case", [c for c in CASES if c["cmd"] == "committed"],
                         ids=lambda c: c["src"].split("/")[-1])
def test_committed_golden(oracle, case):
    l = gu.lookuper(case)
    want = gu.expected_value(case)
    c, cj = case["cfg"], case["cfgj"]
    if not case["joint"]:
        idx = oracle.majority_committed_index(c, l)
        assert idx == want
        # the harness's metamorphic self-checks, datadriven_test.go:175-211
        assert oracle.alt_majority_committed_index(c, l) == idx
        assert oracle.joint_committed_index(c, [], l) == idx
        assert oracle.joint_committed_index(c, c, l) == idx
        for i in c:
            iidx = l.get(i, 0)
            if idx > iidx and iidx > 0:
                for low in (iidx - 1, 0):
                    lo = {k: v for k, v in l.items() if k in c}
                    lo[i] = low
                    assert oracle.majority_committed_index(c, lo) == idx
    else:
        idx = oracle.joint_committed_index(c, cj, l)
        assert idx == want
        assert oracle.joint_committed_index(cj, c, l) == idx  # symmetry, :219


@pytest.mark.parametrize("case", [c for c in CASES if c["cmd"] == "vote"],
                         ids=lambda c: c["src"].split("/")[-1])
def test_vote_golden(oracle, case):
    votes = gu.lookuper(case)
    want = gu.expected_value(case)
    c, cj = case["cfg"], case["cfgj"]
    if not case["joint"]:
        assert oracle.majority_vote_result(c, votes) == want
    else:
        r = oracle.joint_vote_result(c, cj, votes)
        assert r == want
        assert oracle.joint_vote_result(cj, c, votes) == r  # symmetry, :239


def test_quick_vs_alternative(oracle):
    """quorum/quick_test.go:28-64 — random (config, idx-map) pairs, <=9 ids,
    idx < n, main algorithm vs alternativeMajorityCommittedIndex."""
    rng = np.random.default_rng(0x5EED)

    def small_map():
        n = int(rng.integers(0, 10))
        ids = rng.permutation(2 * n)[:n] if n else []
        return {int(i): int(rng.integers(0, n)) for i in ids}

    for _ in range(20000):
        c = list(small_map().keys())
        l = small_map()
        assert oracle.majority_committed_index(c, l) == oracle.alt_majority_committed_index(c, l)


def test_ten_million_random_cases_vs_alternative(oracle):
    """The same differential at volume (SURVEY.md §8c plan: >= 10^7 cases), with
    edge generators the reference's quick test lacks: up to 64 voters, ids missing
    from the indexer, zeros, ties, values >= 2^63."""
    assert oracle.selfcheck_committed(10_000_000) == 0


def test_batch_views_match_golden(oracle):
    """The SoA views the GPU tests compare against reproduce the goldens too."""
    for swap in (False, True):
        plane, m0, m1, exp = gu.cases_to_planes(CASES, "committed", swap=swap)
        got = oracle.batch_committed(plane, m0, m1)
        assert np.array_equal(got, exp)
        votes, v0, v1, vexp = gu.cases_to_planes(CASES, "vote", swap=swap)
        res, gr, rj = oracle.batch_tally_votes(votes, v0, v1)
        assert np.array_equal(res, vexp)
    # wider slot counts / mask widths must not change anything
    for S in (9, 16, 33, 64):
        plane, m0, m1, exp = gu.cases_to_planes(CASES, "committed", nslots=S)
        assert np.array_equal(oracle.batch_committed(plane, m0, m1), exp)


def test_testcommit_table(oracle):
    """raft_test.go:1186-1203: Committed() then raftLog.maybeCommit's term guard."""
    for row in TABLES["TestCommit"]["rows"]:
        n = len(row["matches"])
        plane = np.array(row["matches"], dtype=np.uint64).reshape(n, 1)
        mci = int(oracle.batch_committed(plane, uniform_voters=n)[0])
        terms = {i: t for i, t in row["logs"]}
        got = oracle.log_maybe_commit(0, mci, terms.get(mci, 0), row["smTerm"])
        assert got == row["w"], row


def test_leader_acknowledge_commit_table(oracle):
    """raft_paper_test.go:482-490: leader (id 1) has the entry at li+1; acceptors ack it."""
    li = 1
    for row in TABLES["TestLeaderAcknowledgeCommit"]["rows"]:
        n = row["size"]
        match = np.full((n, 1), li, dtype=np.uint64)
        match[0, 0] = li + 1
        for a in row["acceptors"]:
            match[a - 1, 0] = li + 1
        mci = int(oracle.batch_committed(match, uniform_voters=n)[0])
        assert (mci > li) == row["wack"], row


def test_progress_update_table(oracle):
    t = TABLES["TestProgressUpdate"]
    for row in t["rows"]:
        ok, m, n = oracle.maybe_update(t["prevM"], t["prevN"], row["update"])
        assert (ok, m, n) == (row["wok"], row["wm"], row["wn"])


def test_quorum_active_and_acks(oracle):
    """tracker.go:217-227 / read_only.go:68-76 over a 3-voter group
    (TestLeaderStepdownWhenQuorumActive/Lost, raft_test.go:1911-1944)."""
    in_mask = np.array([0b111] * 4, dtype=np.uint8)
    out_mask = np.zeros(4, dtype=np.uint8)
    active = np.array([0b001, 0b011, 0b101, 0b111], dtype=np.uint8)
    got = oracle.batch_quorum_active(active, 3, in_mask, out_mask)
    assert got.tolist() == [0, 1, 1, 1]
    got = oracle.batch_ack_quorum(active, 3, in_mask, out_mask)
    assert got.tolist() == [1, 3, 3, 3]
    # bootstrap: both halves empty -> Won / active (tracker.go:133-139, majority.go:179-184)
    z = np.zeros(1, dtype=np.uint8)
    assert oracle.batch_quorum_active(z, 3, z, z).tolist() == [1]
    assert oracle.batch_ack_quorum(z, 3, z, z).tolist() == [3]
    assert oracle.batch_committed(np.zeros((3, 1), dtype=np.uint64), z, z).tolist() == [gu.U64MAX]


def test_scatter_is_monotone_max(oracle):
    rng = np.random.default_rng(1)
    S, G, N = 4, 50, 2000
    match = rng.integers(0, 100, size=(S, G)).astype(np.uint64)
    recs = np.zeros(N, dtype=oracle.ACK_DTYPE)
    recs["group"] = rng.integers(0, G + 2, size=N)
    recs["slot"] = rng.integers(0, S + 1, size=N)
    recs["index"] = rng.integers(0, 200, size=N)
    want = match.copy()
    skipped = 0
    for r in recs:
        if r["group"] >= G or r["slot"] >= S:
            skipped += 1
            continue
        want[r["slot"], r["group"]] = max(want[r["slot"], r["group"]], r["index"])
    assert oracle.scatter_acks(match, recs) == skipped
    assert np.array_equal(match, want)
