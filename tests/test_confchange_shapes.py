"""SURVEY.md §8 row a4 names "JointConfig.CommittedIndex (+ confchange shapes)": every tracker.Config the reference's
own conf-change transcripts reach (confchange/testdata/*.txt, lifted mechanically into
tests/golden/confchange_shapes.json) through the oracle three ways — the literal map-shaped restatement, the
independent O(n^2) algorithm of quorum/quick_test.go:85-122, and the slot-plane form the GPU is checked against —
and, on the GPU, through the engine."""
import numpy as np
import pytest

import confchange_util as cu
from oracle import pyoracle as po

S = 5  # the largest shape has five ids (voters=(1) learners=(2 3 4 5))


def test_shapes_are_what_the_transcripts_print():
    shapes = cu.load_shapes()
    assert len(shapes) == 17
    assert all(s["src"].startswith("confchange/testdata/") for s in shapes)
    assert max(len(cu.slots_of(s)) for s in shapes) == S
    # learners never vote; learners_next are voters of the outgoing half (tracker/tracker.go:60-76)
    for s in shapes:
        assert not (set(s["learners"]) & (set(s["incoming"]) | set(s["outgoing"]))), s
        assert set(s["learners_next"]) <= set(s["outgoing"]), s


def test_oracle_three_ways_on_the_reference_shapes():
    shapes = cu.load_shapes()
    rng = np.random.default_rng(1707)
    G = len(shapes) * 40
    which, m0, m1, ml, match, votes, active = cu.planes(shapes, G, S, rng)
    got = po.batch_committed(match, m0, m1)
    vres, _, _ = po.batch_tally_votes(votes, m0, m1, ml)
    for g in range(G):
        sh = shapes[which[g]]
        sl = cu.slots_of(sh)
        l = {i: int(match[k, g]) for i, k in sl.items() if match[k, g] != 0}  # 0 = not in the indexer (majority.go:149-161)
        lit = po.joint_committed_index(sh["incoming"], sh["outgoing"], l)
        alt = min(po.alt_majority_committed_index(sh["incoming"], l), po.alt_majority_committed_index(sh["outgoing"], l))
        assert lit == alt == int(got[g]), (sh, l)
        v = {i: votes[k, g] == 2 for i, k in sl.items() if votes[k, g] != 0}
        assert po.joint_vote_result(sh["incoming"], sh["outgoing"], v) == int(vres[g]), (sh, v)


@pytest.mark.gpu
def test_engine_on_the_reference_shapes():
    import raft_amd
    shapes = cu.load_shapes()
    rng = np.random.default_rng(1708)
    G = len(shapes) * 700 + 3
    which, m0, m1, ml, match, votes, active = cu.planes(shapes, G, S, rng)
    with raft_amd.Engine(G, S) as e:
        e.upload_config(m0, m1, ml)
        e.upload_match(match)
        e.commit()
        assert np.array_equal(e.read_commit(), po.batch_committed(match, m0, m1))
        e.upload_votes(votes)
        e.tally_votes()
        res, gr, rj = e.read_vote_result()
        ores, ogr, orj = po.batch_tally_votes(votes, m0, m1, ml)
        assert np.array_equal(res, ores) and np.array_equal(gr, ogr) and np.array_equal(rj, orj)
        e.upload_active(active)
        e.quorum_active()
        assert np.array_equal(e.read_quorum_active(), po.batch_quorum_active(active, S, m0, m1, ml))
    # the same shapes through the stateless call quorum.BatchCommittedIndex binds to
    assert np.array_equal(raft_amd.batch_committed_index(match, m0, m1), po.batch_committed(match, m0, m1))
