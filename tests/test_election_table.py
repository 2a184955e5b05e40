"""raft_paper_test.go:193-234 TestLeaderElectionInOneRoundRPC (tests/golden/tables.json, lifted mechanically): a
candidate's own vote plus the vote responses of the table -> raft.poll's verdict (raft.go:1041-1049), which decides the
role the reference's test checks (raft.go:1650-1662).  On the oracle: the literal map-shaped VoteResult and the plane
form; on the GPU: from vote records and from serialized MsgVoteResp bytes of campaigning groups."""
import numpy as np
import pytest

import golden_util as gu
from oracle import pyoracle as po

VERDICT = {"StateLeader": 3, "StateFollower": 2, "StateCandidate": 1}  # VoteWon / VoteLost / VotePending
S = 5


def _rows():
    t = gu.load_tables()["TestLeaderElectionInOneRoundRPC"]
    assert len(t["rows"]) == 13
    return t["rows"]


def _planes(rows):
    G = len(rows)
    votes = np.zeros((S, G), dtype=np.uint8)
    in_mask = np.zeros(G, dtype=np.uint8)
    for g, r in enumerate(rows):
        in_mask[g] = (1 << r["size"]) - 1  # ids 1..size in slots 0..size-1 (idsBySize, raft_test.go)
        votes[0, g] = 2                    # the candidate votes for itself (raft.go:1015-1020 campaign -> poll)
        for i, v in r["votes"]:
            votes[i - 1, g] = 2 if v else 1
    return votes, in_mask


def test_election_table_on_the_oracle():
    rows = _rows()
    votes, in_mask = _planes(rows)
    res, gr, rj = po.batch_tally_votes(votes, in_mask, np.zeros_like(in_mask))
    for g, r in enumerate(rows):
        ids = list(range(1, r["size"] + 1))
        m = {1: True}
        m.update({i: v for i, v in r["votes"]})
        assert po.majority_vote_result(ids, m) == VERDICT[r["state"]] == int(res[g]), r
        assert int(gr[g]) == sum(m.values()) and int(rj[g]) == len(m) - sum(m.values()), r


@pytest.mark.gpu
def test_election_table_on_the_engine_from_records_and_from_bytes():
    import raft_amd
    import wire_util as wu
    from raft_amd import capi
    rows = _rows()
    G = len(rows)
    _, in_mask = _planes(rows)
    want = np.array([VERDICT[r["state"]] for r in rows], dtype=np.uint8)
    ids = np.zeros((S, G), dtype=np.uint64)
    for g, r in enumerate(rows):
        ids[:r["size"], g] = np.arange(1, r["size"] + 1)
    for from_bytes in (False, True):
        with raft_amd.Engine(G, S) as e:
            e.upload_config(in_mask, np.zeros_like(in_mask))
            e.reset_votes()
            own = np.zeros(G, dtype=capi.MSG_DTYPE)  # every candidate's own vote
            own["group"], own["slot"], own["type"], own["reject"] = np.arange(G), 0, capi.MSG_VOTE_RESP, 0
            e.record_votes(own)
            pairs = [(g, i, v) for g, r in enumerate(rows) for i, v in r["votes"]]
            if not from_bytes:
                m = np.zeros(len(pairs), dtype=capi.MSG_DTYPE)
                m["group"] = [g for g, _, _ in pairs]
                m["slot"] = [i - 1 for _, i, _ in pairs]
                m["type"], m["reject"] = capi.MSG_VOTE_RESP, [0 if v else 1 for _, _, v in pairs]
                e.record_votes(m)
            else:
                e.upload_ids(ids)
                e.upload_term(np.ones(G, dtype=np.uint64))  # the test checks r.Term == 1
                e.upload_read_ctx(np.zeros(G, dtype=np.uint64), np.zeros(G, dtype=np.uint8))
                e.upload_campaign(np.full(G, capi.MSG_VOTE_RESP, dtype=np.uint8))
                blobs = [wu.encode_message(type=wu.MSG_VOTE_RESP, to=1, frm=i, term=1, reject=not v) for _, i, v in pairs]
                data, offsets = wu.pack(blobs)
                _, status, counts = e.ingest_wire(data, offsets, np.array([g for g, _, _ in pairs], dtype=np.uint32))
                assert (status == capi.WIRE_VOTE).all() and counts[capi.WIRE_VOTE] == len(pairs)
            e.tally_votes()
            res, gr, rj = e.read_vote_result()
            assert np.array_equal(res, want), (from_bytes, res, want)
            ores, ogr, orj = po.batch_tally_votes(e.read_votes(), in_mask, np.zeros_like(in_mask))
            assert np.array_equal(gr, ogr) and np.array_equal(rj, orj) and np.array_equal(res, ores)
