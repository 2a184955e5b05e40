"""Replays tests/golden/figure7.json — the leader's side of the reference's Raft-paper
Figure 7 transcript (testdata/probe_and_replicate.txt) — through a driver: the election's vote
tally, becomeLeader's Progress reset, every MsgApp sent (its prev index must be Next-1 of a
Progress that is not paused, its Commit the leader's commit index), every MsgAppResp stepped
(rejections after the leader's findConflictByTerm), and raft.maybeCommit after each response
(the commit index moves exactly where the transcript shows it move).

A driver holds one 7-voter group (or many copies of it) and offers
    record_vote(id, granted) -> (result, granted, rejected)
    become_leader(self_id, last_index, first_index, committed)   # raft.go:933-972
    step(kind, id, **fields) -> action byte                      # kind: appended | sent_app | app_resp
    progress(id) -> {"state", "match", "next", "is_paused"}
    plan(id) -> SEND_* of maybeSendAppend(id, true) right now
    maybe_commit() -> commit index after raft.maybeCommit
so that the oracle (tests/test_figure7_cpu.py) and the engine (tests/test_gpu_figure7.py) replay the
same steps."""
import json
import os

import progress_util as pu

HERE = os.path.dirname(os.path.abspath(__file__))
po = pu.po
VOTE = {"VotePending": 1, "VoteLost": 2, "VoteWon": 3}


def load():
    with open(os.path.join(HERE, "golden", "figure7.json")) as fh:
        return json.load(fh)


class HostLog:
    """The leader's log as the host keeps it (index -> term), for the two lookups the path leaves to the
    host: term(i) and findConflictByTerm (log.go:178-190, :381-407)."""

    def __init__(self, entries, snap_term=0):
        self.first = entries[0][0]
        self.terms = [t for _, t in entries]
        self.snap_term = snap_term

    @property
    def last(self):
        return self.first + len(self.terms) - 1

    def append(self, index, term):
        assert index == self.last + 1
        self.terms.append(term)

    def term(self, i):
        """-> term, or None where raftLog.term returns an error."""
        if i + 1 < self.first or i > self.last:
            return None
        return self.snap_term if i + 1 == self.first else self.terms[i - self.first]

    def find_conflict_by_term(self, index, term):
        while index > 0:
            ours = self.term(index)
            if ours is None:
                return index, 0
            if ours <= term:
                return index, ours
            index -= 1
        return 0, 0

    def term_range(self, term):
        """[first, last] index of the entries carrying `term` (empty: first > last)."""
        idx = [self.first + k for k, t in enumerate(self.terms) if t == term]
        return (idx[0], idx[-1]) if idx else (self.last + 1, self.last)


def replay(fx, drv):
    ids, me = fx["voters"], fx["leader"]
    for v in fx["votes"]:
        got = drv.record_vote(v["from"], v["granted"])
        assert got == (VOTE[v["result"]], v["granted_total"], v["rejected_total"]), (v, got)
    log = HostLog(fx["log"])
    drv.become_leader(me, log.last, log.first, fx["committed"], fx["term"])
    committed = fx["committed"]
    evs = fx["events"]
    for k, ev in enumerate(evs):
        where = "%s:%d" % (fx["src"], ev["line"])
        if ev["ev"] == "append":
            log.append(ev["index"], ev["term"])
            assert ev["term"] == fx["term"]
            assert drv.step("appended", me, index=ev["index"]) == 0
            # the leader acknowledges its own append (raft.go:808-818 -> the MsgAppResp branch)
            act = drv.step("app_resp", me, index=ev["index"])
            assert act & po.ACT_UPDATED, where
            assert drv.maybe_commit() == committed, where
        elif ev["ev"] == "send":
            st = drv.progress(ev["to"])
            assert st["next"] == ev["prev"] + 1, (where, st)
            assert not st["is_paused"], (where, st)
            assert log.term(ev["prev"]) == ev["prev_term"], where
            assert committed == ev["commit"], where
            assert ev["prev"] + ev["entries"] <= log.last, where
            assert drv.plan(ev["to"]) == (po.SEND_APPEND if ev["entries"] else po.SEND_EMPTY), where
            drv.step("sent_app", ev["to"], entries=ev["entries"], index=ev["prev"] + ev["entries"], nbytes=ev["entries"])
        elif ev["ev"] == "recv":
            before = drv.progress(ev["from"])
            if ev["reject"]:
                hint, hint_term = log.find_conflict_by_term(ev["hint"], ev["log_term"])
                act = drv.step("app_resp", ev["from"], index=ev["index"], reject=True, hint=ev["hint"],
                               log_term=ev["log_term"], resolved=hint)
                assert act & po.ACT_SEND_APPEND, where  # raft.go:1497-1503
                after = drv.progress(ev["from"])
                assert after["state"] == po.STATE_PROBE and not after["is_paused"], (where, after)
            else:
                act = drv.step("app_resp", ev["from"], index=ev["index"])
                assert bool(act & po.ACT_UPDATED) == (ev["index"] > before["match"]
                                                      or (before["state"] == po.STATE_PROBE and ev["index"] == before["match"])), where
                after = drv.progress(ev["from"])
                assert after["state"] == po.STATE_REPLICATE and after["match"] == max(before["match"], ev["index"]), (where, after)
            c = drv.maybe_commit()
            nxt = evs[k + 1] if k + 1 < len(evs) else None
            # what the leader did next in the transcript is what the action byte asks for: an append to the
            # sender after a rejection or an un-pausing ack (:1505-1509); nothing after an ack that changed nothing
            wants_send = bool(act & po.ACT_SEND_APPEND) or (act & (po.ACT_UPDATED | po.ACT_WAS_PAUSED)) == (po.ACT_UPDATED | po.ACT_WAS_PAUSED)
            if nxt is None or nxt["ev"] == "recv":
                assert not wants_send, (where, act)
            elif nxt["ev"] == "send":
                assert wants_send and nxt["to"] == ev["from"], (where, act)
            if nxt is not None and nxt["ev"] == "commit":
                assert c == nxt["value"] and c > committed, (where, c)
                committed = c
            else:
                assert c == committed, (where, c, committed)
        elif ev["ev"] == "commit":
            assert committed == ev["value"], where
            # bcastAppend (raft.go:1503-1504): the sends that follow are exactly the peers the plan names
            sent, j = [], k + 1
            while j < len(evs) and evs[j]["ev"] == "send":
                sent.append(evs[j]["to"])
                j += 1
            planned = [i for i in ids if i != me and drv.plan(i) != po.SEND_NONE]
            assert planned == sent, (where, planned, sent)
    for i in ids:
        st = drv.progress(i)
        assert (st["state"], st["match"], st["next"]) == (po.STATE_REPLICATE, log.last, log.last + 1), (i, st)
    return committed
