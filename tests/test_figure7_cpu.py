"""The reference's Raft-paper Figure 7 transcript (testdata/probe_and_replicate.txt, as
tests/golden/figure7.json) replayed on the oracle: votes, Progress, findConflictByTerm and
maybeCommit together, the way the leader of that transcript ran them."""
import numpy as np

import figure7_util as f7
import golden_util as gu
import progress_util as pu

po = pu.po


class OracleDriver:
    def __init__(self, oracle, fx, cap=256):
        from oracle import pyoracle
        self.o, self.py = oracle, pyoracle
        self.ids = fx["voters"]
        self.S, self.cap = len(self.ids), cap
        self.slot = {i: k for k, i in enumerate(self.ids)}
        self.votes = np.zeros((self.S, 1), dtype=np.uint8)
        self.log = f7.HostLog(fx["log"])

    def record_vote(self, frm, granted):
        m = np.zeros(1, dtype=self.py.MSG_DTYPE)
        m["slot"], m["type"], m["reject"] = self.slot[frm], 6, 0 if granted else 1
        assert self.o.record_votes(self.votes, m) == 0
        res, gr, rj = self.o.batch_tally_votes(self.votes, uniform_voters=self.S)
        return int(res[0]), int(gr[0]), int(rj[0])

    def become_leader(self, me, last, first, committed, term):
        S = self.S
        self.pl = self.py.progress_planes(S, 1, self.cap)
        self.pl["next"][:] = last + 1
        s = self.slot[me]
        self.pl["match"][s, 0] = last
        self.pl["pstate"][s, 0] = po.STATE_REPLICATE
        self.active, self.acks = pu.mask_planes(S, 1)
        self.last = np.array([last], dtype=np.uint64)
        self.first = np.array([first], dtype=np.uint64)
        self.committed, self.term, self.me = committed, term, me

    def step(self, kind, who, index=0, reject=False, hint=0, log_term=0, resolved=0, entries=0, nbytes=0):
        m = np.zeros(1, dtype=self.py.MSG_DTYPE)
        m["slot"], m["index"] = self.slot[who], index
        aux = 0
        if kind == "appended":
            m["type"] = po.EV_APPENDED
            self.log.append(index, self.term)
        elif kind == "sent_app":
            m["type"], m["flags"], aux = po.EV_SENT_APP, 1 if entries else 0, nbytes
        else:
            m["type"], m["reject"] = po.EV_APP_RESP, reject
            if reject:
                if log_term > 0:  # raft.go:1362-1459: only then is the hint refined against the leader's log
                    got = self.o.find_conflict_by_term(self.log.first, 0, self.log.terms, hint, log_term)
                    assert got == self.log.find_conflict_by_term(hint, log_term) and got[0] == resolved
                else:
                    assert resolved == hint
                aux = resolved
        act = self.py.step_progress(self.pl, self.cap, 0, self.active, self.acks, self.last, self.first, m,
                                    np.array([aux], dtype=np.uint64))
        return int(act[0])

    def set_progress(self, who, state, nxt):
        """sm.trk.Progress[who].State / .Next = ... as the reference's tests do by hand."""
        self.pl["pstate"][self.slot[who], 0] = state
        self.pl["next"][self.slot[who], 0] = nxt

    def progress(self, who):
        return pu.cell_state(self.pl, self.cap, 0, self.slot[who], 0)

    def plan(self, who):
        self_mask = np.array([1 << self.slot[self.me]], dtype=np.uint8)
        return int(self.py.plan_appends(self.pl, self.cap, 0, None, self_mask, self.active, self.first, self.last)[self.slot[who], 0])

    def maybe_commit(self):
        mci = int(self.o.batch_committed(self.pl["match"], uniform_voters=self.S)[0])
        before = self.committed
        adv, self.committed = self.o.raftlog_maybe_commit(self.log.first, 0, self.log.terms, before, mci, self.term)
        # the (term_first, term_last) summary the engine works from decides the same
        tf, tl = self.log.term_range(self.term)
        assert adv == (mci > before and tf <= mci <= tl)
        return self.committed


def test_find_conflict_by_term_table(oracle):
    """log_test.go:58-113 TestFindConflictByTerm, on the oracle and on the tests' own host-log stand-in."""
    t = gu.load_tables()["TestFindConflictByTerm"]
    assert len(t["rows"]) == 21
    for row in t["rows"]:
        terms = row["terms"][1:]
        first, snap = row["from"] + 1, row["terms"][0]
        got, got_term = oracle.find_conflict_by_term(first, snap, terms, row["index"], row["term"])
        assert got == row["want"], row
        log = f7.HostLog([[first + k, tt] for k, tt in enumerate(terms)], snap)
        want_term = log.term(got) or 0  # zeroTermOnOutOfBounds, log_test.go:108-110
        assert got_term == want_term, row
        assert log.find_conflict_by_term(row["index"], row["term"]) == (got, got_term), row


def test_figure7_transcript_on_the_oracle(oracle):
    fx = f7.load()
    assert f7.replay(fx, OracleDriver(oracle, fx)) == 21


def test_figure7_with_a_small_inflight_window(oracle):
    """The transcript never has more than one MsgApp in flight per follower, so MaxInflightMsgs = 1 replays the same."""
    fx = f7.load()
    assert f7.replay(fx, OracleDriver(oracle, fx, cap=1)) == 21
