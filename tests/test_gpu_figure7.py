"""The reference's Raft-paper Figure 7 transcript (testdata/probe_and_replicate.txt, as
tests/golden/figure7.json) replayed on the engine through the C ABI: every response arrives as
serialized raftpb.Message bytes (rafthip_step_wire), rejections come back for the leader's
findConflictByTerm and are stepped with the resolved hint, votes go through
rafthip_record_votes / rafthip_tally_votes, and rafthip_commit_advance must move the commit index
exactly where the transcript does.  G copies of the 7-voter group run side by side, each with its
voters in a different slot order, and all must agree at every step."""
import numpy as np
import pytest

import figure7_util as f7
import progress_util as pu
import raft_amd
import wire_util as wu
from raft_amd import capi

pytestmark = pytest.mark.gpu
po = pu.po


class EngineDriver:
    def __init__(self, e, fx, G, cap):
        self.e, self.G, self.cap = e, G, cap
        self.voters = fx["voters"]
        self.S = S = len(self.voters)
        g = np.arange(G)
        # group g keeps voter k in slot (k - g) mod S
        self.ids = np.array([[self.voters[(s + gg) % S] for gg in g] for s in range(S)], dtype=np.uint64)
        self.groups = g.astype(np.uint32)
        self.log = f7.HostLog(fx["log"])
        e.set_uniform_config(S)
        e.reset_votes()

    def slots(self, who):
        return ((self.voters.index(who) - np.arange(self.G)) % self.S).astype(np.uint8)

    def same(self, a):
        a = np.asarray(a)
        assert (a == a[0]).all(), a
        return int(a[0])

    def events(self, who, typ, index=0, reject=0, flags=0):
        m = np.zeros(self.G, dtype=capi.MSG_DTYPE)
        m["group"], m["slot"], m["type"] = self.groups, self.slots(who), typ
        m["index"], m["reject"], m["flags"] = index, reject, flags
        return m

    def record_vote(self, frm, granted):
        self.e.record_votes(self.events(frm, capi.MSG_VOTE_RESP, reject=0 if granted else 1))
        self.e.tally_votes()
        res, gr, rj = self.e.read_vote_result()
        return self.same(res), self.same(gr), self.same(rj)

    def become_leader(self, me, last, first, committed, term):
        S, G, e = self.S, self.G, self.e
        pl = pu.po.progress_planes(S, G, self.cap)
        pl["next"][:] = last + 1
        mine = self.slots(me)
        pl["match"][mine, np.arange(G)] = last
        pl["pstate"][mine, np.arange(G)] = po.STATE_REPLICATE
        active, acks = pu.mask_planes(S, G)
        e.progress_config(self.cap, 0)
        e.upload_match(pl["match"])
        e.upload_progress(pl["pstate"], pl["next"], pl["pending"])
        e.upload_inflights(pl["infl_meta"], pl["infl_bytes"], pl["ring_index"], pl["ring_bytes"])
        e.upload_active(active)
        e.upload_acks(acks)
        # the entries that carry the leader's term: none yet in a real election — [last+1, last] is empty until the
        # first append (RAFTHIP_MSG_APPENDED) — but a test may hand the leader a log that already holds some
        # (raft_test.go:2617-2625 TestLeaderAppResp)
        tf, tl = self.log.term_range(term)
        assert tl == last
        e.upload_term_range(np.full(G, tf, dtype=np.uint64), np.full(G, tl, dtype=np.uint64))
        e.upload_first_index(np.full(G, first, dtype=np.uint64))
        e.upload_commit(np.full(G, committed, dtype=np.uint64))
        e.upload_ids(self.ids)
        e.upload_term(np.full(G, term, dtype=np.uint64))
        e.upload_read_ctx(np.zeros(G, dtype=np.uint64), np.zeros(G, dtype=np.uint8))
        e.upload_self((np.uint64(1) << mine.astype(np.uint64)).astype(np.uint8))
        self.term, self.me = term, me

    def step(self, kind, who, index=0, reject=False, hint=0, log_term=0, resolved=0, entries=0, nbytes=0):
        e, G = self.e, self.G
        if kind == "appended":
            self.log.append(index, self.term)
            return self.same(e.step_progress(self.events(who, capi.MSG_APPENDED, index=index)))
        if kind == "sent_app":
            return self.same(e.step_progress(self.events(who, capi.MSG_SENT_APP, index=index, flags=1 if entries else 0),
                                             np.full(G, nbytes, dtype=np.uint64)))
        blob = wu.encode_message(type=wu.MSG_APP_RESP, to=self.me, frm=who, term=self.term, index=index, reject=reject,
                                 rejectHint=hint, logTerm=log_term)
        data, offsets = wu.pack([blob] * G)
        msgs, status, counts, act = e.step_wire(data, offsets, self.groups)
        assert (msgs["slot"] == self.slots(who)).all() and (msgs["index"] == index).all()
        if not (reject and log_term > 0):
            assert self.same(status) == capi.WIRE_APPLIED and counts[capi.WIRE_APPLIED] == G
            return self.same(act)
        # raft.go:1458-1461: the hint needs the leader's log; the engine hands the message back unstepped
        assert self.same(status) == capi.WIRE_NEEDS_LOG and self.same(act) == capi.ACT_DEFERRED
        assert (msgs["type"] == capi.MSG_NEEDS_HOST).all()
        assert self.log.find_conflict_by_term(hint, log_term)[0] == resolved
        msgs["type"] = capi.MSG_APP_RESP
        return self.same(e.step_progress(msgs, np.full(G, resolved, dtype=np.uint64)))

    def set_progress(self, who, state, nxt):
        """sm.trk.Progress[who].State / .Next = ... as the reference's tests do by hand."""
        got = self.e.read_progress(rings=False)
        s = self.slots(who)
        got["pstate"][s, np.arange(self.G)] = state
        got["next"][s, np.arange(self.G)] = nxt
        self.e.upload_progress(got["pstate"], got["next"], got["pending"])

    def progress(self, who):
        got = self.e.read_progress(rings=False)
        got["match"] = self.e.read_match()
        s = self.slots(who)
        cells = [pu.cell_state(got, self.cap, 0, int(s[g]), g) for g in sorted({0, min(1, self.G - 1), self.G // 2, self.G - 1})]
        assert all(c == cells[0] for c in cells), cells
        for k in ("match", "next", "pstate"):
            self.same(got[k][s, np.arange(self.G)])
        return cells[0]

    def plan(self, who):
        plan, nxt = self.e.plan_appends()
        s = self.slots(who)
        got = self.same(plan[s, np.arange(self.G)])
        assert self.same(plan[self.slots(self.me), np.arange(self.G)]) == capi.SEND_NONE  # never to itself
        return got

    def maybe_commit(self):
        before = self.e.read_commit()
        pairs, n = self.e.commit_advance()
        after = self.e.read_commit()
        assert n in (0, self.G) and (n == self.G) == bool((after != before).any())
        if n:
            assert np.array_equal(pairs["group"], np.arange(self.G)) and (pairs["commit"] == after).all()
        return self.same(after)


@pytest.mark.parametrize("form", [capi.FLAG_STEP_LINKED, capi.FLAG_STEP_PARTITION], ids=["linked", "partitioned"])
@pytest.mark.parametrize("G,cap", [(1, 256), (777, 256), (4099, 1)])
def test_figure7_transcript_on_the_engine(G, cap, form):
    fx = f7.load()
    with raft_amd.Engine(G, len(fx["voters"]), flags=form) as e:
        drv = EngineDriver(e, fx, G, cap)
        assert f7.replay(fx, drv) == 21
        # every follower caught up and the entry of the leader's term is committed (raft.go:2001-2005)
        assert e.committed_in_term().all()
        hb = e.heartbeat_commit()
        assert (hb == 21).all()
