"""RawNode.Tick scenarios shared by the oracle tests (CPU) and the engine tests (GPU).

The scenarios are the tick loops of the reference's own tests, restated on the batched state (one group per
variation, so one pass ticks all variations at once); each cites the test it follows.  A `ticker` is either
OracleTicker below or EngineTicker (tests/test_gpu_tick.py): the same five planes, the RecentActive bits and one
`tick()` that returns the action bytes and the per-action counts.
"""
import numpy as np

FOLLOWER, CANDIDATE, LEADER, PRE_CANDIDATE = 0, 1, 2, 3
HUP, CHECK_QUORUM, STEP_DOWN, ABORT_TRANSFER, BEAT = 1, 2, 4, 8, 16
PROMOTABLE, TRANSFER = 1, 2


def counts_of(actions):
    return np.array([int(np.count_nonzero(actions & (1 << b))) for b in range(5)], dtype=np.uint64)


class OracleTicker:
    """The CPU restatement of tickElection / tickHeartbeat (oracle/quorum_oracle.c oracle_batch_tick)."""

    def __init__(self, oracle, G, S, election_tick, heartbeat_tick, check_quorum, uniform_voters=0, masks=None):
        self.o, self.G, self.S = oracle, G, S
        self.cfg = (election_tick, heartbeat_tick, check_quorum)
        self.uniform_voters, self.masks = uniform_voters, masks
        mdt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[1 if S <= 8 else 2 if S <= 16 else 4 if S <= 32 else 8]
        self.mdt = mdt
        self.el = np.zeros(G, np.uint32)
        self.hb = np.zeros(G, np.uint32)
        self.rt = np.zeros(G, np.uint32)
        self.role = np.zeros(G, np.uint8)
        self.flags = np.zeros(G, np.uint8)
        self.active = np.zeros(G, mdt)
        self.self_mask = np.zeros(G, mdt)

    def load(self, el=None, hb=None, rt=None, role=None, flags=None, active=None, self_mask=None):
        for name, a in (("el", el), ("hb", hb), ("rt", rt), ("role", role), ("flags", flags), ("active", active),
                        ("self_mask", self_mask)):
            if a is not None:
                cur = getattr(self, name)
                setattr(self, name, np.ascontiguousarray(a, dtype=cur.dtype).copy())

    def heard(self, groups):
        return self.o.tick_heard_from_leader(self.el, groups)

    def tick(self):
        m0, m1, ml = self.masks if self.masks else (None, None, None)
        act = self.o.batch_tick(self.el, self.hb, self.rt, self.role, self.flags, *self.cfg, self.S, self.active,
                                self.self_mask, m0, m1, ml, self.uniform_voters)
        return act, counts_of(act)

    def state(self):
        return self.el.copy(), self.hb.copy(), self.rt.copy(), self.role.copy(), self.flags.copy(), self.active.copy()


# ---------------------------------------------------------------------------
# the reference's tick loops
# ---------------------------------------------------------------------------

def leader_bcast_beat(mk):
    """raft_paper_test.go:109-132 TestLeaderBcastBeat: heartbeat interval hi, a leader of {1,2,3}; after hi ticks
    the leader has sent its heartbeats — and not before.  One group per hi = 1..4."""
    his = [1, 2, 3, 4]
    for hi in his:
        t = mk(G=1, S=3, election_tick=10, heartbeat_tick=hi, check_quorum=False, uniform_voters=3)
        t.load(role=[LEADER], flags=[PROMOTABLE], active=[0b001], self_mask=[0b001])
        for i in range(1, hi + 1):
            act, counts = t.tick()
            assert bool(act[0] & BEAT) == (i == hi), (hi, i, act)
            assert counts[4] == (1 if i == hi else 0)
        assert t.state()[1][0] == 0  # heartbeatElapsed back at zero


def nonleader_start_election(mk):
    """raft_paper_test.go:151-185 testNonleaderStartElection (follower and candidate): et = 10, 2*et - 1 ticks; the
    node campaigns exactly once (r.Term == 2).  One group per (state, randomized timeout in [et, 2et-1]); the
    campaign's reset (raft.go:760-789) is the host's: a new randomized timeout for the group that fired."""
    et = 10
    rts = list(range(et, 2 * et))
    roles = [FOLLOWER] * len(rts) + [CANDIDATE] * len(rts) + [PRE_CANDIDATE] * len(rts)
    rt = np.array(rts * 3, dtype=np.uint32)
    G = len(roles)
    t = mk(G=G, S=3, election_tick=et, heartbeat_tick=1, check_quorum=False, uniform_voters=3)
    t.load(role=roles, flags=[PROMOTABLE] * G, rt=rt)
    fired = np.zeros(G, dtype=np.int64)
    rng = np.random.default_rng(5)
    for i in range(1, 2 * et):
        act, counts = t.tick()
        hup = (act & HUP) != 0
        assert np.array_equal(hup, rt == i) or fired.any(), (i, act)
        assert np.all((act & ~np.uint8(HUP)) == 0)
        assert counts[0] == np.count_nonzero(hup)
        fired += hup
        if hup.any():  # becomeCandidate -> reset: electionElapsed 0 (the tick did that), a fresh timeout
            el, hb, cur_rt, role, flags, _ = t.state()
            assert np.all(el[hup] == 0)
            cur_rt[hup] = et + rng.integers(0, et, size=int(hup.sum()))
            role[hup] = CANDIDATE
            t.load(rt=cur_rt, role=role)
    assert np.all(fired == 1), fired


def learner_election_timeout(mk):
    """raft_test.go:382-398 TestLearnerElectionTimeout: a learner is not promotable (raft.go:1883-1886) and does not
    campaign when its (randomized) timeout passes; beside it the same node as a voter, which does."""
    et = 10
    t = mk(G=2, S=2, election_tick=et, heartbeat_tick=1, check_quorum=False, uniform_voters=1)
    t.load(role=[FOLLOWER, FOLLOWER], flags=[0, PROMOTABLE], rt=[et, et])
    for i in range(1, et + 1):
        act, _ = t.tick()
        assert act[0] == 0
        assert bool(act[1] & HUP) == (i == et)
    el = t.state()[0]
    assert el[0] == et and el[1] == 0  # the learner's counter keeps running (raft.go:824)


def leader_stepdown_when_quorum_active(mk):
    """raft_test.go:1911-1927 TestLeaderStepdownWhenQuorumActive: et = 5, checkQuorum; before each of et + 1 ticks
    peer 2 answers a heartbeat (RecentActive[2] = true, raft.go:1528): still leader."""
    et = 5
    t = mk(G=1, S=3, election_tick=et, heartbeat_tick=1, check_quorum=True, uniform_voters=3)
    t.load(role=[LEADER], flags=[PROMOTABLE], active=[0b001], self_mask=[0b001])
    for i in range(1, et + 2):
        a = t.state()[5]
        a[0] |= 0b010
        t.load(active=a)
        act, counts = t.tick()
        assert bool(act[0] & CHECK_QUORUM) == (i == et), (i, act)
        assert not (act[0] & STEP_DOWN)
        assert act[0] & BEAT  # heartbeat interval 1: MsgBeat at every tick
        if i == et:
            assert t.state()[5][0] == 0b001  # raft.go:1236-1242: everyone but ourselves
    assert t.state()[3][0] == LEADER


def leader_stepdown_when_quorum_lost(mk):
    """raft_test.go:1929-1944 TestLeaderStepdownWhenQuorumLost: the same without any response: follower after
    et + 1 ticks (at tick et, to be exact), and no MsgBeat in the tick that stepped down (raft.go:852-854)."""
    et = 5
    t = mk(G=1, S=3, election_tick=et, heartbeat_tick=1, check_quorum=True, uniform_voters=3)
    t.load(role=[LEADER], flags=[PROMOTABLE | TRANSFER], active=[0b001], self_mask=[0b001], rt=[7])
    for i in range(1, et + 2):
        act, counts = t.tick()
        if i < et:
            assert act[0] == BEAT
        elif i == et:
            assert act[0] == CHECK_QUORUM | STEP_DOWN, act
            assert counts[1] == 1 and counts[2] == 1 and counts[4] == 0 and counts[3] == 0
            el, hb, _, role, flags, active = t.state()
            # reset(): both timers 0, the transfer aborted, every Progress fresh
            assert (el[0], hb[0], role[0], flags[0], active[0]) == (0, 0, FOLLOWER, PROMOTABLE, 0)
        else:
            assert act[0] == 0 and t.state()[0][0] == 1  # a follower's tickElection now
    assert t.state()[3][0] == FOLLOWER


def leader_transfer_timeout(mk):
    """raft_test.go:3788-3813 TestLeaderTransferTimeout (newNetwork: et = 10, hb = 1): the transfer to an isolated
    node is still pending after heartbeatTimeout ticks and aborted after electionTimeout ticks, the node still
    leader (electionElapsed was reset when the transfer began, raft.go stepLeader MsgTransferLeader).  Without
    CheckQuorum and with it (the peers answer, so the quorum is active)."""
    et, hb = 10, 1
    for cq in (False, True):
        t = mk(G=1, S=3, election_tick=et, heartbeat_tick=hb, check_quorum=cq, uniform_voters=3)
        t.load(role=[LEADER], flags=[PROMOTABLE | TRANSFER], active=[0b011], self_mask=[0b001])
        for i in range(1, et + 1):
            act, _ = t.tick()
            flags, role = t.state()[4][0], t.state()[3][0]
            if i < et:
                assert flags & TRANSFER and not (act[0] & ABORT_TRANSFER), i
            else:
                assert act[0] & ABORT_TRANSFER and not (flags & TRANSFER)
                assert bool(act[0] & CHECK_QUORUM) == cq
            assert role == LEADER


def transcript_tick_runs(mk):
    """tests/golden/tick_transcripts.json — what the reference's interaction transcripts print around their tick
    commands (generated by make_golden.py from testdata/*.txt).  A run whose ticked node sends heartbeats or steps
    down is a leader's: the number of ticks that step MsgBeat must equal the MsgHeartbeat count per peer in the node's
    next Ready, every peer but the node itself (learners too) gets them, and CheckQuorum steps the leader down in
    exactly the command whose output says so — checkquorum.txt:56-64: six ticks at ElectionTick 3, the step-down in
    the sixth, five heartbeats per peer, because the tick that steps down sends none (raft.go:852-854).  The peers
    were active when the run starts (they answered in the stabilize before it; checkquorum.txt:53-55 says so).
    Otherwise it is a follower's, with the timeout the transcript set (forget_leader.txt:160-179: no campaign in two
    ticks, a campaign in the third)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tick_transcripts.json")) as fh:
        files = json.load(fh)["files"]
    leader_runs = follower_runs = 0
    for name, f in files.items():
        ids = [int(x) for x in f["nodes"]["voters"]] + [int(x) for x in f["nodes"]["learners"]]
        S, V = len(ids), len(f["nodes"]["voters"])
        et, ht = f["election_tick"], f["heartbeat_tick"]
        drawn = {d["node"]: d["timeout"] for d in f["set_randomized_election_timeout"]}
        tickers = {}
        for run in f["runs"]:
            node = run["ticks"][0]["node"]
            hb = run["next_ready_heartbeats"]
            leader = bool(hb) or any(t["stepped_down"] for t in run["ticks"])
            if not leader and node not in drawn:
                continue  # a follower whose timeout the transcript does not state
            if node not in tickers:
                t = mk(G=1, S=S, election_tick=et, heartbeat_tick=ht, check_quorum=f["nodes"]["checkquorum"], uniform_voters=V)
                me = 1 << ids.index(node)
                t.load(role=[LEADER if leader else FOLLOWER], flags=[PROMOTABLE], rt=[drawn.get(node, 2 * et)],
                       active=[(1 << S) - 1], self_mask=[me])
                tickers[node] = t
            t = tickers[node]
            beats = 0
            for cmd in run["ticks"]:
                acts = 0
                for _ in range(cmd["n_ticks"]):
                    act, _counts = t.tick()
                    acts |= int(act[0])
                    beats += bool(act[0] & BEAT)
                assert bool(acts & STEP_DOWN) == cmd["stepped_down"], (name, cmd["line"])
                assert bool(acts & HUP) == cmd["campaigned"], (name, cmd["line"])
            if leader:
                leader_runs += 1
                if hb is not None:
                    assert sorted(int(k) for k in hb) == sorted(i for i in ids if i != node), (name, run)
                    assert all(v == beats for v in hb.values()), (name, run, beats)
            else:
                follower_runs += 1
    assert leader_runs >= 7 and follower_runs >= 2, (leader_runs, follower_runs)


def follower_hears_from_leader(mk):
    """stepFollower resets electionElapsed on every MsgApp / MsgHeartbeat / MsgSnap from the leader
    (raft.go:1681-1692): a follower that keeps hearing never campaigns, one that stops hearing campaigns one
    randomized timeout after the last message, one that never hears at the timeout itself
    (raft_paper_test.go:151-185, the follower case) and again every timeout after that."""
    et, rt = 10, 13
    t = mk(G=3, S=3, election_tick=et, heartbeat_tick=1, check_quorum=False, uniform_voters=3)
    t.load(role=[FOLLOWER] * 3, flags=[PROMOTABLE] * 3, rt=[rt] * 3)
    fired = [[], [], []]
    for i in range(1, 3 * rt + 1):
        assert t.heard([0] if i > 5 else [0, 1, 1]) == 0
        act, _ = t.tick()
        for g in range(3):
            if act[g] & HUP:
                fired[g].append(i)
    assert fired == [[], [5 + rt - 1, 5 + 2 * rt - 1], [rt, 2 * rt, 3 * rt]], fired
    assert t.heard([2, 7, 1 << 20]) == 2  # ids outside the engine are skipped and counted


SCENARIOS = [transcript_tick_runs, follower_hears_from_leader, leader_bcast_beat, nonleader_start_election, learner_election_timeout, leader_stepdown_when_quorum_active,
             leader_stepdown_when_quorum_lost, leader_transfer_timeout]
