"""The C++ host mirror of the Go API (raft_amd/host/: quorum::BatchCommittedIndex,
quorum::BatchVoteResult, tracker::BatchProgressTracker), driven through its
line-oriented CLI the way the reference's datadriven harness drives `quorum`
(quorum/datadriven_test.go:36-250).

CPU part: marshalling of the 127 golden cases into [slot][group] planes, error
behaviour, loud failure without a GPU.  GPU part (marked gpu): the golden
cases end to end, and random multi-group scenarios (conf changes, acks, votes,
CheckQuorum, ReadIndex) against an id-keyed Python model that calls the
per-group oracle functions — i.e. the reference's shapes on both sides.
"""
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# RAFTHOST_CLI points the whole file at another build of the driver (e.g. the
# AddressSanitizer one of tools/asan_host.sh)
CLI = os.environ.get("RAFTHOST_CLI", os.path.join(ROOT, "raft_amd", "lib", "rafthost_cli"))
CASES = gu.load_cases()
NAMES = {1: "VotePending", 2: "VoteLost", 3: "VoteWon"}


def run_cli(script, timeout=120):
    assert os.path.exists(CLI), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    p = subprocess.run([CLI], input=script, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr
    return p.stdout.splitlines()


def ids(xs):
    return ",".join(str(x) for x in sorted(xs))


def case_line(c):
    l = gu.lookuper(c)
    if c["cmd"] == "committed":
        kv = ",".join("%d:%d" % (i, v) for i, v in l.items())
        return "committed in=%s out=%s idx=%s" % (ids(c["cfg"]), ids(c["cfgj"]), kv)
    kv = ",".join("%d:%s" % (i, "y" if v else "n") for i, v in l.items())
    return "vote in=%s out=%s votes=%s" % (ids(c["cfg"]), ids(c["cfgj"]), kv)


def parse_planes(lines):
    """'planes ...' + 'group ...' lines -> list of (S, [(in, out, cells)])."""
    out = []
    for ln in lines:
        t = ln.split()
        if t[0] == "planes":
            kv = dict(x.split("=") for x in t[1:])
            out.append((int(kv["slots"]), int(kv["mask_bytes"]), []))
        elif t[0] == "group":
            kv = dict(x.split("=") for x in t[2:])
            out[-1][2].append((int(kv["in"]), int(kv["out"]), [int(x) for x in kv["cells"].split(",")]))
    return out


def test_marshal_golden_cases_matches_reference_layout():
    script = "\n".join(case_line(c) for c in CASES) + "\nmarshal\n"
    planes = parse_planes(run_cli(script))
    assert len(planes) == 2
    for (S, mb, groups), cmd in zip(planes, ("committed", "vote")):
        plane, m0, m1, _ = gu.cases_to_planes(CASES, cmd)
        assert S == plane.shape[0] and mb == gu.mask_bytes(S)
        assert len(groups) == plane.shape[1]
        for g, (a, b, cells) in enumerate(groups):
            assert a == int(m0[g]) and b == int(m1[g]), (cmd, g)
            assert cells == [int(x) for x in plane[:, g]], (cmd, g)


def test_marshal_rejects_more_than_64_voters():
    out = run_cli("committed in=%s out= idx=\nmarshal\n" % ids(range(1, 66)))
    assert out and out[0].startswith("error -22") and "at most 64" in out[0]


def test_wide_group_uses_wider_masks():
    out = run_cli("committed in=%s out=%s idx=20:7\nmarshal\n" % (ids(range(1, 21)), ids(range(15, 31))))
    (S, mb, groups), = parse_planes(out)
    assert (S, mb) == (30, 4)
    a, b, cells = groups[0]
    assert a == (1 << 20) - 1 and b == ((1 << 30) - 1) & ~((1 << 14) - 1)
    assert cells[19] == 7 and sum(cells) == 7


def test_host_mirror_fails_loudly_without_gpu():
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    out = run_cli("committed in=1 out= idx=1:3\nrun\ntracker 4 3\n")
    assert len(out) == 2 and all(l.startswith("error -19") for l in out), out


# ---------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------

@pytest.mark.gpu
def test_golden_cases_through_batch_entry_points():
    script = "\n".join(case_line(c) for c in CASES) + "\nrun\n"
    out = run_cli(script)
    want = []
    for cmd in ("committed", "vote"):
        for i, c in enumerate(x for x in CASES if x["cmd"] == cmd):
            e = c["expected"]
            want.append("%s %d %s" % (cmd, i, e))
    assert out == want


class Model:
    """tracker.ProgressTracker per group, id-keyed, evaluated with the per-group oracle."""

    def __init__(self, G):
        self.G = G
        self.cfg = [(set(), set(), set()) for _ in range(G)]
        self.match = [dict() for _ in range(G)]
        self.active = [dict() for _ in range(G)]
        self.votes = [dict() for _ in range(G)]
        self.acks = [set() for _ in range(G)]
        self.commit = [0] * G  # the engine's commit plane starts at zero

    def set_config(self, g, inc, out, lrn):
        self.cfg[g] = (set(inc), set(out), set(lrn))
        allv = set(inc) | set(out) | set(lrn)
        for i in list(self.match[g]):
            if i not in allv:  # confchange.go:233-247
                del self.match[g][i], self.active[g][i]
        for i in allv:
            if i not in self.match[g]:  # confchange.go:257-270
                self.match[g][i] = 0
                self.active[g][i] = True

    def reset(self, g):  # raft.go:774-784
        for i in self.match[g]:
            self.match[g][i] = 0
            self.active[g][i] = False
        self.votes[g] = {}

    def ack(self, g, i, n):
        if i not in self.match[g]:
            return False
        self.match[g][i] = max(self.match[g][i], n)  # progress.go:168-177
        return True

    def committed(self, o, g):
        inc, out, _ = self.cfg[g]
        return o.joint_committed_index(sorted(inc), sorted(out), self.match[g])

    def tally(self, o, g):  # tracker.go:269-290
        inc, out, lrn = self.cfg[g]
        gr = sum(1 for i in self.match[g] if i not in lrn and self.votes[g].get(i) is True)
        rj = sum(1 for i in self.match[g] if i not in lrn and self.votes[g].get(i) is False)
        return gr, rj, o.joint_vote_result(sorted(inc), sorted(out), self.votes[g])

    def quorum_active(self, o, g):  # tracker.go:217-227
        inc, out, lrn = self.cfg[g]
        votes = {i: a for i, a in self.active[g].items() if i not in lrn}
        return int(o.joint_vote_result(sorted(inc), sorted(out), votes) == 3)

    def ack_quorum(self, o, g):  # read_only.go:68-76, raft.go:1552
        inc, out, _ = self.cfg[g]
        return o.joint_vote_result(sorted(inc), sorted(out), {i: True for i in self.acks[g]})


def rand_config(rng, S, pool):
    """A confchange-shaped config: outgoing = old voters, incoming = edited copy,
    learners disjoint from both (confchange.go:51-145,276-332)."""
    k = int(rng.integers(1, min(S, 7) + 1))
    old = set(rng.choice(pool, size=k, replace=False).tolist())
    kind = rng.random()
    if kind < 0.4:
        inc, out = old, set()
    else:
        inc = set(old)
        for _ in range(int(rng.integers(1, 3))):
            if rng.random() < 0.5 and len(inc) > 1:
                inc.discard(int(rng.choice(sorted(inc))))
            else:
                inc.add(int(rng.choice(pool)))
        out = old
    lrn = set()
    room = S - len(inc | out)
    if room > 0 and rng.random() < 0.5:
        free = [i for i in pool if i not in inc | out]
        lrn = set(rng.choice(free, size=min(room, int(rng.integers(1, 3)), len(free)), replace=False).tolist())
    while len(inc | out | lrn) > S:
        if lrn:
            lrn.pop()
        elif inc - out:
            inc.discard(sorted(inc - out)[0])
        else:
            inc.discard(out.pop())
    if not inc:
        inc = {int(pool[0])}
    return inc, out, lrn - inc - out


@pytest.mark.gpu
@pytest.mark.parametrize("S,G,seed", [(8, 40, 1), (5, 64, 2), (12, 24, 3)])
def test_tracker_scenarios(oracle, S, G, seed):
    rng = np.random.default_rng(seed)
    pool = np.arange(1, S + 4)
    m = Model(G)
    script, want = ["tracker %d %d" % (G, S)], ["tracker ok"]

    def check_all():
        script.append("commit")
        ch = []
        for g in range(G):
            c = m.committed(oracle, g)
            if c != m.commit[g]:
                ch.append("changed %d %s" % (g, "inf" if c == gu.U64MAX else c))
                m.commit[g] = c
        want.extend(ch)
        want.append("commit-done %d" % len(ch))
        script.append("tally")
        for g in rng.choice(G, size=6, replace=False).tolist():
            script.append("voteof %d" % g)
            gr, rj, r = m.tally(oracle, g)
            want.append("voteof %d %d %d %s" % (g, gr, rj, NAMES[r]))
            script.append("commitof %d" % g)
            want.append("commitof %d %s" % (g, "inf" if m.commit[g] == gu.U64MAX else m.commit[g]))
        script.append("quorumactive")
        want.append("quorumactive " + "".join(str(m.quorum_active(oracle, g)) for g in range(G)))
        script.append("ackquorum")
        want.append("ackquorum " + "".join(str(m.ack_quorum(oracle, g)) for g in range(G)))

    check_all()  # every group still has the empty config: inf, VoteWon, active
    for g in range(G):
        if rng.random() < 0.9:
            inc, out, lrn = rand_config(rng, S, pool)
            m.set_config(g, inc, out, lrn)
            script.append("setconfig %d in=%s out=%s learners=%s" % (g, ids(inc), ids(out), ids(lrn)))
    check_all()
    for rnd in range(12):
        for _ in range(6 * G):
            g = int(rng.integers(0, G))
            i = int(rng.choice(pool))
            op = rng.random()
            if op < 0.5:
                n = int(rng.integers(0, 1000)) if rng.random() < 0.97 else int(rng.integers(1 << 62, 1 << 63)) * 2 + 1
                script.append("ack %d %d %d" % (g, i, n))
                if not m.ack(g, i, n):
                    want.append("dropped")
            elif op < 0.65:
                v = bool(rng.random() < 0.6)
                script.append("recordvote %d %d %s" % (g, i, "y" if v else "n"))
                m.votes[g].setdefault(i, v)  # first vote sticks, any id (tracker.go:260-265)
            elif op < 0.8:
                a = bool(rng.random() < 0.5)
                script.append("active %d %d %d" % (g, i, a))
                if i in m.active[g]:
                    m.active[g][i] = a
                else:
                    want.append("error -22 group %d has no Progress for id %d" % (g, i))
            elif op < 0.92:
                script.append("recvack %d %d" % (g, i))
                m.acks[g].add(i)
            elif op < 0.94:
                script.append("resetvotes %d" % g)
                m.votes[g] = {}
            elif op < 0.96:
                script.append("resetacks %d" % g)
                m.acks[g] = set()
            elif op < 0.98:
                script.append("reset %d" % g)
                m.reset(g)
            else:
                inc, out, lrn = rand_config(rng, S, pool)
                m.set_config(g, inc, out, lrn)
                script.append("setconfig %d in=%s out=%s learners=%s" % (g, ids(inc), ids(out), ids(lrn)))
        g = int(rng.integers(0, G))
        i = int(rng.choice(pool))
        script.append("matchof %d %d" % (g, i))
        if i in m.match[g]:
            want.append("matchof %d %d %d" % (g, i, m.match[g][i]))
        else:
            want.append("error -22 group %d has no Progress for id %d" % (g, i))
        check_all()
    out = run_cli("\n".join(script) + "\n", timeout=300)
    for k, (a, b) in enumerate(zip(out, want)):
        assert a == b, (k, a, b)
    assert len(out) == len(want)


@pytest.mark.gpu
@pytest.mark.parametrize("S,G,seed", [(7, 48, 4), (12, 20, 5)])
def test_tracker_leader_tick(oracle, S, G, seed):
    """The leader-side tick through the host mirror: responses stepped in
    (raft.go:1327-1561), raft.maybeCommit with the term guard (raft.go:755-758,
    log.go:447-456), heartbeat commits (raft.go:669-685), MsgCheckQuorum
    (raft.go:1231-1243) — against the id-keyed model."""
    rng = np.random.default_rng(seed)
    pool = np.arange(1, S + 4)
    m = Model(G)
    committed = [0] * G
    rng_state = [None] * G  # (self, term_first, last_index) for leaders
    script, want = ["tracker %d %d" % (G, S)], ["tracker ok"]
    for g in range(G):
        inc, out, lrn = rand_config(rng, S, pool)
        m.set_config(g, inc, out, lrn)
        script.append("setconfig %d in=%s out=%s learners=%s" % (g, ids(inc), ids(out), ids(lrn)))
        for i in m.active[g]:
            m.active[g][i] = False
            script.append("active %d %d 0" % (g, i))
        if rng.random() < 0.85:
            self_id = int(rng.choice(sorted(inc | out))) if rng.random() < 0.9 else 99  # 99: leader outside the config
            tf, li = int(rng.integers(1, 30)), 0
            li = tf + int(rng.integers(0, 40))
            rng_state[g] = [self_id, tf, li]
            script.append("leader %d %d %d %d" % (g, self_id, tf, li))
            if self_id in m.active[g]:
                m.active[g][self_id] = True
        else:
            script.append("follower %d" % g)
        c = int(rng.integers(0, 20))
        committed[g] = c
        script.append("setcommitted %d %d" % (g, c))

    def tick():
        for _ in range(5 * G):
            g = int(rng.integers(0, G))
            i = int(rng.choice(pool))
            op = rng.random()
            if op < 0.1 and rng_state[g]:
                rng_state[g][2] += int(rng.integers(0, 5))
                script.append("appended %d %d" % (g, rng_state[g][2]))
                continue
            typ = 4 if op < 0.7 else 9
            rej = int(rng.random() < 0.1)
            ctx = int(rng.random() < 0.5)
            top = rng_state[g][2] if rng_state[g] else 30
            idx = int(rng.integers(0, top + 1))
            script.append("step %d %d %d %d %d %d" % (g, i, typ, idx, rej, ctx))
            if i not in m.match[g]:
                want.append("dropped")
                continue
            m.active[g][i] = True
            if typ == 4 and not rej:
                m.ack(g, i, idx)
            if typ == 9 and ctx:
                m.acks[g].add(i)
        script.append("maybecommit")
        adv = []
        for g in range(G):
            mci = m.committed(oracle, g)
            st = rng_state[g]
            if st and mci > committed[g] and st[1] <= mci <= st[2]:
                committed[g] = mci
                adv.append("advanced %d %d" % (g, mci))
        want.extend(adv)
        want.append("maybecommit-done %d" % len(adv))
        for g in rng.choice(G, size=8, replace=False).tolist():
            for i in sorted(m.match[g])[:3]:
                script.append("hbcommit %d %d" % (g, i))
                want.append("hbcommit %d %d %d" % (g, i, min(m.match[g][i], committed[g])))
        script.append("ackquorum")
        want.append("ackquorum " + "".join(str(m.ack_quorum(oracle, g)) for g in range(G)))

    for rnd in range(6):
        tick()
        if rnd % 2 == 1:
            script.append("checkquorum")
            want.append("checkquorum " + "".join(str(m.quorum_active(oracle, g)) for g in range(G)))
            for g in range(G):
                self_id = rng_state[g][0] if rng_state[g] else None
                for i in m.active[g]:
                    if i != self_id:
                        m.active[g][i] = False
            script.append("quorumactive")  # what the next CheckQuorum would see with no traffic
            want.append("quorumactive " + "".join(str(m.quorum_active(oracle, g)) for g in range(G)))
    out = run_cli("\n".join(script) + "\n", timeout=300)
    for k, (a, b) in enumerate(zip(out, want)):
        assert a == b, (k, a, b)
    assert len(out) == len(want)
    assert any(l.startswith("advanced") for l in want)


@pytest.mark.gpu
def test_tracker_raw_node_tick(oracle):
    """RawNode.Tick for every group through the host mirror (SetTickConfig / SetTickState / Tick): tickElection and
    tickHeartbeat (raft.go:823-862) with MsgCheckQuorum and its RecentActive reset (raft.go:1231-1243), heartbeat
    responses stepped in between the ticks — events, counts and the RecentActive state the next CheckQuorum would
    see, against the id-keyed model stepped in the reference's statement order."""
    import tick_scenarios as ts
    from test_tick_oracle import py_tick
    rng = np.random.default_rng(4242)
    G, S, et, ht = 48, 5, 4, 2
    pool = np.arange(1, S + 3)
    m = Model(G)
    st = []  # per group: [el, hb, rt, role, flags, self_id]
    script, want = ["tracker %d %d" % (G, S), "tickconfig %d %d 1" % (et, ht)], ["tracker ok"]
    for g in range(G):
        inc, out, lrn = rand_config(rng, S, pool)
        m.set_config(g, inc, out, lrn)
        script.append("setconfig %d in=%s out=%s learners=%s" % (g, ids(inc), ids(out), ids(lrn)))
        for i in m.active[g]:
            m.active[g][i] = False
            script.append("active %d %d 0" % (g, i))
        role = ts.LEADER if rng.random() < 0.6 else int(rng.choice([ts.FOLLOWER, ts.CANDIDATE, ts.PRE_CANDIDATE]))
        self_id = None
        if role == ts.LEADER:
            self_id = int(rng.choice(sorted(inc | out))) if rng.random() < 0.9 else 99
            script.append("leader %d %d 1 10" % (g, self_id))
            if self_id in m.active[g]:
                m.active[g][self_id] = True
        else:
            script.append("follower %d" % g)
        flags = (ts.PROMOTABLE if rng.random() < 0.8 else 0) | (ts.TRANSFER if role == ts.LEADER and rng.random() < 0.3 else 0)
        st.append([int(rng.integers(0, et)), int(rng.integers(0, ht)), int(rng.integers(et, 2 * et)), role, flags, self_id])
        script.append("tickstate %d %d %d %d %d %d %d" % (g, st[g][0], st[g][1], st[g][2], role, flags & 1, (flags >> 1) & 1))
    mask = ts.HUP | ts.STEP_DOWN | ts.ABORT_TRANSFER
    seen = 0
    for rnd in range(3 * et):
        for _ in range(2 * G):  # heartbeat responses: RecentActive (raft.go:1528)
            g, i = int(rng.integers(0, G)), int(rng.choice(pool))
            script.append("step %d %d 9 0 0 0" % (g, i))
            if i not in m.match[g]:
                want.append("dropped")
            else:
                m.active[g][i] = True
        heard = sorted(set(rng.integers(0, G, size=G // 3).tolist()))  # stepFollower: electionElapsed = 0 (raft.go:1681-1692)
        script.append("heard groups=%s" % ",".join(str(g) for g in heard))
        for g in heard:
            st[g][0] = 0
        script.append("tick %d" % mask)
        counts = [0] * 5
        for g in range(G):
            el, hb, rt, role, flags, self_id = st[g]
            el, hb, role2, flags, _, act = py_tick(el, hb, rt, role, flags, et, ht, True, 1, 1, bool(m.quorum_active(oracle, g)))
            if act & ts.STEP_DOWN:
                for i in m.active[g]:
                    m.active[g][i] = False
                self_id = None
            elif act & ts.CHECK_QUORUM:
                for i in m.active[g]:
                    if i != self_id:
                        m.active[g][i] = False
            st[g] = [el, hb, rt, role2, flags, self_id]
            for b in range(5):
                counts[b] += (act >> b) & 1
            if act & mask:
                want.append("tickevent %d %d" % (g, act))
        seen |= sum(1 << b for b in range(5) if counts[b])
        want.append("tick " + " ".join(str(c) for c in counts))
        if rnd % 4 == 3:
            script.append("quorumactive")
            want.append("quorumactive " + "".join(str(m.quorum_active(oracle, g)) for g in range(G)))
    assert seen == 31, seen  # every action occurred
    out = run_cli("\n".join(script) + "\n", timeout=300)
    for k, (a, b) in enumerate(zip(out, want)):
        assert a == b, (k, a, b)
    assert len(out) == len(want)


@pytest.mark.gpu
def test_tracker_progress_scenarios():
    """The reference's Progress / Inflights scenarios (tests/golden/progress.json) through
    the C++ mirror: peer 2 of a three-voter group, one flush per event and the state /
    action the reference's tests state after each; then the same events in one flush."""
    import progress_util as pu
    from oracle import pyoracle as po

    fx = pu.load_fixture()
    ev_type = pu.EV
    for sc in fx["scenarios"]:
        steps = pu.scenario_steps(sc)
        init = sc["init"]
        head = ["tracker 4 3", "setconfig 2 in=1,2,3 out= learners=", "progress %d %d" % (sc["cap"], sc["max_bytes"]),
                "leader 2 1 1 %d" % sc["last_index"], "firstindex 2 %d" % sc["first_index"],
                "initprogress 2 1 rows=2:%d:%d:%d:%d:%d" % (po.STATE_NAMES[init["state"]], init["match"], init["next"],
                                                              init.get("pending", 0), int(bool(init.get("paused"))))]

        def ev_line(st):
            if st["ev"] == "appended":
                return "appended 2 %d" % st["index"]
            flag = int(bool(st.get("entries"))) if st["ev"] == "sent_app" else 0
            aux = st.get("bytes", 0) if st["ev"] == "sent_app" else st.get("hint", 0)
            return "ev 2 2 %d %d %d %d %d" % (ev_type[st["ev"]], st.get("index", 0), int(bool(st.get("reject"))), flag, aux)

        script = list(head)
        plans = []  # (output line index among the "plan" lines, step, wanted kind)
        for i, st in enumerate(steps):
            want = pu.sent_plan(st, sc["src"])
            if want is not None:  # a send the reference performed from this state: PlanAppends must name it
                script.append("plan 2 1")
                plans.append((i, want))
            script += [ev_line(st), "flushprogress", "progressof 2 2"]
        one_batch = len(steps) <= po.STEP_LIMIT and not any(st["ev"] == "appended" for st in steps)
        if one_batch:
            script += head[2 + 1:]  # re-initialise the same tracker: leader, firstindex, initprogress
            script += [ev_line(st) for st in steps] + ["flushprogress", "progressof 2 2"]
        out = run_cli("\n".join(script) + "\n")
        assert not [l for l in out if l.startswith("error") or l == "dropped"], out[:5]
        acts = [l for l in out if l.startswith("actions")]
        plan_lines = [l for l in out if l.startswith("plan ")]
        assert len(plan_lines) == len(plans)
        for l, (i, want) in zip(plan_lines, plans):
            rows = {int(a): (int(b), int(c)) for a, b, c in (t.split(":") for t in l.split()[2].split(","))}
            assert set(rows) == {2, 3} and rows[2][0] == want, (sc["src"], i, l)  # never to the leader itself (id 1)
        views = [dict(kv.split("=") for kv in l.split()[3:]) for l in out if l.startswith("progressof")]
        assert len(acts) == len(views) == len(steps) + (1 if one_batch else 0)
        for i, st in enumerate(steps):
            got = {k: int(v) for k, v in views[i].items()}
            got["is_paused"], got["paused"], got["full"] = bool(got["is_paused"]), bool(got["paused"]), bool(got["full"])
            pu.check_expect(st, got, (sc["src"], i))
            pu.check_actions(st, int(acts[i].split()[1]) if len(acts[i].split()) > 1 else 0, (sc["src"], i))
            if st["ev"] in ("app_resp", "heartbeat_resp"):
                assert got["active"] == 1
        if one_batch:
            assert views[-1] == views[len(steps) - 1], sc["src"]
            assert len(acts[-1].split()[1].split(",")) == len(steps)


@pytest.mark.gpu
def test_tracker_bcast_heartbeat(oracle):
    """BcastHeartbeat through the host mirror: the heartbeats the reference's tests show
    (tests/golden/scenarios.json heartbeat_commit_cases), each as the bytes Message.Marshal writes — compared with
    the oracle's marshaller and parsed back; with and without a pending ReadIndex context; a group in the middle of
    a tracker with other groups around it."""
    import json
    from oracle import pyoracle as po
    with open(os.path.join(gu.HERE, "golden", "scenarios.json")) as fh:
        cases = json.load(fh)["heartbeat_commit_cases"]
    G, S = len(cases) + 2, 4
    script, want = ["tracker %d %d" % (G, S)], ["tracker ok"]
    for k, c in enumerate(cases):
        g = k + 1
        script.append("setconfig %d in=%s out= learners=" % (g, ids(c["ids"])))
        script.append("leader %d %d 1 %d" % (g, c["leader"], max(c["match"])))
        for i, mt in zip(c["ids"], c["match"]):
            script.append("ack %d %d %d" % (g, i, mt))
        script.append("setcommitted %d %d" % (g, c["committed"]))
        for ctx in (b"", b"rd-%d" % k):
            script.append(("heartbeats %d %d %s" % (g, c["term"], ctx.hex())).rstrip())
            for peer in sorted(int(p) for p in c["want"]):
                blob = po.message_marshal(type=8, to=peer, frm=c["leader"], term=c["term"], commit=c["want"][str(peer)],
                                          context=ctx or None)
                want.append("heartbeat %d %d %s" % (g, peer, blob.hex()))
                rc, f = oracle.message_unmarshal(blob)
                assert rc == 0 and (f["to"], f["from"], f["term"], f["commit"]) == (peer, c["leader"], c["term"], c["want"][str(peer)])
            want.append("heartbeats-done %d" % len(c["want"]))
    # a group that is not led here has nothing to send; a context longer than the engine tracks is an error
    script += ["setconfig 0 in=1,2,3 out= learners=", "follower 0", "heartbeats 0 1", "heartbeats 1 1 %s" % ("ab" * 9)]
    want += ["heartbeats-done 0"]
    out = [l for l in run_cli("\n".join(script) + "\n") if not l.endswith(" ok")][0:]
    want = [l for l in want if not l.endswith(" ok")]
    assert out[:len(want)] == want, [(a, b) for a, b in zip(out, want) if a != b][:3]
    assert len(out) == len(want) + 1 and out[-1].startswith("error "), out[-1]


@pytest.mark.gpu
def test_setconfig_new_peer_starts_as_the_reference_transcripts_print_it():
    """Changer.initProgress (confchange/confchange.go:257-270) against the reference's own output: every first
    appearance of a peer in confchange/testdata/*.txt (tests/golden/confchange_shapes.json "init_progress", lifted
    by make_golden.py) gives the LastIndex its command ran with and the Progress line it printed — `next` equals
    that LastIndex, 0 included (the mirror used to start such a peer at Next 1)."""
    import json
    recs = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "confchange_shapes.json")))["init_progress"]
    assert any(r["last_index"] == 0 for r in recs) and any(r["last_index"] > 0 for r in recs)
    script = ["tracker %d 3" % len(recs)]
    for g, r in enumerate(recs):
        keeper = 1000 + r["id"]  # a peer that is already there, so that the new one is added by a conf change
        script.append("setconfig %d in=%d out= learners=" % (g, keeper))
    script.append("progress 2 0")
    for g, r in enumerate(recs):
        keeper = 1000 + r["id"]
        if r["learner"]:
            script.append("setconfig %d in=%d out= learners=%d last=%d" % (g, keeper, r["id"], r["last_index"]))
        else:
            script.append("setconfig %d in=%d,%d out= learners= last=%d" % (g, keeper, r["id"], r["last_index"]))
        script.append("progressof %d %d" % (g, r["id"]))
    out = run_cli("\n".join(script) + "\n")
    assert not [l for l in out if l.startswith("error") or l == "dropped"], out
    views = [dict(kv.split("=") for kv in l.split()[3:]) for l in out if l.startswith("progressof")]
    assert len(views) == len(recs)
    for r, v in zip(recs, views):
        assert r["state"] == "StateProbe" and (v["state"], v["match"], v["next"]) == ("0", str(r["match"]), str(r["next"])), (r, v)


@pytest.mark.gpu
def test_setconfig_reuses_a_slot_with_a_fresh_progress():
    """A removed peer's slot goes to a new id: the new peer must start as Changer.initProgress creates it
    (confchange/confchange.go:257-270: StateProbe, Match 0, Next = LastIndex, empty Inflights) — not
    with the removed peer's StateReplicate, Next and full Inflights — and the surviving peer keeps everything."""
    from oracle import pyoracle as po
    rep = po.STATE_NAMES["StateReplicate"]
    script = ["tracker 4 3", "setconfig 2 in=1,2,3 out= learners=", "progress 2 0", "leader 2 1 1 20", "firstindex 2 1",
              # peers 2 and 3 replicate at 10/11; both get two MsgApps in flight: Inflights full (cap 2)
              "initprogress 2 1 rows=2:%d:10:11:0:0,3:%d:10:11:0:0" % (rep, rep)]
    for peer in (2, 3):
        script += ["ev 2 %d %d 11 0 1 100" % (peer, 0xF1), "ev 2 %d %d 12 0 1 100" % (peer, 0xF1)]
    script += ["flushprogress", "progressof 2 2", "progressof 2 3",
               # 3 leaves, 4 joins as a learner (takes 3's slot); the leader's last index is 20
               "setconfig 2 in=1,2 out= learners=4 last=20", "progressof 2 4", "progressof 2 2", "plan 2 1"]
    out = run_cli("\n".join(script) + "\n")
    assert not [l for l in out if l.startswith("error") or l == "dropped"], out
    views = [dict(kv.split("=") for kv in l.split()[3:]) for l in out if l.startswith("progressof")]
    full2, full3, new4, kept2 = views
    assert full2["state"] == full3["state"] == str(rep) and full2["full"] == full3["full"] == "1" and full3["next"] == "13"
    assert (new4["state"], new4["match"], new4["next"], new4["full"], new4["paused"]) == ("0", "0", "20", "0", "0")
    assert kept2 == full2  # the surviving peer is left alone
    plan = [l for l in out if l.startswith("plan ")][0]
    rows = {int(a): (int(b), int(c)) for a, b, c in (t.split(":") for t in plan.split()[2].split(","))}
    # a probing, unpaused new learner is sent entries from Next = 20 (the log has them: last index 20); the full
    # replicating peer is paused
    assert rows[4] == (1, 20) and rows[2][0] == 0, plan
