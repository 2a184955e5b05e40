#!/usr/bin/env python3
"""Re-encode the reference's golden vectors for the quorum hot path as JSON — mechanically.

Run in the build container (needs /root/reference, read as TEXT only):

    python tests/golden/make_golden.py

Writes, into tests/golden/:
  quorum_datadriven.json  the 127 datadriven cases of quorum/testdata/{majority_commit,joint_commit,
                          majority_vote,joint_vote}.txt
  tables.json             the tables of TestCommit, TestLeaderAcknowledgeCommit, TestFindConflictByTerm,
                          TestLeaderAppResp, TestLeaderIncreaseNext, TestLeaderElectionInOneRoundRPC,
                          TestLeaderOnlyCommitsLogFromCurrentTerm, TestTerm and TestProgressUpdate, lifted from the Go table literals (golit.py)
  progress_tables.json    every table and imperative case of tracker/progress_test.go and
                          tracker/inflights_test.go, lifted the same way
  figure7.json            the leader's side of testdata/probe_and_replicate.txt, line by line
  transcript_facts.json   what the interaction transcripts print (Progress strings, messages, vote and
                          quorum log lines) with line numbers: what the hand-reconstructed scenarios cite
  confchange_shapes.json  every distinct tracker.Config printed by confchange/testdata/*.txt (shapes of joint
                          configs with learners the reference reaches): inputs for differential tests
Nothing in these six files is typed by hand; tests/test_fixture_provenance.py regenerates them into a
scratch directory whenever /root/reference is present and requires byte identity.  Only inputs and
expected outputs are kept: the expected value of a datadriven stanza is the LAST line of its output block
(quorum/datadriven_test.go:213,222,234,242); `clean` records that the block holds no "<--" line, i.e. the
harness's metamorphic self-checks (datadriven_test.go:175-219,239) all agreed in the reference.
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golit  # noqa: E402

REF = os.environ.get("RAFT_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ["majority_commit.txt", "joint_commit.txt", "majority_vote.txt", "joint_vote.txt"]
ARG = re.compile(r"(\w+)=(\([^)]*\)|\S+)")


def parse_list(v):
    v = v.strip()
    if v.startswith("("):
        v = v[1:-1]
    return [x.strip() for x in v.split(",") if x.strip() != ""]


def parse_file(name):
    path = os.path.join(REF, "quorum", "testdata", name)
    lines = open(path, encoding="utf-8").read().split("\n")
    cases = []
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.startswith("#") or ln.strip() == "":
            i += 1
            continue
        # a command line, followed by ---- and the expected block up to a blank line
        cmdline, lineno = ln, i + 1
        assert lines[i + 1] == "----", (name, lineno)
        j = i + 2
        block = []
        while j < len(lines) and lines[j].strip() != "":
            block.append(lines[j])
            j += 1
        i = j
        parts = cmdline.split(None, 1)
        cmd = parts[0]
        args = dict((k, v) for k, v in ARG.findall(parts[1])) if len(parts) > 1 else {}
        case = {"src": "quorum/testdata/%s:%d" % (name, lineno), "cmd": cmd}
        case["cfg"] = [int(x) for x in parse_list(args.get("cfg", "()"))]
        if "cfgj" in args:
            case["joint"] = True
            case["cfgj"] = [] if args["cfgj"] == "zero" else [int(x) for x in parse_list(args["cfgj"])]
        else:
            case["joint"] = False
            case["cfgj"] = []
        if cmd == "committed":
            case["idx"] = [None if x == "_" else int(x) for x in parse_list(args.get("idx", "()"))]
        else:
            case["votes"] = [None if x == "_" else x for x in parse_list(args.get("votes", "()"))]
        last = block[-1]
        if cmd == "committed":
            # majority_commit.txt:5 prints Describe() without a newline before the index
            last = last.replace("<empty majority quorum>", "")
            case["expected"] = "inf" if last.strip() == "∞" else int(last.strip())
        else:
            case["expected"] = last.strip()
        case["clean"] = not any("<--" in b for b in block)
        cases.append(case)
    return cases


RECV = re.compile(r"^  (\d+)->(\d+) (MsgAppResp|MsgVoteResp) Term:(\d+) Log:(\d+)/(\d+)(?: Rejected \(Hint: (\d+)\))?$")
SEND = re.compile(r"^  (\d+)->(\d+) MsgApp Term:(\d+) Log:(\d+)/(\d+) Commit:(\d+)(?: Entries:\[(.*)\])?$")
ENTRY = re.compile(r"(\d+)/(\d+) Entry")


def parse_leader_view(name, leader, start_marker):
    """The leader's side of an interaction transcript (testdata/<name>), from the line holding
    start_marker on: the votes it counted, the log it appended, every MsgAppResp it stepped, every
    MsgApp it sent and every commit index its Ready showed — in transcript order, which is causal
    order (a Ready lists HardState, then Entries, then the Messages the preceding steps produced)."""
    path = os.path.join(REF, "testdata", name)
    lines = open(path, encoding="utf-8").read().split("\n")
    start = next(i for i, ln in enumerate(lines) if start_marker in ln)
    # the leader's log as `raft-log <leader>` printed it before the election
    k = lines.index("raft-log %d" % leader)
    log = []
    j = k + 2
    while lines[j].strip():
        t, i = ENTRY.match(lines[j]).groups()
        log.append([int(i), int(t)])
        j += 1
    votes, events = [], []
    block, who, sub, term, leading = None, None, None, None, False
    pending_vote = None
    for n in range(start, len(lines)):
        ln = lines[n]
        m = re.match(r"^> (\d+) (receiving messages|handling Ready)$", ln)
        if m:
            who, block, sub = int(m.group(1)), m.group(2), None
            continue
        if not ln.startswith("  "):
            who = None
            continue
        if who != leader:
            continue
        m = re.match(r"^  INFO %d received MsgVoteResp (rejection )?from (\d+) at term (\d+)$" % leader, ln)
        if m:
            pending_vote = {"from": int(m.group(2)), "granted": m.group(1) is None, "line": n + 1}
            term = int(m.group(3))
            continue
        m = re.match(r"^  INFO %d has received (\d+) MsgVoteResp votes and (\d+) vote rejections$" % leader, ln)
        if m:
            pending_vote.update(granted_total=int(m.group(1)), rejected_total=int(m.group(2)), result="VotePending")
            votes.append(pending_vote)
            continue
        if re.match(r"^  INFO %d became leader at term \d+$" % leader, ln):
            votes[-1]["result"] = "VoteWon"
            leading = True
            continue
        if block == "handling Ready":
            if re.match(r"^  (Entries|CommittedEntries|Messages):$", ln):
                sub = ln.strip()[:-1]
                continue
            m = re.match(r"^  HardState Term:\d+(?: Vote:\d+)? Commit:(\d+)", ln)
            if m:
                if not events or not leading:
                    commit0 = int(m.group(1))
                if leading:
                    events.append({"ev": "commit", "value": int(m.group(1)), "line": n + 1})
                continue
            if sub == "Entries" and leading:
                t, i = ENTRY.match(ln.strip()).groups()
                events.append({"ev": "append", "index": int(i), "term": int(t), "line": n + 1})
                continue
            m = SEND.match(ln)
            if m and sub == "Messages" and leading:
                assert int(m.group(1)) == leader
                events.append({"ev": "send", "to": int(m.group(2)), "prev_term": int(m.group(4)),
                               "prev": int(m.group(5)), "commit": int(m.group(6)),
                               "entries": len(ENTRY.findall(m.group(7) or "")), "line": n + 1})
        elif block == "receiving messages":
            m = RECV.match(ln)
            if m and m.group(3) == "MsgAppResp" and int(m.group(2)) == leader:
                ev = {"ev": "recv", "from": int(m.group(1)), "index": int(m.group(6)), "reject": m.group(7) is not None,
                      "line": n + 1}
                if ev["reject"]:
                    ev["log_term"], ev["hint"] = int(m.group(5)), int(m.group(7))
                events.append(ev)
    return {"src": "testdata/" + name, "leader": leader, "term": term, "log": log, "committed": commit0,
            "votes": votes, "events": events}


# ---------------------------------------------------------------------------------------------
# Tables lifted from the reference's Go tests (golit.py reads the literals; nothing is typed here)
# ---------------------------------------------------------------------------------------------

def _go(path):
    return open(os.path.join(REF, path), encoding="utf-8").read()


def _src(path, first, last):
    return "%s:%d-%d" % (path, first, last)


def _zero(rows, ints=(), bools=()):
    for r in rows:
        for k in ints:
            if r.get(k) is None:
                r[k] = 0
        for k in bools:
            if r.get(k) is None:
                r[k] = False
    return rows


def make_tables():
    out = {}
    body, a, b = golit.func_source(_go("raft_test.go"), "TestCommit")
    rows = golit.struct_table(body)
    # Match per peer, log (index, term), leader term, expected raftLog.committed after maybeCommit
    out["TestCommit"] = {"src": _src("raft_test.go", a, b),
                         "rows": [{"matches": r["matches"], "logs": [[e["Index"], e["Term"]] for e in r["logs"]],
                                   "smTerm": r["smTerm"], "w": r["w"]} for r in rows]}
    body, a, b = golit.func_source(_go("raft_paper_test.go"), "TestLeaderAcknowledgeCommit")
    rows = golit.struct_table(body)
    # cluster size, the non-leader peers that accept the entry, whether the entry commits
    out["TestLeaderAcknowledgeCommit"] = {
        "src": _src("raft_paper_test.go", a, b),
        "rows": [{"size": r["size"], "acceptors": sorted(k for k, v in (r["nonLeaderAcceptors"] or {}).items() if v),
                  "wack": r["wack"]} for r in rows]}
    body, a, b = golit.func_source(_go("log_test.go"), "TestFindConflictByTerm")
    rows = golit.struct_table(body)
    # ents(from, terms): ents[0] is the snapshot (index, term), the rest the log; want = the index
    # findConflictByTerm(index, term) returns
    for r in rows:
        assert r["ents"]["call"] == "ents", r
    out["TestFindConflictByTerm"] = {
        "src": _src("log_test.go", a, b),
        "rows": [{"from": r["ents"]["args"][0], "terms": r["ents"]["args"][1], "index": r["index"], "term": r["term"],
                  "want": r["want"]} for r in rows]}
    # stepLeader's MsgAppResp branch end to end (raft.go:1333-1526): a 3-peer leader whose log holds the entries of
    # `ents` plus the empty entry becomeLeader appends; peer 2 answers (index, reject, RejectHint = index); expected
    # Progress[2].Match / Next afterwards, how many messages the leader sent, and their Index (prev) and Commit
    body, a, b = golit.func_source(_go("raft_test.go"), "TestLeaderAppResp")
    rows = golit.struct_table(body)
    peers = re.search(r"withPeers\(([^)]*)\)", body)
    frm = re.search(r"From:\s*(\d+),", body)
    e0 = body.index("[]pb.Entry{", body.index("ents:"))
    e1 = golit.match_brace(body, body.index("{", e0))
    ents = golit.parse_expr(body[e0:e1 + 1])
    assert ents[0] in ({}, []) and rows and peers and frm, (ents, peers, frm)  # ents[0] is MemoryStorage's dummy entry
    out["TestLeaderAppResp"] = {
        "src": _src("raft_test.go", a, b),
        "peers": [int(x) for x in peers.group(1).split(",")], "from": int(frm.group(1)),
        "log": [[e["Index"], e["Term"]] for e in ents[1:]],
        "rows": [{k: (int(v) if not isinstance(v, bool) else v) for k, v in r.items()} for r in rows]}
    # bcastAppend after a proposal (raft.go:1276-1310 -> :689-696 -> maybeSendAppend :600-666 -> UpdateOnEntriesSend,
    # tracker/progress.go:141-166): a 2-peer leader with `previousEnts` + becomeLeader's empty entry, Progress[2] forced
    # to (state, next), one proposal; expected Progress[2].Next afterwards
    body, a, b = golit.func_source(_go("raft_test.go"), "TestLeaderIncreaseNext")
    p0 = body.index("[]pb.Entry{", body.index("previousEnts :="))
    prev = golit.parse_expr(body[p0:golit.match_brace(body, body.index("{", p0)) + 1])
    rows = golit.struct_table(body.replace("len(previousEnts)", str(len(prev))),
                              env={"tracker.StateProbe": 0, "tracker.StateReplicate": 1, "tracker.StateSnapshot": 2})
    peers = re.search(r"withPeers\(([^)]*)\)", body)
    assert len(re.findall(r"Type: pb\.MsgProp", body)) == 1 and len(rows) == 2, rows
    out["TestLeaderIncreaseNext"] = {"src": _src("raft_test.go", a, b), "peers": [int(x) for x in peers.group(1).split(",")],
                                     "log": [[e["Index"], e["Term"]] for e in prev], "rows": rows}
    # raft.poll -> TallyVotes -> VoteResult (raft.go:1041-1049, :1650-1662): a candidate (id 1, votes for itself) in a
    # cluster of `size` receives the MsgVoteResp of `votes` (id -> granted); the role it ends in is the tally's verdict
    # (StateLeader = VoteWon, StateFollower = VoteLost, StateCandidate = VotePending)
    body, a, b = golit.func_source(_go("raft_paper_test.go"), "TestLeaderElectionInOneRoundRPC")
    rows = golit.struct_table(body)
    assert len(rows) == 13 and {r["state"] for r in rows} == {"StateLeader", "StateFollower", "StateCandidate"}, rows
    out["TestLeaderElectionInOneRoundRPC"] = {
        "src": _src("raft_paper_test.go", a, b),
        "rows": [{"size": r["size"], "votes": sorted([int(k), bool(v)] for k, v in (r["votes"] or {}).items()),
                  "state": r["state"]} for r in rows]}
    # raftLog.maybeCommit's term guard through the leader (log.go:447-456 via raft.go:755-758; Raft paper 5.4.2): a
    # 2-peer leader at term 3 over a log of earlier terms appends its empty entry and one proposal; peer 2 acknowledges
    # `index`; expected raftLog.committed
    body, a, b = golit.func_source(_go("raft_paper_test.go"), "TestLeaderOnlyCommitsLogFromCurrentTerm")
    p0 = body.index("[]pb.Entry{", body.index("ents :="))
    ents = golit.parse_expr(body[p0:golit.match_brace(body, body.index("{", p0)) + 1])
    hs = re.search(r"loadState\(pb\.HardState\{Term: (\d+)\}\)", body)
    assert len(re.findall(r"Type: pb\.MsgProp", body)) == 1 and hs
    out["TestLeaderOnlyCommitsLogFromCurrentTerm"] = {
        "src": _src("raft_paper_test.go", a, b), "peers": [int(x) for x in re.search(r"withPeers\(([^)]*)\)", body).group(1).split(",")],
        "log": [[e["Index"], e["Term"]] for e in ents], "term_before_campaign": int(hs.group(1)),
        "rows": golit.struct_table(body)}
    # raftLog.term (log.go:381-407): a log whose snapshot sits at `offset` (term 1) followed by entries offset+i of term
    # i for i in [1, num); term(idx) -> (term, error)
    body, a, b = golit.func_source(_go("log_test.go"), "TestTerm")
    env = {k: int(re.search(r"%s := uint64\((\d+)\)" % k, body).group(1)) for k in ("offset", "num")}
    rows = golit.struct_table(body, env=env)
    out["TestTerm"] = {"src": _src("log_test.go", a, b), "offset": env["offset"], "num": env["num"],
                       "rows": [{"idx": r["idx"], "term": r["term"] or 0, "err": r["err"]} for r in rows]}
    upd = progress_update_table()
    out["TestProgressUpdate"] = {"src": upd["src"], "prevM": upd["prev_match"], "prevN": upd["prev_next"], "rows": upd["rows"]}
    return out


def _consts(body, names):
    """`a, b := uint64(3), uint64(5)` / `match := uint64(1)` at the top of a test body -> {name: value}."""
    env = {}
    for text, _ in golit.statements(body):
        m = re.match(r"^([\w, ]+) := (.*)$", text)
        if not m:
            continue
        lhs = [x.strip() for x in m.group(1).split(",")]
        if not set(lhs) <= set(names):
            continue
        vals = golit.parse_expr("[]x{" + m.group(2) + "}", env)
        env.update(zip(lhs, vals))
    assert set(env) == set(names), (env, names)
    return env


def progress_update_table():
    body, a, b = golit.func_source(_go("tracker/progress_test.go"), "TestProgressUpdate")
    env = _consts(body, ["prevM", "prevN"])
    rows = golit.struct_table(body, env)
    return {"src": _src("tracker/progress_test.go", a, b), "prev_match": env["prevM"], "prev_next": env["prevN"],
            "rows": rows}


def _progress_literal(d):
    """&Progress{State: ..., Match: ...} -> the fields the fixtures use."""
    known = {"State": "state", "Match": "match", "Next": "next", "PendingSnapshot": "pending",
             "MsgAppFlowPaused": "paused"}
    out = {}
    for k, v in d.items():
        if k == "Inflights":
            assert v == {"call": "NewInflights", "args": [256, 0]}, v
            continue
        out[known[k]] = v
    return out


class _Walk:
    """The statements of an imperative test body, consumed one pattern at a time; a statement no
    pattern understands is an error — the generator never guesses."""

    def __init__(self, path, func):
        self.path = path
        body, self.first, self.last = golit.func_source(_go(path), func)
        self.stmts = golit.statements(body)
        self.src = _src(path, self.first, self.last)

    def run(self, patterns):
        for text, _ in self.stmts:
            for rx, fn in patterns:
                m = re.match(rx, text)
                if m:
                    fn(m)
                    break
            else:
                raise golit.GoSyntaxError("%s: statement not understood: %s" % (self.src, text))


def make_progress_tables():
    pt = "tracker/progress_test.go"
    out = {"_comment": "Generated by make_golden.py from the reference's tracker tests: Go table literals and the "
                       "statements of the imperative tests, read as text (golit.py).  No hand edits."}
    body, a, b = golit.func_source(_go(pt), "TestProgressIsPaused")
    out["is_paused"] = {"src": _src(pt, a, b), "rows": golit.struct_table(body)}

    # TestProgressResume: p := &Progress{...}; calls; assert.False(t, p.MsgAppFlowPaused) after some of them
    w = _Walk(pt, "TestProgressResume")
    res = {"src": w.src, "steps": []}
    w.run([
        (r"^p := (&Progress\{.*\})$", lambda m: res.update(_progress_literal(golit.parse_expr(m.group(1))))),
        (r"^p\.MaybeDecrTo\((\d+), (\d+)\)$",
         lambda m: res["steps"].append({"op": "maybe_decr_to", "rejected": int(m.group(1)), "hint": int(m.group(2))})),
        (r"^p\.MaybeUpdate\((\d+)\)$", lambda m: res["steps"].append({"op": "maybe_update", "n": int(m.group(1))})),
        (r"^p\.MsgAppFlowPaused = true$", lambda m: res["steps"].append({"op": "set_paused"})),
        (r"^assert\.False\(t, p\.MsgAppFlowPaused\)$", lambda m: res["steps"][-1].update(paused_after=False)),
    ])
    out["resume"] = res

    body, a, b = golit.func_source(_go(pt), "TestProgressBecomeProbe")
    env = _consts(body, ["match"])
    rows = golit.struct_table(body, env)
    out["become_probe"] = {"src": _src(pt, a, b),
                           "rows": [dict({"pending": 0}, **_progress_literal(r["p"]), wnext=r["wnext"]) for r in rows]}

    for func, key, call_rx, extra in (
            ("TestProgressBecomeReplicate", "become_replicate", r"^p\.BecomeReplicate\(\)$", None),
            ("TestProgressBecomeSnapshot", "become_snapshot", r"^p\.BecomeSnapshot\((\d+)\)$", "snapshoti")):
        w = _Walk(pt, func)
        rec = {"src": w.src}
        want = {}

        def called(m, rec=rec, extra=extra):
            if extra:
                rec[extra] = int(m.group(1))

        def asserted(m, rec=rec, want=want):
            env = {"p.Match": want.get("Match", rec.get("match"))}
            want[m.group(2)] = golit.parse_expr(m.group(1), env)

        w.run([
            (r"^p := (&Progress\{.*\})$", lambda m, rec=rec: rec.update(_progress_literal(golit.parse_expr(m.group(1))))),
            (call_rx, called),
            (r"^assert\.Equal\(t, (.+), p\.(\w+)\)$", asserted),
        ])
        assert want["Match"] == rec["match"], (func, want)  # both tests assert that Match is untouched
        if key == "become_replicate":
            assert want["State"] == "StateReplicate"
            rec["wnext"] = want["Next"]
        else:
            assert want["State"] == "StateSnapshot"
            rec["wpending"] = want["PendingSnapshot"]
        out[key] = rec

    out["maybe_update"] = progress_update_table()
    body, a, b = golit.func_source(_go(pt), "TestProgressMaybeDecr")
    out["maybe_decr"] = {"src": _src(pt, a, b), "rows": golit.struct_table(body)}

    it = "tracker/inflights_test.go"

    def want_literal(d):
        buf = d["buffer"]
        assert buf["call"] == "inflightsBuffer" and len(buf["args"]) == 2, buf
        return {"start": d.get("start", 0), "count": d.get("count", 0), "bytes": d.get("bytes", 0),
                "index": buf["args"][0], "sizes": buf["args"][1]}, d["size"]

    def walk_inflights(func):
        """cases: {var: {"size", "start", "steps": [...]}}, in order of first appearance."""
        w = _Walk(it, func)
        cases, wants = {}, {}

        def new_lit(m):
            d = golit.parse_expr(m.group(2))
            assert d["buffer"] == {"call": "make", "args": ["<type>", d["size"]]}, d
            cases[m.group(1)] = {"size": d["size"], "start": d.get("start", 0), "steps": []}

        def new_call(m):
            assert int(m.group(3)) == 0  # no byte limit in these tests
            cases[m.group(1)] = {"size": int(m.group(2)), "start": 0, "steps": []}

        def add_loop(m):
            assert m.group(1) == m.group(3) == m.group(5) == m.group(7) == m.group(8)  # the loop variable throughout
            cases[m.group(6)]["steps"].append({"add": [int(m.group(2)), int(m.group(4))]})

        def free(m):
            cases[m.group(1)]["steps"].append({"free_le": int(m.group(2))})

        def want(m):
            wants[m.group(1)] = want_literal(golit.parse_expr(m.group(2)))

        def equal(m):
            lit, size = wants.pop(m.group(1))
            c = cases[m.group(2)]
            assert size == c["size"], (size, c)
            assert "want" not in c["steps"][-1]
            c["steps"][-1]["want"] = lit

        w.run([
            (r"^(\w+) := (&Inflights\{.*\})$", lambda m: want(m) if m.group(1).startswith("want") else new_lit(m)),
            (r"^(\w+) := NewInflights\((\d+), (\d+)\)$", new_call),
            (r"^for (\w+) := (\d+); (\w+) < (\d+); (\w+)\+\+ \{ (\w+)\.Add\(uint64\((\w+)\), uint64\(100\+(\w+)\)\) \}$", add_loop),
            (r"^(\w+)\.FreeLE\((\d+)\)$", free),
            (r"^require\.Equal\(t, (\w+), (\w+)\)$", equal),
        ])
        assert not wants
        return w.src, list(cases.values())

    src, cases = walk_inflights("TestInflightsAdd")
    out["inflights_add"] = {"src": src, "cases": cases}
    src, cases = walk_inflights("TestInflightFreeTo")
    assert len(cases) == 1 and cases[0]["start"] == 0
    out["inflights_free_to"] = {"src": src, "size": cases[0]["size"], "steps": cases[0]["steps"]}

    body, a, b = golit.func_source(_go(it), "TestInflightsFull")
    rows = _zero(golit.struct_table(body), ints=("size", "maxBytes", "fullAt", "freeLE", "againAt"))
    out["inflights_full"] = {"src": _src(it, a, b),
                             "_comment": "add (i, 100+i) from 0 until Full at fullAt; FreeLE(freeLE); add on until Full "
                                         "again at againAt; one more Add panics", "rows": rows}

    body, a, b = golit.func_source(_go(it), "TestInflightsReset")

    def one(rx):
        ms = re.findall(rx, body)
        assert len(ms) == 1, (rx, ms)
        return ms[0]

    size, max_bytes = one(r"in := NewInflights\((\d+), (\d+)\)")
    out["inflights_reset"] = {"src": _src(it, a, b), "size": int(size), "maxBytes": int(max_bytes),
                              "epochs": int(one(r"for epoch := 0; epoch < (\d+); epoch\+\+")),
                              "adds_per_epoch": int(one(r"for i := 0; i < (\d+); i\+\+")),
                              "add_bytes": int(one(r"in\.Add\(index, (\d+)\)")),
                              "keep": int(one(r"in\.FreeLE\(index - (\d+)\)"))}
    assert int(one(r"require\.Equal\(t, (\d+), in\.Count\(\)\)\n\t\}")) == out["inflights_reset"]["keep"]
    return out




# ---------------------------------------------------------------------------------------------
# Facts the interaction transcripts print, with their line numbers.  The scenario fixtures that
# are reconstructed by hand (progress.json, scenarios.json: the transcripts run their set-up
# with `log-level none`, so the event stream has to be read off the script) cite transcript
# lines for every expected value; tests/test_fixture_provenance.py checks each citation
# against these facts.
# ---------------------------------------------------------------------------------------------

TRANSCRIPTS = ["heartbeat_resp_recovers_from_probing.txt", "replicate_pause.txt", "snapshot_succeed_via_app_resp.txt",
               "snapshot_succeed_via_app_resp_behind.txt", "checkquorum.txt", "prevote.txt",
               "forget_leader_prevote_checkquorum.txt", "probe_and_replicate.txt"]
PROGRESS_STR = re.compile(r"(?:^(\d+): )?(State(?:Probe|Replicate|Snapshot)) match=(\d+) next=(\d+)( learner)?( paused)?"
                          r"(?: pendingSnap=(\d+))?( inactive)?(?: inflight=(\d+)(\[full\])?)?")
MSG_LINE = re.compile(r"^\s*(\d+)->(\d+) (Msg\w+) Term:(\d+) Log:(\d+)/(\d+)(?: Commit:(\d+))?(?: Vote:(\d+))?"
                      r"(?: Rejected \(Hint: (\d+)\))?(?: Entries:\[(.*)\])?")
INFO_LINE = re.compile(r"^\s*(?:INFO|DEBUG|WARN) (\d+) (received Msg(?:Pre)?VoteResp (?:rejection )?from \d+ at term \d+"
                       r"|has received \d+ Msg(?:Pre)?VoteResp votes and \d+ vote rejections"
                       r"|stepped down to follower since quorum is not active"
                       r"|became (?:leader|candidate|pre-candidate|follower) at term \d+)$")


def transcript_facts(name):
    lines = open(os.path.join(REF, "testdata", name), encoding="utf-8").read().split("\n")
    prog, msgs, info = [], [], []
    for n, ln in enumerate(lines, 1):
        for m in PROGRESS_STR.finditer(ln):
            prog.append({"line": n, "id": int(m.group(1)) if m.group(1) else None, "state": m.group(2),
                         "match": int(m.group(3)), "next": int(m.group(4)), "learner": bool(m.group(5)),
                         "paused": bool(m.group(6)), "pending": int(m.group(7) or 0), "inactive": bool(m.group(8)),
                         "inflight": int(m.group(9) or 0), "full": bool(m.group(10))})
        m = MSG_LINE.match(ln)
        if m:
            rec = {"line": n, "from": int(m.group(1)), "to": int(m.group(2)), "type": m.group(3), "term": int(m.group(4)),
                   "log_term": int(m.group(5)), "index": int(m.group(6)), "commit": int(m.group(7) or 0),
                   "reject": m.group(9) is not None, "entries": len(ENTRY.findall(m.group(10) or ""))}
            if m.group(9) is not None:
                rec["hint"] = int(m.group(9))
            msgs.append(rec)
        m = INFO_LINE.match(ln)
        if m:
            info.append({"line": n, "node": int(m.group(1)), "text": m.group(2)})
    return {"progress": prog, "messages": msgs, "info": info}


CONF_LINE = re.compile(r"^voters=\(([^)]*)\)(?:&&\(([^)]*)\))?(?:\s+learners=\(([^)]*)\))?(?:\s+learners_next=\(([^)]*)\))?"
                       r"(\s+autoleave)?\s*$")


def confchange_shapes():
    """Every tracker.Config the conf-change transcripts print (confchange/testdata/*.txt: the `voters=(..)&&(..)
    learners=(..) learners_next=(..)` line that follows each change, tracker/tracker.go:78-118 Config.String):
    the shapes of incoming / outgoing / learner sets the reference itself reaches.  Distinct shapes, each with the
    first place it is printed."""
    seen = {}
    d = os.path.join(REF, "confchange", "testdata")
    for name in sorted(os.listdir(d)):
        for ln, line in enumerate(open(os.path.join(d, name), encoding="utf-8").read().split("\n"), 1):
            m = CONF_LINE.match(line)
            if not m:
                continue
            ids = [[int(x) for x in (g or "").split()] for g in m.groups()[:4]]
            key = json.dumps(ids)
            if key not in seen:
                seen[key] = {"incoming": ids[0], "outgoing": ids[1], "learners": ids[2], "learners_next": ids[3],
                             "src": "confchange/testdata/%s:%d" % (name, ln)}
    out = sorted(seen.values(), key=lambda c: c["src"])
    assert len(out) >= 15 and any(c["outgoing"] for c in out) and any(c["learners_next"] for c in out), len(out)
    return out


PROGRESS_LINE = re.compile(r"^(\d+): (State\w+) match=(\d+) next=(\d+)( learner)?")


def confchange_init_progress():
    """Changer.initProgress as the conf-change transcripts print it (confchange/confchange.go:257-270: a new peer's
    Progress is {Match: 0, Next: c.LastIndex, StateProbe}).  The datadriven harness starts every file at LastIndex 0
    and increments it after each command, failed ones included (confchange/datadriven_test.go:33-36,48-51), so the
    k-th command of a file (k from 0) runs with LastIndex k; every Progress line of a peer the file had not printed
    before is that peer's initProgress.  One record per first appearance: the LastIndex the command ran with and
    the state / match / next the transcript prints."""
    out = []
    d = os.path.join(REF, "confchange", "testdata")
    for name in sorted(os.listdir(d)):
        lines = open(os.path.join(d, name), encoding="utf-8").read().split("\n")
        known, cmd, i = set(), -1, 0
        while i < len(lines):
            if lines[i].strip() == "----":  # the line before the separator block is the input; a new command starts here
                cmd += 1
                i += 1
                present = set()
                while i < len(lines) and lines[i].strip() != "":
                    m = PROGRESS_LINE.match(lines[i])
                    if m:
                        pid = int(m.group(1))
                        present.add(pid)
                        if pid not in known:
                            out.append({"id": pid, "last_index": cmd, "state": m.group(2), "match": int(m.group(3)),
                                        "next": int(m.group(4)), "learner": bool(m.group(5)),
                                        "src": "confchange/testdata/%s:%d" % (name, i + 1)})
                    i += 1
                if present:  # a failed command prints its error only and changes nothing
                    known = present  # a removed peer that comes back is initialised again
                continue
            i += 1
    assert len(out) >= 25 and sum(1 for r in out if r["next"] == 0) >= 5, len(out)
    assert all(r["next"] == r["last_index"] and r["match"] == 0 and r["state"] == "StateProbe" for r in out), \
        [r for r in out if r["next"] != r["last_index"]][:3]
    return out


TICK_TRANSCRIPTS = ["checkquorum.txt", "heartbeat_resp_recovers_from_probing.txt", "forget_leader.txt",
                    "forget_leader_prevote_checkquorum.txt", "replicate_pause.txt", "snapshot_succeed_via_app_resp.txt",
                    "slow_follower_after_compaction.txt", "confchange_v1_remove_leader.txt"]
TICK_CMD = re.compile(r"^tick-(election|heartbeat) (\d+)\s*$")
ADD_NODES = re.compile(r"^add-nodes (\d+)(.*)$")


def tick_transcripts():
    """What the reference's interaction transcripts show around their tick commands (testdata/*.txt; the harness ticks
    a node ElectionTick or HeartbeatTick times, rafttest/interaction_env_handler_tick.go:23-39, with the defaults of
    rafttest/interaction_env.go raftConfigStub).  Per transcript: the two tick constants, the add-nodes line's sets
    and switches, and every run of tick commands that nothing but comments separates — each command with the lines
    it printed (a step-down, a campaign) — followed, when the very next command is a `stabilize` or `process-ready`,
    by the number of MsgHeartbeat lines per destination in the ticked node's first Ready of that command: the
    heartbeats the ticks produced."""
    stub = _go("rafttest/interaction_env.go")
    et = int(re.search(r"ElectionTick:\s*(\d+)", stub).group(1))
    ht = int(re.search(r"HeartbeatTick:\s*(\d+)", stub).group(1))
    out = {}
    for name in TICK_TRANSCRIPTS:
        lines = open(os.path.join(REF, "testdata", name), encoding="utf-8").read().split("\n")
        nodes, runs, cur, drawn = None, [], None, []
        i = 0
        while i < len(lines):
            ln = lines[i]
            m = re.match(r"^set-randomized-election-timeout (\d+) timeout=(\d+)\s*$", ln)
            if m:
                drawn.append({"line": i + 1, "node": int(m.group(1)), "timeout": int(m.group(2))})
            m = ADD_NODES.match(ln)
            if m and nodes is None:
                rest = m.group(2)

                def ids(key):
                    mm = re.search(key + r"=\(([^)]*)\)", rest)
                    return parse_list(mm.group(1)) if mm and mm.group(1) else []
                nodes = {"line": i + 1, "count": int(m.group(1)), "voters": ids("voters"), "learners": ids("learners"),
                         "checkquorum": "checkquorum=true" in rest, "prevote": "prevote=true" in rest}
            m = TICK_CMD.match(ln)
            if m and i + 1 < len(lines) and lines[i + 1] == "----":
                j = i + 2
                printed = []
                while j < len(lines) and lines[j] != "":
                    if lines[j] != "ok":
                        printed.append(lines[j])
                    j += 1
                if cur is None:
                    cur = {"ticks": [], "next_ready_heartbeats": None}
                    runs.append(cur)
                node = int(m.group(2))
                cur["ticks"].append({"line": i + 1, "kind": m.group(1), "node": node, "n_ticks": et if m.group(1) == "election" else ht,
                                     "stepped_down": any("stepped down to follower since quorum is not active" in x for x in printed),
                                     "campaigned": any("is starting a new election" in x for x in printed),
                                     "printed": len(printed)})
                i = j
                continue
            if cur is not None and ln and not ln.startswith("#"):
                # the first command after the run
                mm = re.match(r"^(stabilize|process-ready)( .*)?$", ln)
                if mm and i + 1 < len(lines) and lines[i + 1] == "----":
                    node = cur["ticks"][-1]["node"]
                    j, inside, hb = i + 2, False, {}
                    while j < len(lines) and lines[j] != "":
                        if lines[j].startswith("> "):
                            if inside:
                                break
                            inside = lines[j] == "> %d handling Ready" % node
                        elif inside or mm.group(1) == "process-ready":
                            h = re.match(r"^\s*%d->(\d+) MsgHeartbeat " % node, lines[j])
                            if h:
                                hb[h.group(1)] = hb.get(h.group(1), 0) + 1
                        j += 1
                    cur["next_ready_heartbeats"] = hb
                    cur["next_command_line"] = i + 1
                cur = None
            i += 1
        out[name] = {"election_tick": et, "heartbeat_tick": ht, "nodes": nodes, "set_randomized_election_timeout": drawn, "runs": runs}
    return out


def write_json(path, obj):
    with open(path, "w") as fh:
        json.dump(obj, fh, indent=1)
        fh.write("\n")


def generate(out_dir):
    """Every generated fixture into out_dir (tests/golden itself, or a scratch directory for the
    regeneration check of tests/test_fixture_provenance.py)."""
    allc = []
    counts = {}
    for f in FILES:
        c = parse_file(f)
        counts[f] = len(c)
        allc.extend(c)
    assert counts == {"majority_commit.txt": 16, "joint_commit.txt": 50,
                      "majority_vote.txt": 22, "joint_vote.txt": 39}, counts
    write_json(os.path.join(out_dir, "quorum_datadriven.json"),
               {"provenance": "quorum/testdata/*.txt @ reference, parsed by make_golden.py", "cases": allc})
    write_json(os.path.join(out_dir, "tables.json"), make_tables())
    write_json(os.path.join(out_dir, "progress_tables.json"), make_progress_tables())
    write_json(os.path.join(out_dir, "transcript_facts.json"),
               {"_doc": "What the reference's interaction transcripts (testdata/*.txt) print, by line: Progress strings "
                        "(status rows carry the peer id), message lines, vote / quorum log lines.  Generated by "
                        "make_golden.py; the hand-reconstructed scenarios cite these lines.",
                "files": {name: transcript_facts(name) for name in TRANSCRIPTS}})
    fig7 = parse_leader_view("probe_and_replicate.txt", 1, "# Elect node 1 as leader and stabilize.")
    fig7["voters"] = [1, 2, 3, 4, 5, 6, 7]  # add-nodes 7 (probe_and_replicate.txt:34)
    fig7["_doc"] = ("The leader's side of the Raft-paper Figure 7 transcript: votes counted, entries appended, "
                    "MsgAppResp stepped, MsgApp sent (prev = Next-1) and commit indexes shown, in order; "
                    "`line` is the transcript line.  Parsed by make_golden.py, no hand edits.")
    kinds = [e["ev"] for e in fig7["events"]]
    assert kinds.count("recv") == 17 and kinds.count("commit") == 1 and len(fig7["votes"]) == 6, kinds
    write_json(os.path.join(out_dir, "figure7.json"), fig7)
    write_json(os.path.join(out_dir, "confchange_shapes.json"),
               {"_doc": "Every distinct tracker.Config the reference's conf-change transcripts print (incoming && outgoing "
                        "voters, learners, learners_next), with the first line that prints it.  Shapes only: the transcripts "
                        "hold no commit indexes, so these are inputs for the differential tests, not expected outputs.",
                "configs": confchange_shapes(),
                "init_progress_doc": "Changer.initProgress as the same transcripts print it: the Progress line of every peer at "
                                     "its first appearance, with the LastIndex its command ran with (the harness increments "
                                     "LastIndex per command, confchange/datadriven_test.go:33-51).  next == last_index in "
                                     "every record (confchange/confchange.go:262).",
                "init_progress": confchange_init_progress()})
    write_json(os.path.join(out_dir, "tick_transcripts.json"),
               {"_doc": "The tick commands of the reference's interaction transcripts with what they printed and the heartbeats "
                        "the ticked node's next Ready holds (make_golden.py tick_transcripts; no hand edits).  Pins "
                        "tickHeartbeat / tickElection (raft.go:823-862): how many MsgBeat a run of ticks steps, when "
                        "CheckQuorum steps a leader down, and that the tick which steps it down sends no heartbeat.",
                "files": tick_transcripts()})
    return counts


GENERATED = ["quorum_datadriven.json", "tables.json", "progress_tables.json", "transcript_facts.json", "figure7.json",
             "confchange_shapes.json", "tick_transcripts.json"]


def main():
    counts = generate(HERE)
    print(counts, "-> ", ", ".join(GENERATED))


if __name__ == "__main__":
    sys.exit(main())
