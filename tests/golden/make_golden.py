#!/usr/bin/env python3
"""Re-encode the reference's golden vectors for the quorum hot path as JSON.

Run in the build container (needs /root/reference, read as TEXT only):

    python tests/golden/make_golden.py

Writes tests/golden/quorum_datadriven.json (the 127 datadriven cases of
quorum/testdata/{majority_commit,joint_commit,majority_vote,joint_vote}.txt)
and tests/golden/tables.json (the small tables of TestCommit,
TestLeaderAcknowledgeCommit, TestProgressUpdate and TestFindConflictByTerm,
transcribed below with their file:line) and tests/golden/figure7.json (the
leader's side of testdata/probe_and_replicate.txt: what it received, what it
sent and what it committed, line by line).  Only inputs and expected outputs are kept: the expected
value of a datadriven stanza is the LAST line of its output block
(quorum/datadriven_test.go:213,222,234,242); `clean` records that the block
holds no "<--" line, i.e. the harness's metamorphic self-checks
(datadriven_test.go:175-219,239) all agreed in the reference.
"""
import json
import os
import re
import sys

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


def main():
    allc = []
    counts = {}
    for f in FILES:
        c = parse_file(f)
        counts[f] = len(c)
        allc.extend(c)
    print(counts, "total", len(allc))
    assert counts == {"majority_commit.txt": 16, "joint_commit.txt": 50,
                      "majority_vote.txt": 22, "joint_vote.txt": 39}, counts
    with open(os.path.join(HERE, "quorum_datadriven.json"), "w") as fh:
        json.dump({"provenance": "quorum/testdata/*.txt @ reference, parsed by make_golden.py",
                   "cases": allc}, fh, indent=1)

    tables = {
        # raft_test.go:1186-1203 — TestCommit: Match per peer, log (index,term), leader term,
        # expected raftLog.committed after maybeCommit (initial committed = 0).
        "TestCommit": {
            "src": "raft_test.go:1186-1203",
            "rows": [
                {"matches": [1], "logs": [[1, 1]], "smTerm": 1, "w": 1},
                {"matches": [1], "logs": [[1, 1]], "smTerm": 2, "w": 0},
                {"matches": [2], "logs": [[1, 1], [2, 2]], "smTerm": 2, "w": 2},
                {"matches": [1], "logs": [[1, 2]], "smTerm": 2, "w": 1},
                {"matches": [2, 1, 1], "logs": [[1, 1], [2, 2]], "smTerm": 1, "w": 1},
                {"matches": [2, 1, 1], "logs": [[1, 1], [2, 1]], "smTerm": 2, "w": 0},
                {"matches": [2, 1, 2], "logs": [[1, 1], [2, 2]], "smTerm": 2, "w": 2},
                {"matches": [2, 1, 2], "logs": [[1, 1], [2, 1]], "smTerm": 2, "w": 0},
                {"matches": [2, 1, 1, 1], "logs": [[1, 1], [2, 2]], "smTerm": 1, "w": 1},
                {"matches": [2, 1, 1, 1], "logs": [[1, 1], [2, 1]], "smTerm": 2, "w": 0},
                {"matches": [2, 1, 1, 2], "logs": [[1, 1], [2, 2]], "smTerm": 1, "w": 1},
                {"matches": [2, 1, 1, 2], "logs": [[1, 1], [2, 1]], "smTerm": 2, "w": 0},
                {"matches": [2, 1, 2, 2], "logs": [[1, 1], [2, 2]], "smTerm": 2, "w": 2},
                {"matches": [2, 1, 2, 2], "logs": [[1, 1], [2, 1]], "smTerm": 2, "w": 0},
            ],
        },
        # raft_paper_test.go:482-490 — TestLeaderAcknowledgeCommit: cluster size, the
        # non-leader peers that accept the entry, whether the entry commits.
        "TestLeaderAcknowledgeCommit": {
            "src": "raft_paper_test.go:482-490",
            "rows": [
                {"size": 1, "acceptors": [], "wack": True},
                {"size": 3, "acceptors": [], "wack": False},
                {"size": 3, "acceptors": [2], "wack": True},
                {"size": 3, "acceptors": [2, 3], "wack": True},
                {"size": 5, "acceptors": [], "wack": False},
                {"size": 5, "acceptors": [2], "wack": False},
                {"size": 5, "acceptors": [2, 3], "wack": True},
                {"size": 5, "acceptors": [2, 3, 4], "wack": True},
                {"size": 5, "acceptors": [2, 3, 4, 5], "wack": True},
            ],
        },
        # log_test.go:66-95 — TestFindConflictByTerm: ents[0] is the snapshot (index, term), the rest
        # the log; want = the index findConflictByTerm(index, term) returns (:106-107); its term is
        # term(want) with out-of-bounds read as 0 (:108-110).
        "TestFindConflictByTerm": {
            "src": "log_test.go:66-95",
            "rows": [{"from": f, "terms": t, "index": i, "term": tm, "want": w}
                     for f, t in ((0, [0, 2, 2, 5, 5, 5]),) for i, tm, w in
                     ((100, 2, 100), (5, 6, 5), (5, 5, 5), (5, 4, 2), (5, 2, 2), (5, 1, 0), (1, 2, 1), (1, 1, 0), (0, 0, 0))]
                    + [{"from": f, "terms": t, "index": i, "term": tm, "want": w}
                       for f, t in ((10, [3, 3, 3, 4, 4, 4]),) for i, tm, w in
                       ((30, 3, 30), (14, 9, 14), (14, 4, 14), (14, 3, 12), (14, 2, 9), (11, 5, 11), (10, 5, 10),
                        (10, 3, 10), (10, 2, 9), (9, 2, 9), (4, 2, 4), (0, 0, 0))],
        },
        # tracker/progress_test.go:124,132-135 — TestProgressUpdate: prev Match=3, Next=5.
        "TestProgressUpdate": {
            "src": "tracker/progress_test.go:124-135",
            "prevM": 3, "prevN": 5,
            "rows": [
                {"update": 2, "wm": 3, "wn": 5, "wok": False},
                {"update": 3, "wm": 3, "wn": 5, "wok": False},
                {"update": 4, "wm": 4, "wn": 5, "wok": True},
                {"update": 5, "wm": 5, "wn": 6, "wok": True},
            ],
        },
    }
    with open(os.path.join(HERE, "tables.json"), "w") as fh:
        json.dump(tables, fh, indent=1)

    fig7 = parse_leader_view("probe_and_replicate.txt", 1, "# Elect node 1 as leader and stabilize.")
    fig7["voters"] = [1, 2, 3, 4, 5, 6, 7]  # add-nodes 7 (probe_and_replicate.txt:34)
    fig7["_doc"] = ("The leader's side of the Raft-paper Figure 7 transcript: votes counted, entries appended, "
                    "MsgAppResp stepped, MsgApp sent (prev = Next-1) and commit indexes shown, in order; "
                    "`line` is the transcript line.  Parsed by make_golden.py, no hand edits.")
    kinds = [e["ev"] for e in fig7["events"]]
    assert kinds.count("recv") == 17 and kinds.count("commit") == 1 and len(fig7["votes"]) == 6, kinds
    with open(os.path.join(HERE, "figure7.json"), "w") as fh:
        json.dump(fig7, fh, indent=1)


if __name__ == "__main__":
    sys.exit(main())
