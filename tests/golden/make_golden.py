#!/usr/bin/env python3
"""Re-encode the reference's golden vectors for the quorum hot path as JSON.

Run in the build container (needs /root/reference, read as TEXT only):

    python tests/golden/make_golden.py

Writes tests/golden/quorum_datadriven.json (the 127 datadriven cases of
quorum/testdata/{majority_commit,joint_commit,majority_vote,joint_vote}.txt)
and tests/golden/tables.json (the small tables of TestCommit,
TestLeaderAcknowledgeCommit and TestProgressUpdate, transcribed below with
their file:line).  Only inputs and expected outputs are kept: the expected
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


if __name__ == "__main__":
    sys.exit(main())
