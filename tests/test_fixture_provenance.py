"""Where the fixtures come from, checked.

1. Generated fixtures (tests/golden/make_golden.py: quorum_datadriven, tables, progress_tables,
   transcript_facts, figure7) are regenerated from /root/reference into a scratch directory and must
   be byte-identical to the committed files.  Skipped where the reference is absent (the GPU box).
2. The scenario fixtures that are reconstructed by hand (progress.json, scenarios.json) cite
   transcript lines for their expected values; every such citation is compared with
   transcript_facts.json, which is generated.  Runs everywhere (both sides are committed data).
"""
import json
import os
import re
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, GOLDEN)


def _load(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


@pytest.mark.skipif(not os.path.isdir(os.environ.get("RAFT_REFERENCE", "/root/reference")), reason="reference tree not present")
def test_generated_fixtures_regenerate_byte_identically(tmp_path):
    import make_golden
    make_golden.generate(str(tmp_path))
    for name in make_golden.GENERATED:
        got = open(os.path.join(str(tmp_path), name), "rb").read()
        want = open(os.path.join(GOLDEN, name), "rb").read()
        assert got == want, "%s differs from what make_golden.py produces from the reference" % name
    # and nothing else in tests/golden claims to be generated
    others = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".json") and f not in make_golden.GENERATED)
    assert others == ["progress.json", "scenarios.json"], others


def test_go_literal_reader():
    import golit
    assert golit.parse_expr("[]uint64{1, 2+3, prevM - 1}", {"prevM": 3}) == [1, 5, 2]
    assert golit.parse_expr("&Progress{State: StateProbe, Match: uint64(1), Inflights: NewInflights(256, 0)}") == {
        "State": "StateProbe", "Match": 1, "Inflights": {"call": "NewInflights", "args": [256, 0]}}
    assert golit.parse_expr("map[uint64]bool{2: true, 3: false}") == {2: True, 3: False}
    assert golit.parse_expr("[]pb.Entry{{Index: 1, Term: 1}, {Index: 2, Term: 2}}") == [{"Index": 1, "Term": 1}, {"Index": 2, "Term": 2}]
    rows = golit.struct_table("tests := []struct {\n a, b uint64 // x\n w bool\n}{\n {1, 2, true}, // c\n {a: 3, w: false},\n}\nfor {}")
    assert rows == [{"a": 1, "b": 2, "w": True}, {"a": 3, "b": None, "w": False}]
    with pytest.raises(golit.GoSyntaxError):
        golit.parse_expr("{1, 2")


def _progress_at(facts, lo, hi, peer):
    rows = [r for r in facts["progress"] if lo <= r["line"] <= hi]
    mine = [r for r in rows if r["id"] == peer]
    if mine:
        return mine[-1]
    anon = [r for r in rows if r["id"] is None]
    return anon[-1] if len(anon) == 1 else None


def test_scenario_expectations_agree_with_the_transcripts():
    facts = _load("transcript_facts.json")["files"]
    checked = 0
    for sc in _load("progress.json")["scenarios"]:
        m = re.match(r"testdata/(\S+\.txt):\d+-\d+ \(Progress of peer (\d+) on leader \d+", sc["src"])
        if not m:
            continue  # imperative Go tests: see the fixture's _comment
        f, peer = facts[m.group(1)], int(m.group(2))
        for st in sc["steps"]:
            exp, src = st.get("expect"), st.get("src", "")
            for lo, hi in re.findall(r":(\d+)-(\d+)", src):
                row = _progress_at(f, int(lo), int(hi), peer)
                if row is None or not exp:
                    continue
                got = {"state": row["state"], "match": row["match"], "next": row["next"], "is_paused": row["paused"],
                       "inflight": row["inflight"], "full": row["full"], "pending": row["pending"]}
                if row["id"] is None:
                    # a Progress string inside a log line is printed in the middle of the step (e.g. raft.go:1483 logs
                    # before BecomeProbe / BecomeReplicate): only Match and Next are final there
                    got = {"match": row["match"], "next": row["next"]}
                for k, v in exp.items():
                    if k in got:
                        assert got[k] == v, (sc["src"], src, k, got[k], v)
                        checked += 1
    assert checked >= 55, checked  # the transcripts pin this many expected fields (58 today)


def test_heartbeat_cases_agree_with_the_transcripts():
    facts = _load("transcript_facts.json")["files"]
    checked = 0
    for c in _load("scenarios.json")["heartbeat_commit_cases"]:
        m = re.match(r"testdata/(\S+\.txt):(\d+)-(\d+)", c["src"])
        if not m:
            continue  # raft_test.go TestBcastBeat: imperative Go
        f, lo, hi = facts[m.group(1)], int(m.group(2)), int(m.group(3))
        rows = {r["id"]: r for r in f["progress"] if lo <= r["line"] <= hi and r["id"] is not None}
        assert [rows[i]["match"] for i in c["ids"]] == c["match"], c["src"]
        hb = {m_["to"]: m_ for m_ in f["messages"]
              if lo <= m_["line"] <= hi and m_["type"] == "MsgHeartbeat" and m_["from"] == c["leader"]}
        assert {str(k): v["commit"] for k, v in hb.items()} == c["want"], c["src"]
        assert all(v["term"] == c["term"] for v in hb.values())
        assert c["committed"] == max(c["want"].values())  # some follower is caught up in each of these cases
        checked += 1
    assert checked == 3


def test_vote_and_quorum_sequences_agree_with_the_transcripts():
    """The tallies a campaigning node logs (`has received N ... votes and M vote rejections`), in order, and
    the CheckQuorum step-down line, against the hand-made sequences that cite them."""
    facts = _load("transcript_facts.json")["files"]
    sc = _load("scenarios.json")
    checked = 0
    for seq in sc["wire_vote_sequences"] + sc["vote_sequences"]:
        m = re.match(r"testdata/(\S+\.txt):(\d+)-(\d+)", seq["src"])
        if not m:
            continue
        f, lo, hi = facts[m.group(1)], int(m.group(2)), int(m.group(3))
        me = seq.get("self", seq.get("candidate"))
        logged = []
        for i in f["info"]:
            t = re.match(r"has received (\d+) Msg(?:Pre)?VoteResp votes and (\d+) vote rejections", i["text"])
            if t and lo <= i["line"] <= hi and (me is None or i["node"] == me):
                logged.append((int(t.group(1)), int(t.group(2))))
        # every step that polls (a recorded vote: `record`, or a message with status VOTE) logs one tally
        polled = [(s["granted"], s["rejected"]) for s in seq["steps"]
                  if "granted" in s and (s.get("record") or s.get("status") == "VOTE")]
        # (a poll that happens under `log-level=none` leaves no line: the logged tallies are a prefix)
        assert logged and polled[:len(logged)] == logged, (seq["src"], logged, polled)
        checked += len(logged)
    assert checked >= 7, checked  # tallies the transcripts log for the cited sequences
    for seq in sc["check_quorum_sequences"]:
        m = re.match(r"testdata/(\S+\.txt):(\d+)-(\d+)", seq["src"])
        if not m:
            continue
        f, lo, hi = facts[m.group(1)], int(m.group(2)), int(m.group(3))
        down = [i for i in f["info"] if lo <= i["line"] <= hi and i["text"].startswith("stepped down to follower since quorum")]
        assert len(down) == sum(1 for s in seq["steps"] if not s["quorum_active"]) == 1
        assert down[0]["node"] == seq["self"]
        checked += 1
