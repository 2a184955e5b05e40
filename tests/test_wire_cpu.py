"""CPU tests of the serialized-message path's oracle (oracle_message_unmarshal,
oracle_step_wire_msgs): raftpb.Message.Unmarshal restated from
raftpb/raft.pb.go:1801-2179 + skipRaft :3027-3104.

The reference holds no serialized-message vectors and its Go marshaller cannot
run here, so the decode is pinned two ways: (1) on well-formed messages against
the protobuf runtime's parser over a descriptor with raft.proto's field numbers
(the published wire format gogo/protobuf v1.3.2 implements); (2) on malformed
input by vectors derived by hand from the cited lines of Unmarshal — where
gogo's generated code is stricter than the generic runtime (wrong wire type on
a known field, int32 truncation of field numbers) only (2) applies.
"""
import numpy as np
import pytest

import golden_util as gu
import wire_util as wu

OK, EOF, OVERFLOW, BAD_LENGTH, BAD_WIRETYPE, BAD_TAG, END_GROUP = 0, -1, -2, -3, -4, -5, -6


def test_encoder_matches_the_protobuf_runtime(oracle):
    M = wu.pb_message_class()
    rng = np.random.default_rng(1)
    for i in range(300):
        kw = dict(type=int(rng.integers(0, 24)), to=int(rng.integers(0, 1 << 63)), frm=int(rng.integers(0, 1 << 64, dtype=np.uint64)),
                  term=int(rng.integers(0, 1 << 20)), logTerm=int(rng.integers(0, 1 << 20)),
                  index=int(rng.integers(0, 1 << 64, dtype=np.uint64)), commit=int(rng.integers(0, 1 << 40)),
                  reject=bool(rng.integers(0, 2)), rejectHint=int(rng.integers(0, 1 << 30)), vote=int(rng.integers(0, 9)))
        if i % 3 == 0:
            kw["context"] = bytes(rng.integers(0, 256, size=int(rng.integers(0, 12)), dtype=np.uint8))
        if i % 7 == 0:
            kw["entries"] = [wu.encode_entry(2, 5, 0, b"x" * int(rng.integers(0, 4))), b""]
        if i % 11 == 0:
            kw["snapshot"] = b""
        b = wu.encode_message(**kw)
        m = M()
        m.ParseFromString(b)
        want = wu.pb_fields(m)
        assert want["index"] == kw["index"] and want["from"] == kw["frm"] and want["type"] == kw["type"]
        rc, got = oracle.message_unmarshal(b)
        assert rc == 0 and got == want, (kw, got, want)
        # and the runtime's own serialization (unset fields omitted) decodes to the same fields
        rc, got2 = oracle.message_unmarshal(m.SerializeToString())
        assert rc == 0 and got2 == want


def test_unknown_fields_are_skipped_like_the_runtime_skips_them(oracle):
    M = wu.pb_message_class()
    base = wu.encode_message(type=4, frm=7, term=3, index=99)
    extras = [wu.tag(15, 0) + wu.varint(1 << 62), wu.tag(100, 1) + bytes(8), wu.tag(31, 5) + bytes(4),
              wu.tag(77, 2) + wu.varint(5) + b"hello",
              wu.tag(20, 3) + wu.tag(21, 0) + wu.varint(9) + wu.tag(22, 3) + wu.tag(22, 4) + wu.tag(20, 4)]
    for x in extras:
        for b in (base + x, x + base, base[:4] + x + base[4:]):
            rc, got = oracle.message_unmarshal(b)
            m = M()
            m.ParseFromString(b)
            assert rc == 0 and got == wu.pb_fields(m)


# (bytes, rc, expected fields or None) — each line cites what decides it
HAND = [
    (b"", OK, {}),                                                        # loop never runs (:1804)
    (b"\x08", EOF, None),                                                 # tag, then no value byte (:1839)
    (b"\x08" + b"\x80" * 10 + b"\x01", OVERFLOW, None),                   # 11th varint byte (:1836)
    (b"\x08" + b"\xff" * 9 + b"\x7f", OK, {"type": -1}),                  # ten bytes are fine; int32 keeps the low 32 bits (:1844)
    (b"\x30" + b"\xff" * 9 + b"\x7f", OK, {"index": (1 << 64) - 1}),      # bits past 63 fall off (:1938)
    (b"\x0c", END_GROUP, None),                                           # wire type 4 at the top level (:1823)
    (b"\x00", BAD_TAG, None),                                             # field number 0 (:1826)
    (wu.varint((0x80000000 << 3) | 0) + b"\x01", BAD_TAG, None),         # int32(wire >> 3) < 0 (:1821,1826)
    (wu.varint((((1 << 35) | 1) << 3) | 0) + b"\x05", OK, {"type": 5}),   # field number truncated to int32 = 1 (:1821)
    (b"\x09" + bytes(8), BAD_WIRETYPE, None),                             # fixed64 for Type (:1831)
    (b"\x32\x01\x00", BAD_WIRETYPE, None),                                # length-delimited for Index (:1926)
    (b"\x60\x01", BAD_WIRETYPE, None),                                    # varint for Context (:2073)
    (b"\x3a\x05abc", EOF, None),                                          # entry longer than the buffer (:1970)
    (b"\x62" + wu.varint(1 << 63), BAD_LENGTH, None),                     # negative length (:2091)
    (b"\x62" + wu.varint((1 << 63) - 1), BAD_LENGTH, None),               # iNdEx + byteLen overflows int (:2095)
    (b"\x62" + wu.varint(1 << 62), EOF, None),                            # postIndex > l (:2098)
    (b"\x62\x00", OK, {"context": b""}),                                  # empty, not nil (:2102-2104)
    (b"\x62\x02ab\x62\x01c", OK, {"context": b"c"}),                      # last occurrence wins (:2101)
    (b"\x30\x05\x30\x09", OK, {"index": 9}),                              # m.Index = 0, then decoded again (:1929)
    (b"\x50\x02", OK, {"reject": True}),                                  # bool(v != 0) (:2052)
    (b"\x50\x80\x00", OK, {"reject": False}),                             # non-canonical zero
    (b"\x3a\x00", OK, {}),                                                # an empty Entry{} decodes (:1973-1976)
    (b"\x3a\x02\x10\x01", OK, {"nested": True}),                          # a real entry: measured, not decoded here
    (b"\x4a\x00", OK, {}),                                                # empty snapshot (:2026-2031)
    (b"\x72\x00", OK, {}),                                                # empty response
    (b"\x7a\x03abc", OK, {}),                                             # unknown field 15, length-delimited (skipRaft :3064-3083)
    (b"\x7a\x05abc", EOF, None),                                          # ... running past the end (:2168)
    (b"\x7a" + wu.varint(1 << 63), BAD_LENGTH, None),                     # ... negative length (:3080)
    (b"\x79" + bytes(7), EOF, None),                                      # unknown fixed64 cut short (:3063, :2168)
    (b"\x7d" + bytes(3), EOF, None),                                      # unknown fixed32 cut short (:3092)
    (b"\x78" + b"\x80" * 10 + b"\x00", OVERFLOW, None),                   # unknown varint too long (:3051)
    (b"\x7b", EOF, None),                                                 # group never closed (:3103)
    (b"\x7b\x7c", OK, {}),                                                # group opened and closed (:3084-3090)
    (b"\x7b\x7b\x7c", EOF, None),                                         # nested group, one end missing
    (b"\x7b\x08\x05\x7c\x30\x07", OK, {"index": 7}),                      # fields inside a skipped group are not decoded
    (b"\x7e", BAD_WIRETYPE, None),                                        # wire type 6 (:3094)
    (b"\x7f", BAD_WIRETYPE, None),                                        # wire type 7
    (b"\xf8", EOF, None),                                                 # tag varint cut short (:1811)
]


@pytest.mark.parametrize("i", range(len(HAND)))
def test_hand_derived_vectors(oracle, i):
    data, rc, fields = HAND[i]
    got_rc, got = oracle.message_unmarshal(data)
    assert got_rc == rc, (data, got_rc)
    if rc == OK:
        want = {"type": 0, "to": 0, "from": 0, "term": 0, "logTerm": 0, "index": 0, "commit": 0, "rejectHint": 0,
                "vote": 0, "reject": False, "context": None, "nested": False}
        want.update(fields)
        assert got == want


def make_state(rng, S, G):
    mdt = gu.mask_dtype(S)
    ids = wu.random_ids(rng, S, G)
    term = rng.integers(1, 6, size=G).astype(np.uint64)
    ctxs = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 9)), dtype=np.uint8)) if rng.random() < 0.7 else b""
            for _ in range(G)]
    match = rng.integers(0, 1 << 40, size=(S, G), dtype=np.uint64)
    active = np.zeros(G, dtype=mdt)
    acks = np.zeros(G, dtype=mdt)
    return ids, term, ctxs, match, active, acks


def test_step_wire_equals_stepping_the_decoded_records(oracle):
    """The serialized path = decode + the record path (oracle_step_leader_msgs) on the applied messages."""
    rng = np.random.default_rng(5)
    S, G, n = 5, 40, 4000
    ids, term, ctxs, match, active, acks = make_state(rng, S, G)
    msgs, groups = wu.random_batch(rng, n, G, ids, term, ctxs)
    data, offsets = wu.pack(msgs)
    cw, cl = wu.ctx_words(ctxs)
    m1, a1, k1 = match.copy(), active.copy(), acks.copy()
    out, status, counts = oracle.step_wire_msgs(m1, a1, k1, ids, term, cw, cl, data, offsets, groups)
    assert counts.sum() == n and all(counts[:8] > 0), counts  # every class is exercised
    assert np.array_equal(np.bincount(status, minlength=10), counts.astype(np.int64))
    applied = out[status == 0]
    assert (applied["slot"] < S).all() and (out["slot"][status != 0] == 0xFF).all()
    # the ids the applied records resolved to are the senders
    for rec, b in list(zip(out, msgs))[:500]:
        rc, f = oracle.message_unmarshal(b)
        if rc == 0 and rec["slot"] != 0xFF:
            assert int(ids[rec["slot"], rec["group"]]) == f["from"] and f["from"] != 0
    m2, a2, k2 = match.copy(), active.copy(), acks.copy()
    recs = applied.copy()
    recs["flags"] &= 1
    dropped, ignored = oracle.step_leader_msgs(m2, a2, k2, np.zeros(G, dtype=np.uint64), recs)
    assert (dropped, ignored) == (0, 0)
    assert np.array_equal(m1, m2) and np.array_equal(a1, a2) and np.array_equal(k1, k2)
    assert (m1 != match).any() and a1.any() and k1.any()


def test_step_wire_bad_offsets_are_malformed(oracle):
    rng = np.random.default_rng(6)
    S, G = 3, 4
    ids, term, ctxs, match, active, acks = make_state(rng, S, G)
    b = wu.encode_message(type=4, frm=int(ids[ids[:, 0] != 0, 0][0]) if (ids[:, 0] != 0).any() else 1, term=int(term[0]), index=5)
    data, _ = wu.pack([b, b, b])
    L = len(b)
    offsets = np.array([0, L, 5 * L, 3 * L], dtype=np.uint64)  # message 1 ends past the data, message 2 runs backwards
    cw, cl = wu.ctx_words(ctxs)
    _, status, _ = oracle.step_wire_msgs(match, active, acks, ids, term, cw, cl, data, offsets, np.zeros(3, dtype=np.uint32))
    assert status[1] == 6 and status[2] == 6 and status[0] in (0, 2)


def test_vectorised_encoder_decodes_like_the_scalar_one(oracle):
    rng = np.random.default_rng(9)
    n = 2000
    frm = rng.integers(1, 1 << 64, size=n, dtype=np.uint64) >> rng.integers(0, 64, size=n).astype(np.uint64)
    term = rng.integers(1, 1 << 30, size=n).astype(np.uint64)
    index = rng.integers(0, 1 << 64, size=n, dtype=np.uint64) >> rng.integers(0, 64, size=n).astype(np.uint64)
    ctx = rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
    data, offsets = wu.encode_responses(np.random.default_rng(3), frm, term, index, ctx)
    M = wu.pb_message_class()
    kinds = set()
    for i in range(n):
        b = bytes(data[int(offsets[i]):int(offsets[i + 1])])
        rc, got = oracle.message_unmarshal(b)
        m = M()
        m.ParseFromString(b)
        assert rc == 0 and got == wu.pb_fields(m)
        assert got["index"] == int(index[i]) and got["type"] in (4, 9)
        assert got["from"] in (int(frm[i]), 0x7FFFFFFF) and got["term"] in (int(term[i]), int(term[i]) - 1)
        if got["context"] is not None:
            assert got["context"] == int(ctx[i]).to_bytes(8, "little")
        # and the scalar encoder writes the same bytes
        assert b == wu.encode_message(type=got["type"], to=got["to"], frm=got["from"], term=got["term"], index=got["index"],
                                      reject=got["reject"], context=got["context"])
        kinds.add((got["type"], got["context"] is not None, got["reject"]))
    assert len(kinds) >= 4


class OracleVoteDriver:
    """One group of tests/golden/scenarios.json's wire_vote_sequences on the oracle."""

    def __init__(self, oracle, voters):
        self.o, self.voters = oracle, voters
        S = len(voters)
        self.ids = np.array(voters, dtype=np.uint64).reshape(S, 1)
        self.votes = np.zeros((S, 1), dtype=np.uint8)
        self.campaign = np.zeros(1, dtype=np.uint8)
        self.term = np.zeros(1, dtype=np.uint64)
        self.match = np.zeros((S, 1), dtype=np.uint64)
        self.bits = (np.zeros(1, dtype=np.uint8), np.zeros(1, dtype=np.uint8))

    def become(self, resp_type, term):
        self.campaign[0], self.term[0] = resp_type, term
        self.votes[:] = 0

    def record(self, pairs):
        for i, v in pairs:
            cell = self.votes[self.voters.index(i)]
            if cell[0] == 0:
                cell[0] = 2 if v else 1

    def wire(self, blob):
        data, offsets = wu.pack([blob])
        z = np.zeros(1, dtype=np.uint64)
        before = self.votes.copy()
        out, status, counts = self.o.step_wire_msgs(self.match, self.bits[0], self.bits[1], self.ids, self.term, z,
                                                    np.zeros(1, dtype=np.uint8), data, offsets, np.zeros(1, dtype=np.uint32),
                                                    campaign=self.campaign, votes=self.votes)
        assert counts[status[0]] == 1 and counts.sum() == 1
        if status[0] != 9:
            assert np.array_equal(self.votes, before) and out["slot"][0] == 0xFF
        assert not self.match.any() and not self.bits[0].any()  # a vote response touches no Progress
        return int(status[0])

    def tally(self):
        res, gr, rj = self.o.batch_tally_votes(self.votes, uniform_voters=len(self.voters))
        return int(res[0]), int(gr[0]), int(rj[0])


def _vote_sequences():
    import json
    import os
    with open(os.path.join(gu.HERE, "golden", "scenarios.json")) as fh:
        return json.load(fh)["wire_vote_sequences"]


def test_vote_responses_from_the_reference_transcripts(oracle):
    """testdata/prevote.txt and forget_leader_prevote_checkquorum.txt: which vote responses a (pre-)candidate polls
    (raft.go:1068-1076 term gate, :1628-1648 myVoteRespType) and the tallies it logged after each."""
    for seq in _vote_sequences():
        wu.replay_vote_sequence(seq, OracleVoteDriver(oracle, seq["voters"]))


def test_figure7_election_from_the_wire(oracle):
    """The election of testdata/probe_and_replicate.txt:381-468 with every response as serialized bytes."""
    import figure7_util as f7
    fx = f7.load()
    drv = OracleVoteDriver(oracle, fx["voters"])
    drv.become(wu.MSG_VOTE_RESP, fx["term"])
    for v in fx["votes"]:
        if v["from"] == fx["leader"]:
            drv.record([(v["from"], v["granted"])])
        else:
            blob = wu.encode_message(type=wu.MSG_VOTE_RESP, to=fx["leader"], frm=v["from"], term=fx["term"],
                                     reject=not v["granted"])
            assert drv.wire(blob) == 9
        assert drv.tally() == (wu.VOTE_RESULT[v["result"]], v["granted_total"], v["rejected_total"]), v


def test_vote_wire_gates(oracle):
    """Every branch around a vote response: term gate before the type test, the pre-vote exception only for
    grants, the other response type, unknown senders, first vote sticks, nothing without a campaign plane."""
    drv = OracleVoteDriver(oracle, [1, 2, 3])
    enc = lambda **kw: wu.encode_message(to=1, **kw)
    drv.become(wu.MSG_PRE_VOTE_RESP, 5)
    assert drv.wire(enc(type=18, frm=2, term=6, reject=True)) == 4    # higher-term rejection: becomeFollower, the host's
    assert drv.wire(enc(type=18, frm=2, term=9, reject=False)) == 9   # a grant passes with any higher term
    assert drv.wire(enc(type=18, frm=2, term=5, reject=True)) == 9    # polled, but the first vote sticks
    assert drv.tally() == (1, 1, 0)
    assert drv.wire(enc(type=18, frm=3, term=4, reject=False)) == 3   # stale
    assert drv.wire(enc(type=6, frm=3, term=5, reject=False)) == 1    # a MsgVoteResp while pre-campaigning
    assert drv.wire(enc(type=6, frm=3, term=6, reject=False)) == 4    # ... with a higher term: step down first
    assert drv.wire(enc(type=18, frm=7, term=5, reject=False)) == 2   # no row for the sender
    assert drv.wire(enc(type=18, frm=0, term=5, reject=False)) == 2
    assert drv.wire(enc(type=18, frm=3, term=0, reject=True)) == 9    # Term 0: local message, no gate (raft.go:1054)
    assert drv.tally() == (1, 1, 1)
    drv.become(wu.MSG_VOTE_RESP, 6)
    assert drv.wire(enc(type=18, frm=2, term=7, reject=False)) == 1   # the pre-vote exception passes the gate, the type does not match
    assert drv.wire(enc(type=6, frm=2, term=6, reject=True)) == 9 and drv.wire(enc(type=6, frm=3, term=6, reject=True)) == 9
    assert drv.tally() == (2, 0, 2)  # VoteLost
    drv.become(0, 6)
    assert drv.wire(enc(type=6, frm=2, term=6, reject=False)) == 1
    # without a campaign plane vote responses are simply not this path's
    data, offsets = wu.pack([enc(type=6, frm=2, term=6)])
    z = np.zeros(1, dtype=np.uint64)
    _, status, _ = oracle.step_wire_msgs(drv.match, drv.bits[0], drv.bits[1], drv.ids, drv.term, z, np.zeros(1, dtype=np.uint8),
                                         data, offsets, np.zeros(1, dtype=np.uint32))
    assert status[0] == 1


def test_message_marshal_matches_the_published_format(oracle):
    """oracle.message_marshal (Message.Size + MarshalToSizedBuffer, raftpb/raft.pb.go:937-1025,1288-1325) against
    the protobuf runtime's parser, the tests' own forward encoder, and Unmarshal."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(77)
    Msg = wu.pb_message_class()
    for k in range(400):
        big = lambda: int(rng.integers(0, 1 << 64, dtype=np.uint64) >> np.uint64(rng.integers(0, 64)))
        kw = dict(type=int(rng.integers(0, 24)), to=big(), frm=big(), term=big(), logTerm=big(), index=big(), commit=big(),
                  reject=bool(rng.random() < 0.5), rejectHint=big(), vote=big())
        ctx = None if k % 3 == 0 else bytes(rng.integers(0, 256, size=int(rng.integers(0, 12)), dtype=np.uint8))
        b = po.message_marshal(context=ctx, **kw)
        assert b == wu.encode_message(context=ctx, **kw)
        m = Msg()
        m.ParseFromString(b)
        assert (m.type, m.to, getattr(m, "from"), m.term, m.logTerm, m.index, m.commit, m.reject, m.rejectHint, m.vote) == (
            kw["type"], kw["to"], kw["frm"], kw["term"], kw["logTerm"], kw["index"], kw["commit"], kw["reject"],
            kw["rejectHint"], kw["vote"])
        assert m.HasField("context") == (ctx is not None) and m.context == (ctx or b"")
        rc, f = oracle.message_unmarshal(b)
        assert rc == 0 and f["from"] == kw["frm"] and f["commit"] == kw["commit"] and f["context"] == ctx
    # an int32 type below zero is written sign-extended: ten bytes
    assert po.message_marshal(type=-1)[:11] == b"\x08" + b"\xff" * 9 + b"\x01"


def test_bcast_heartbeat_on_the_reference_cases(oracle):
    """The heartbeats the reference's tests show (tests/golden/scenarios.json heartbeat_commit_cases: From, To,
    Term and Commit of `1->2 MsgHeartbeat Term:1 Log:0/0 Commit:17` etc.), marshalled by oracle.bcast_heartbeat
    and parsed back by the protobuf runtime."""
    import json
    import os
    with open(os.path.join(gu.HERE, "golden", "scenarios.json")) as fh:
        cases = json.load(fh)["heartbeat_commit_cases"]
    Msg = wu.pb_message_class()
    for c in cases:
        n = len(c["ids"])
        ids = np.array(c["ids"], dtype=np.uint64).reshape(n, 1)
        me = c["ids"].index(c["leader"])
        for ctx in (b"", b"ctx-7"):
            cw, cl = wu.ctx_words([ctx])
            out, lens = oracle.bcast_heartbeat(np.array(c["match"], dtype=np.uint64).reshape(n, 1),
                                               np.array([c["committed"]], dtype=np.uint64), ids,
                                               np.array([c["term"]], dtype=np.uint64), cw, cl,
                                               np.array([1 << me], dtype=np.uint8))
            assert lens[me, 0] == 0 and not out[me].any()
            for peer, want in c["want"].items():
                s = c["ids"].index(int(peer))
                m = Msg()
                m.ParseFromString(out[s, 0, :lens[s, 0]].tobytes())
                assert (m.type, m.to, getattr(m, "from"), m.term, m.commit) == (8, int(peer), c["leader"], c["term"], want), c["src"]
                assert (m.logTerm, m.index, m.reject, m.rejectHint, m.vote, len(m.entries)) == (0, 0, False, 0, 0, 0)
                assert m.HasField("context") == bool(ctx) and m.context == ctx
                assert not out[s, 0, lens[s, 0]:].any()
