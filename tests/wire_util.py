"""Serialized raftpb.Message test data.

`encode_message` writes the bytes gogo-protobuf's generated marshaller writes
for raftpb.Message (field order and the always-emitted non-nullable scalars of
raftpb/raft.pb.go:937-1025; field numbers raftpb/raft.proto:71-108).  It is a
test-data generator, checked in tests/test_wire_cpu.py against the protobuf
runtime's own parser on a descriptor built from those field numbers
(`pb_message_class`), i.e. against the published wire format — the reference's
marshaller itself (Go, gogo/protobuf v1.3.2, go.mod:7) cannot run here.
"""
import numpy as np

MSG_APP, MSG_APP_RESP, MSG_HEARTBEAT, MSG_HEARTBEAT_RESP = 3, 4, 8, 9


def varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def tag(field, wt):
    return varint((field << 3) | wt)


def encode_entry(term=0, index=0, type=0, data=None):
    # Entry: Type = 1, Term = 2, Index = 3 always; Data = 4 when non-nil (raft.proto:21-26)
    b = tag(1, 0) + varint(type) + tag(2, 0) + varint(term) + tag(3, 0) + varint(index)
    if data is not None:
        b += tag(4, 2) + varint(len(data)) + data
    return b


def encode_message(type=0, to=0, frm=0, term=0, logTerm=0, index=0, entries=(), commit=0, snapshot=None,
                   reject=False, rejectHint=0, context=None, vote=0, responses=()):
    """entries / responses: already-encoded nested messages; snapshot: encoded Snapshot or None."""
    b = tag(1, 0) + varint(type) + tag(2, 0) + varint(to) + tag(3, 0) + varint(frm) + tag(4, 0) + varint(term)
    b += tag(5, 0) + varint(logTerm) + tag(6, 0) + varint(index)
    for e in entries:
        b += tag(7, 2) + varint(len(e)) + e
    b += tag(8, 0) + varint(commit)
    if snapshot is not None:
        b += tag(9, 2) + varint(len(snapshot)) + snapshot
    b += tag(10, 0) + (b"\x01" if reject else b"\x00") + tag(11, 0) + varint(rejectHint)
    if context is not None:
        b += tag(12, 2) + varint(len(context)) + context
    b += tag(13, 0) + varint(vote)
    for r in responses:
        b += tag(14, 2) + varint(len(r)) + r
    return b


_PB = None


def pb_message_class():
    """A protobuf-runtime message class with raftpb.Message's field numbers and wire types
    (raftpb/raft.proto:21-26,34-37,71-108); enums as int32, SnapshotMetadata as opaque bytes."""
    global _PB
    if _PB is not None:
        return _PB
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="raftpb_wire_test.proto", package="raftpb_wire_test", syntax="proto2")

    def add(msg, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None):
        f = msg.field.add(name=name, number=number, type=ftype, label=label)
        if type_name:
            f.type_name = type_name

    e = fd.message_type.add(name="Entry")
    add(e, "Type", 1, F.TYPE_INT32)
    add(e, "Term", 2, F.TYPE_UINT64)
    add(e, "Index", 3, F.TYPE_UINT64)
    add(e, "Data", 4, F.TYPE_BYTES)
    s = fd.message_type.add(name="Snapshot")
    add(s, "data", 1, F.TYPE_BYTES)
    add(s, "metadata", 2, F.TYPE_BYTES)
    m = fd.message_type.add(name="Message")
    for name, num in (("to", 2), ("from", 3), ("term", 4), ("logTerm", 5), ("index", 6), ("commit", 8),
                      ("rejectHint", 11), ("vote", 13)):
        add(m, name, num, F.TYPE_UINT64)
    add(m, "type", 1, F.TYPE_INT32)
    add(m, "entries", 7, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".raftpb_wire_test.Entry")
    add(m, "snapshot", 9, F.TYPE_MESSAGE, type_name=".raftpb_wire_test.Snapshot")
    add(m, "reject", 10, F.TYPE_BOOL)
    add(m, "context", 12, F.TYPE_BYTES)
    add(m, "responses", 14, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".raftpb_wire_test.Message")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    _PB = message_factory.GetMessageClass(pool.FindMessageTypeByName("raftpb_wire_test.Message"))
    return _PB


def pb_fields(m):
    """The protobuf runtime's view of a parsed message, in the oracle's field-dict shape."""
    return {"type": m.type, "to": m.to, "from": getattr(m, "from"), "term": m.term, "logTerm": m.logTerm,
            "index": m.index, "commit": m.commit, "rejectHint": m.rejectHint, "vote": m.vote, "reject": m.reject,
            "context": m.context if m.HasField("context") else None,
            "nested": any(e.ByteSize() for e in m.entries) or any(r.ByteSize() for r in m.responses)
            or (m.HasField("snapshot") and m.snapshot.ByteSize() > 0)}


def pack(msgs):
    """list of bytes -> (uint8 data, uint64 offsets[n+1])."""
    lens = np.fromiter((len(b) for b in msgs), dtype=np.uint64, count=len(msgs))
    offsets = np.zeros(len(msgs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    data = np.frombuffer(b"".join(msgs), dtype=np.uint8).copy() if msgs else np.zeros(0, dtype=np.uint8)
    return data, offsets


def ctx_words(ctxs):
    """list of bytes (len <= 8, b'' = none pending) -> (ctx u64 little-endian, ctx_len u8)."""
    w = np.array([int.from_bytes(c, "little") for c in ctxs], dtype=np.uint64)
    l = np.array([len(c) for c in ctxs], dtype=np.uint8)
    return w, l


def random_ids(rng, S, G, empty_frac=0.1):
    """[S][G] distinct non-zero raft ids per group, some slots empty (0)."""
    base = rng.integers(1, 1 << 20, size=G, dtype=np.uint64)
    step = rng.integers(1, 1 << 10, size=G, dtype=np.uint64)
    ids = base[None, :] + step[None, :] * np.arange(S, dtype=np.uint64)[:, None]
    big = rng.random(G) < 0.1  # some groups use ids >= 2^63 (ten-byte varints)
    ids[:, big] |= np.uint64(1) << np.uint64(63)
    for g in np.flatnonzero(rng.random(G) < 0.5):  # id order is not slot order
        ids[:, g] = ids[rng.permutation(S), g]
    ids[rng.random((S, G)) < empty_frac] = 0
    return np.ascontiguousarray(ids)


def random_batch(rng, n, G, ids, term, ctxs, bad_frac=0.25):
    """n serialized messages for random groups: mostly well-formed MsgAppResp / MsgHeartbeatResp from
    known peers at the group's term, plus every other case the classifier separates."""
    S = ids.shape[0]
    groups = rng.integers(0, G, size=n).astype(np.uint32)
    out = []
    for i in range(n):
        g = int(groups[i])
        r = rng.random()
        typ = MSG_APP_RESP if rng.random() < 0.6 else MSG_HEARTBEAT_RESP
        frm = int(ids[rng.integers(0, S), g])
        t = int(term[g])
        kw = dict(to=int(rng.integers(0, 1 << 20)), index=int(rng.integers(0, 1 << 40)),
                  reject=bool(rng.random() < 0.1), rejectHint=int(rng.integers(0, 100)),
                  commit=int(rng.integers(0, 1 << 30)), logTerm=int(rng.integers(0, 9)))
        if typ == MSG_HEARTBEAT_RESP:
            c = rng.random()
            if c < 0.4 and len(ctxs[g]):
                kw["context"] = ctxs[g]
            elif c < 0.5:
                kw["context"] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 20)), dtype=np.uint8))
            elif c < 0.6:
                kw["context"] = b""
            elif c < 0.7 and len(ctxs[g]):  # right prefix, wrong length
                kw["context"] = ctxs[g] + b"\x00"
        if r < bad_frac:
            k = int(rng.integers(0, 12))
            if k == 0:
                groups[i] = G + int(rng.integers(0, 5))
            elif k == 1:
                typ = int(rng.choice([0, 3, 5, 6, 8, 10, 19, 300, (1 << 32) + 4]))
            elif k == 2:
                frm = int(rng.integers(1 << 30, 1 << 31))  # nobody
            elif k == 3:
                frm = 0
            elif k == 4:
                t = t - 1 if t > 0 else t + 1
            elif k == 5:
                t = t + 1 + int(rng.integers(0, 3))
            elif k == 6:
                t = 0  # local message
            elif k == 7:
                kw["entries"] = [encode_entry(3, 7, 0, b"payload")]
            elif k == 8:
                kw["snapshot"] = b"" if rng.random() < 0.5 else tag(1, 2) + varint(3) + b"abc"
            elif k == 9:
                kw["responses"] = [b""] if rng.random() < 0.5 else [encode_message(type=4)]
            elif k == 10:
                kw["entries"] = [b""]  # an empty Entry: valid, nothing to decode
        b = encode_message(type=typ, frm=frm, term=t, **kw)
        if r >= bad_frac and rng.random() < 0.15:  # unknown fields, skipped by skipRaft
            extra = [tag(15, 0) + varint(int(rng.integers(0, 1 << 62))), tag(100, 1) + bytes(8), tag(31, 5) + bytes(4),
                     tag(77, 2) + varint(5) + b"hello", tag(20, 3) + tag(21, 0) + varint(9) + tag(20, 4)]
            b += extra[int(rng.integers(0, len(extra)))]
        if r < bad_frac and k == 11 and len(b) > 2:  # damaged bytes
            ba = bytearray(b)
            m = rng.random()
            if m < 0.4:
                del ba[int(rng.integers(1, len(ba))):]
            elif m < 0.8:
                for _ in range(int(rng.integers(1, 4))):
                    ba[int(rng.integers(0, len(ba)))] = int(rng.integers(0, 256))
            else:
                ba += bytes(rng.integers(0, 256, size=int(rng.integers(1, 6)), dtype=np.uint8))
            b = bytes(ba)
        out.append(b)
    return out, groups


def _varint_cols(v):
    """uint64 array -> (uint8 [n][10] varint bytes, int64 lengths)."""
    v = np.ascontiguousarray(v, dtype=np.uint64)
    n = len(v)
    nbits = np.zeros(n, dtype=np.int64)
    t = v.copy()
    for sh in (32, 16, 8, 4, 2, 1):
        big = (t >> np.uint64(sh)) != 0
        nbits += big * sh
        t = np.where(big, t >> np.uint64(sh), t)
    nbits += (t != 0)
    lens = np.maximum((nbits + 6) // 7, 1)
    cols = np.empty((n, 10), dtype=np.uint8)
    for k in range(10):
        cols[:, k] = ((v >> np.uint64(7 * k)) & np.uint64(0x7F)).astype(np.uint8) | ((k < lens - 1) * 0x80).astype(np.uint8)
    return cols, lens


def encode_responses(rng, frm, term, index, ctx8, app_frac=0.6):
    """Vectorised encode_message for n responses (the full-size tests): MsgAppResp with probability
    app_frac, else MsgHeartbeatResp carrying the 8-byte context ctx8 (little-endian u64) half the time;
    5 % stale terms, 3 % unknown senders, 2 % rejects.  Returns (uint8 data, uint64 offsets[n+1])."""
    n = len(frm)
    app = rng.random(n) < app_frac
    typ = np.where(app, MSG_APP_RESP, MSG_HEARTBEAT_RESP).astype(np.uint64)
    term = np.where(rng.random(n) < 0.05, term - np.uint64(1), term).astype(np.uint64)
    frm = np.where(rng.random(n) < 0.03, np.uint64(0x7FFFFFFF), frm).astype(np.uint64)
    reject = (rng.random(n) < 0.02) & app
    has_ctx = ~app & (rng.random(n) < 0.5)
    zero = np.zeros(n, dtype=np.uint64)
    segs = []

    def fixed(byte, present=None):
        segs.append((np.full((n, 1), byte, dtype=np.uint8), np.ones(n, dtype=np.int64) if present is None else present.astype(np.int64)))

    def field(num, vals):
        fixed((num << 3) | 0)
        segs.append(_varint_cols(vals))

    field(1, typ)
    field(2, rng.integers(1, 1 << 20, size=n).astype(np.uint64))
    field(3, frm)
    field(4, term)
    field(5, zero)
    field(6, index)
    field(8, zero)
    field(10, reject.astype(np.uint64))
    field(11, zero)
    fixed((12 << 3) | 2, has_ctx)
    fixed(8, has_ctx)
    segs.append((np.ascontiguousarray(ctx8, dtype="<u8").view(np.uint8).reshape(n, 8), has_ctx * 8))
    field(13, zero)
    M = np.concatenate([c for c, _ in segs], axis=1)
    keep = np.concatenate([np.arange(c.shape[1])[None, :] < l[:, None] for c, l in segs], axis=1)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(keep.sum(axis=1), out=offsets[1:])
    return np.ascontiguousarray(M[keep]), offsets


MSG_VOTE_RESP, MSG_PRE_VOTE_RESP = 6, 18
MSG_TYPES = {"MsgAppResp": MSG_APP_RESP, "MsgHeartbeatResp": MSG_HEARTBEAT_RESP, "MsgVoteResp": MSG_VOTE_RESP,
             "MsgPreVoteResp": MSG_PRE_VOTE_RESP}
CAMPAIGN_OF = {"StatePreCandidate": MSG_PRE_VOTE_RESP, "StateCandidate": MSG_VOTE_RESP, "StateLeader": 0,
               "StateFollower": 0}
STATUS = {"APPLIED": 0, "OTHER_TYPE": 1, "NO_PROGRESS": 2, "STALE_TERM": 3, "HIGHER_TERM": 4, "VOTE": 9}
VOTE_RESULT = {"VotePending": 1, "VoteLost": 2, "VoteWon": 3}


def replay_vote_sequence(seq, drv):
    """One of tests/golden/scenarios.json's wire_vote_sequences through a driver with
    become(resp_type, term) / record([(id, granted)]) / wire(bytes) -> status / tally() -> (res, gr, rj)."""
    for k, st in enumerate(seq["steps"]):
        where = (seq["src"], k)
        if "become" in st:
            drv.become(CAMPAIGN_OF[st["become"]], st["term"])  # every role change resets the votes (raft.go:774-784)
            term = st["term"]
            if "record" in st:
                drv.record([(i, bool(v)) for i, v in st["record"]])
        else:
            m = st["msg"]
            blob = encode_message(type=MSG_TYPES[m["type"]], to=seq["self"], frm=m["from"], term=m["term"], reject=m["reject"])
            assert drv.wire(blob) == STATUS[st["status"]], where
        if "result" in st:
            assert drv.tally() == (VOTE_RESULT[st["result"]], st["granted"], st["rejected"]), where


def random_vote_batch(rng, n, G, ids, term):
    """n serialized responses for random groups: vote and pre-vote responses (granted / rejected; at the group's
    term, one above, one below; from peers with and without a row), a few MsgAppResp between them."""
    S = ids.shape[0]
    groups = rng.integers(0, G, size=n).astype(np.uint32)
    out = []
    for i in range(n):
        g = int(groups[i])
        typ = int(rng.choice([MSG_VOTE_RESP, MSG_PRE_VOTE_RESP, MSG_APP_RESP], p=[0.45, 0.45, 0.1]))
        frm = int(ids[rng.integers(0, S), g]) if rng.random() < 0.95 else int(rng.integers(1, 1 << 20))
        t = int(term[g]) + int(rng.choice([-1, 0, 1], p=[0.1, 0.6, 0.3]))
        out.append(encode_message(type=typ, to=1, frm=frm, term=max(t, 0), reject=bool(rng.random() < 0.4),
                                  index=int(rng.integers(0, 1 << 30))))
    return out, groups
