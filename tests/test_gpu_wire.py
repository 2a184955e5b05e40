"""GPU parity for the serialized-message ingest:
rafthip_ingest_wire decodes gogo-protobuf raftpb.Message bytes on the device
(raftpb/raft.pb.go:1801-2179, skipRaft :3027-3104), applies raft.Step's term
gate (raft.go:1051-1139) and the Progress lookup of raft.go:1327, then the
stepLeader bookkeeping.  Bit-exact vs the oracle stepping the same bytes one
message at a time; through the C ABI."""
import numpy as np
import pytest

import golden_util as gu
import raft_amd
import wire_util as wu
from raft_amd import capi
from test_wire_cpu import HAND, OK, make_state

pytestmark = pytest.mark.gpu


def run_both(oracle, e, state, data, offsets, groups, device_inputs=False):
    ids, term, ctxs, match, active, acks = state
    cw, cl = wu.ctx_words(ctxs)
    wm, wa, wk = match.copy(), active.copy(), acks.copy()
    want_msgs, want_status, want_counts = oracle.step_wire_msgs(wm, wa, wk, ids, term, cw, cl, data, offsets, groups)
    e.upload_match(match)
    e.upload_active(active)
    e.upload_acks(acks)
    e.upload_ids(ids)
    e.upload_term(term)
    e.upload_read_ctx(cw, cl)
    if device_inputs == "pinned":  # page-locked host memory: the kernel reads it over the link, no staging copy
        pins = [raft_amd.PinnedArray(max(len(data), 1), np.uint8), raft_amd.PinnedArray(len(offsets), np.uint64),
                raft_amd.PinnedArray(len(groups), np.uint32)]
        pins[0].array[:len(data)] = data
        pins[1].array[:] = offsets
        pins[2].array[:] = groups
        msgs, status, counts = e.ingest_wire(pins[0].array, pins[1].array, pins[2].array, n=len(groups))
        pins[0].array[:] = 0xFF  # the call has returned: the buffers are the caller's again
        for p in pins:
            p.free()
    elif device_inputs:
        import torch
        keep = [torch.from_numpy(data.copy()).cuda() if len(data) else torch.zeros(1, dtype=torch.uint8, device="cuda"),
                torch.from_numpy(offsets.view(np.int64).copy()).cuda(),
                torch.from_numpy(groups.view(np.int32).copy()).cuda()]
        msgs, status, counts = e.ingest_wire(keep[0], keep[1], keep[2], n=len(groups))
    else:
        msgs, status, counts = e.ingest_wire(data, offsets, groups)
    assert np.array_equal(status, want_status)
    assert np.array_equal(counts, want_counts) and counts[8] == 0 and counts[9] == 0
    assert np.array_equal(msgs, want_msgs)
    assert np.array_equal(e.read_match(), wm)
    assert np.array_equal(e.read_active(), wa)
    assert np.array_equal(e.read_acks(), wk)
    return want_counts


@pytest.mark.parametrize("S,device_inputs", [(3, False), (5, True), (9, "pinned"), (16, True), (33, False), (64, "pinned")])
def test_ingest_wire_random_batches(oracle, S, device_inputs):
    rng = np.random.default_rng(1200 + S)
    G, n = 997, 30000
    state = make_state(rng, S, G)
    msgs, groups = wu.random_batch(rng, n, G, state[0], state[1], state[2])
    data, offsets = wu.pack(msgs)
    with raft_amd.Engine(G, S) as e:
        counts = run_both(oracle, e, state, data, offsets, groups, device_inputs)
        assert counts.sum() == n and (counts[:8] > 0).all()
        assert e.stats()["msgs_ingested"] == n


def test_ingest_wire_hand_vectors(oracle):
    """The hand-derived Unmarshal vectors of tests/test_wire_cpu.py, each as its own message
    and all in one batch: malformed ones are status 6, the rest decode to the stated fields."""
    S, G = 3, 2
    ids = np.array([[1, 5], [2, 6], [3, 7]], dtype=np.uint64)
    term = np.array([2, 2], dtype=np.uint64)
    state = (ids, term, [b"ab", b""], np.zeros((S, G), dtype=np.uint64), np.zeros(G, dtype=np.uint8),
             np.zeros(G, dtype=np.uint8))
    blobs = [b for b, _, _ in HAND]
    data, offsets = wu.pack(blobs)
    groups = np.zeros(len(blobs), dtype=np.uint32)
    with raft_amd.Engine(G, S) as e:
        run_both(oracle, e, state, data, offsets, groups)
        _, status, _ = e.ingest_wire(data, offsets, groups)
        for (b, rc, fields), st in zip(HAND, status):
            assert (st == capi.WIRE_MALFORMED) == (rc != OK), (b, rc, st)
        # one message per call: the byte window starts at arbitrary alignments
        for k, (b, rc, fields) in enumerate(HAND):
            msgs, status, _ = e.ingest_wire(data, offsets[k:k + 2], groups[:1])
            assert (status[0] == capi.WIRE_MALFORMED) == (rc != OK)
            if rc == OK and "index" in fields:
                assert int(msgs["index"][0]) == fields["index"]


def test_ingest_wire_long_messages_leave_the_lds_window(oracle):
    """A workgroup whose 256 messages exceed the staged window parses from global memory:
    long unknown fields and long contexts between ordinary responses."""
    rng = np.random.default_rng(1300)
    S, G, n = 5, 64, 3000
    state = make_state(rng, S, G)
    ids, term, ctxs = state[0], state[1], state[2]
    msgs, groups = wu.random_batch(rng, n, G, ids, term, ctxs, bad_frac=0.05)
    for i in range(0, n, 7):  # ~430 messages of 300..5000 bytes
        pad = bytes(rng.integers(0, 256, size=int(rng.integers(300, 5000)), dtype=np.uint8))
        if i % 2:
            msgs[i] = wu.tag(99, 2) + wu.varint(len(pad)) + pad + msgs[i]
        else:
            msgs[i] = msgs[i] + wu.tag(12, 2) + wu.varint(len(pad)) + pad  # a later, longer context wins
    data, offsets = wu.pack(msgs)
    assert int(offsets[256]) > 24576
    with raft_amd.Engine(G, S) as e:
        counts = run_both(oracle, e, state, data, offsets, groups)
        assert counts[capi.WIRE_APPLIED] > n // 2


def test_neighbouring_lanes_on_different_paths_of_the_parser(oracle):
    """Lanes of one wavefront parse different messages in lockstep: every pair of tails — unknown fields of every wire
    type behind one- and two-byte tags, varints of every length, a context, truncations — side by side in both orders
    (a form of the varint loop once decoded a one-byte tag wrongly when its neighbour's tag took two bytes)."""
    rng = np.random.default_rng(1450)
    S, G = 3, 16
    state = make_state(rng, S, G)
    ids, term = state[0], state[1]
    g = int(np.flatnonzero((ids != 0).all(axis=0))[0]) if (ids != 0).all(axis=0).any() else 0
    frm = int(ids[0, g]) if ids[0, g] else 1
    base = wu.encode_message(type=wu.MSG_APP_RESP, frm=frm, term=int(term[g]), index=77)
    tails = [b"", wu.tag(15, 0) + wu.varint(1), wu.tag(15, 0) + wu.varint((1 << 63) + 5), wu.tag(31, 5) + bytes(4),
             wu.tag(31, 1) + bytes(8), wu.tag(99, 2) + wu.varint(3) + b"abc", wu.tag(200, 0) + wu.varint(300),
             wu.tag(17, 3) + wu.tag(18, 0) + wu.varint(9) + wu.tag(17, 4), wu.tag(12, 2) + wu.varint(2) + b"hi",
             wu.tag(15, 0) + b"\x80", wu.tag(31, 5) + bytes(2), wu.tag(16, 4), wu.tag(3000, 0) + wu.varint(1 << 40),
             wu.tag(15, 0) + b"\x80" * 10 + b"\x01"]
    blobs, groups = [], []
    for a in tails:
        for b in tails:
            blobs += [base + a, base + b]
    data, offsets = wu.pack(blobs)
    groups = np.full(len(blobs), g, dtype=np.uint32)
    with raft_amd.Engine(G, S) as e:
        counts = run_both(oracle, e, state, data, offsets, groups)
        assert counts[capi.WIRE_APPLIED] > 0 and counts[capi.WIRE_MALFORMED] > 0


def test_ingest_wire_bad_offsets_and_empty(oracle):
    rng = np.random.default_rng(1400)
    S, G = 4, 8
    state = make_state(rng, S, G)
    ids, term = state[0], state[1]
    g = 3
    frm = int(ids[ids[:, g] != 0, g][0]) if (ids[:, g] != 0).any() else 1
    b = wu.encode_message(type=wu.MSG_APP_RESP, frm=frm, term=int(term[g]), index=1 << 41)
    data, _ = wu.pack([b, b, b, b])
    L = len(b)
    offsets = np.array([0, L, 9 * L, 3 * L, 4 * L], dtype=np.uint64)  # past the data, then backwards
    groups = np.full(4, g, dtype=np.uint32)
    with raft_amd.Engine(G, S) as e:
        run_both(oracle, e, state, data, offsets, groups)
        msgs, status, counts = e.ingest_wire(data, offsets, groups, n=0)
        assert counts.sum() == 0
        # empty messages (zero-length ranges) decode to the zero Message: type 0 = MsgHup, left to the host
        z = np.zeros(5, dtype=np.uint64)
        msgs, status, counts = e.ingest_wire(np.zeros(0, dtype=np.uint8), z, groups)
        assert (status == capi.WIRE_OTHER_TYPE).all()


def test_ingest_wire_requires_the_id_table():
    with raft_amd.Engine(4, 3) as e:
        with pytest.raises(raft_amd.RafthipError):
            e.ingest_wire(np.zeros(1, dtype=np.uint8), np.array([0, 1], dtype=np.uint64), np.zeros(1, dtype=np.uint32))


def test_ingest_wire_then_commit_full_size(oracle):
    """cfg5 shape through the serialized path: 2^20 groups x (7 voters + 2 learners), 4M
    serialized MsgAppResp / MsgHeartbeatResp, then the commit pass; every plane vs the oracle."""
    G, S, V = 1 << 20, 9, 7
    n = 4 * G
    rng = np.random.default_rng(0x5EED0007)
    ids = np.ascontiguousarray((np.arange(S, dtype=np.uint64)[:, None] + np.uint64(1)) * np.uint64(0x10001)
                               + rng.integers(0, 1 << 16, size=G, dtype=np.uint64)[None, :] * np.uint64(1 << 20))
    term = rng.integers(1, 1 << 20, size=G).astype(np.uint64)
    match = (gu.dist_b(0x5EED0007, S, G) & np.uint64((1 << 40) - 1)) + np.uint64(64)
    cw = rng.integers(1, 1 << 63, size=G, dtype=np.uint64)
    cl = np.full(G, 8, dtype=np.uint8)
    groups = rng.integers(0, G, size=n).astype(np.uint32)
    slots = rng.integers(0, S, size=n)
    data, offsets = wu.encode_responses(rng, ids[slots, groups], term[groups],
                                        match[slots, groups] + rng.integers(0, 64, size=n).astype(np.uint64),
                                        cw[groups])
    active = np.ones(G, dtype=np.uint16)
    acks = np.zeros(G, dtype=np.uint16)
    wm, wa, wk = match.copy(), active.copy(), acks.copy()
    _, want_status, want_counts = oracle.step_wire_msgs(wm, wa, wk, ids, term, cw, cl, data, offsets, groups)
    with raft_amd.Engine(G, S) as e:
        e.set_uniform_config(V)
        e.upload_match(match)
        e.upload_active(active)
        e.upload_acks(acks)
        e.upload_ids(ids)
        e.upload_term(term)
        e.upload_read_ctx(cw, cl)
        _, status, counts = e.ingest_wire(data, offsets, groups, want_msgs=False)
        assert np.array_equal(counts, want_counts) and np.array_equal(status, want_status)
        assert counts[capi.WIRE_APPLIED] > 0.8 * n
        e.commit()
        assert np.array_equal(e.read_match(), wm)
        assert np.array_equal(e.read_active(), wa)
        assert np.array_equal(e.read_acks(), wk)
        assert np.array_equal(e.read_commit(), oracle.batch_committed(wm, uniform_voters=V, threads=0))
        assert (wm != match).any() and wk.any()


@pytest.mark.parametrize("form", [capi.FLAG_STEP_LINKED, capi.FLAG_STEP_PARTITION], ids=["linked", "partitioned"])
@pytest.mark.parametrize("S,cap", [(3, 4), (9, 2), (16, 8), (40, 3)])
def test_step_wire_feeds_the_progress_stepper(oracle, S, cap, form):
    """rafthip_step_wire = the wire decode of rafthip_ingest_wire + rafthip_step_progress on the decoded
    records: rejections probe at RejectHint when LogTerm is 0 and go to the host (status 8, later events
    of that Progress deferred) when it is not."""
    import progress_util as pu
    from oracle import pyoracle as po
    rng = np.random.default_rng(1700 + S)
    G, n = 499, 20000
    ids, term, ctxs, _, _, _ = make_state(rng, S, G)
    pl = pu.random_planes(rng, S, G, cap, 0)
    blobs, groups = wu.random_batch(rng, n, G, ids, term, ctxs)
    # rejections: some with LogTerm 0 (an old follower, raft.go:1343-1345), aimed at Next-1 so that they bite
    for i in range(0, n, 9):
        g = int(groups[i]) % G
        s = int(rng.integers(0, S))
        if ids[s, g] == 0:
            continue
        groups[i] = g
        blobs[i] = wu.encode_message(type=wu.MSG_APP_RESP, frm=int(ids[s, g]), term=int(term[g]),
                                     index=int(pl["next"][s, g]) - 1, reject=True,
                                     rejectHint=int(pl["match"][s, g]) + int(rng.integers(0, 3)),
                                     logTerm=0 if rng.random() < 0.6 else 3)
    data, offsets = wu.pack(blobs)
    cw, cl = wu.ctx_words(ctxs)
    last = pl["match"].max(axis=0) + np.uint64(5)
    first = np.zeros(G, dtype=np.uint64)
    active, acks = pu.mask_planes(S, G)
    want_msgs, want_aux, want_status, want_counts = oracle.decode_wire_msgs(ids, term, cw, cl, data, offsets, groups)
    assert want_counts[capi.WIRE_NEEDS_LOG] > 50 and (want_msgs["type"][want_status == 8] == capi.MSG_NEEDS_HOST).all()
    with raft_amd.Engine(G, S, flags=form) as e:
        e.progress_config(cap, 0)
        e.upload_match(pl["match"])
        e.upload_progress(pl["pstate"], pl["next"], pl["pending"])
        e.upload_inflights(pl["infl_meta"], pl["infl_bytes"], pl["ring_index"], pl["ring_bytes"])
        e.upload_active(active)
        e.upload_acks(acks)
        e.upload_term_range(np.zeros(G, dtype=np.uint64), last)
        e.upload_first_index(first)
        e.upload_ids(ids)
        e.upload_term(term)
        e.upload_read_ctx(cw, cl)
        msgs, status, counts, act = e.step_wire(data, offsets, groups)
        assert np.array_equal(status, want_status) and np.array_equal(counts, want_counts)
        assert np.array_equal(msgs, want_msgs)
        want_act = po.step_progress(pl, cap, 0, active, acks, last, first, want_msgs, want_aux)
        assert np.array_equal(act, want_act)
        got = e.read_progress()
        for k in ("pstate", "next", "pending", "infl_meta", "infl_bytes", "ring_index", "ring_bytes"):
            assert np.array_equal(got[k], pl[k]), k
        assert np.array_equal(e.read_match(), pl["match"])
        assert np.array_equal(e.read_active(), active) and np.array_equal(e.read_acks(), acks)
        # every needs-log message is deferred, and so is everything after it on its Progress
        nl = np.flatnonzero(want_status == 8)
        assert (want_act[nl] == capi.ACT_DEFERRED).all()
        applied_rejects = (want_status == 0) & (want_msgs["reject"] != 0) & (want_msgs["type"] == 4)
        assert (want_act[applied_rejects] & capi.ACT_SEND_APPEND).any()
        # the host steps what came back: hint from its log (here: RejectHint itself), then the deferred rest
        idx = np.flatnonzero(want_act == capi.ACT_DEFERRED)
        redo = want_msgs[idx].copy()
        redo["type"][redo["type"] == capi.MSG_NEEDS_HOST] = capi.MSG_APP_RESP
        a2 = e.step_progress(redo, np.ascontiguousarray(want_aux[idx]))
        w2 = po.step_progress(pl, cap, 0, active, acks, last, first, redo, np.ascontiguousarray(want_aux[idx]))
        assert np.array_equal(a2, w2) and (w2 & capi.ACT_DEFERRED == 0).all()
        assert np.array_equal(e.read_progress()["next"], pl["next"])


def test_small_batches_from_host_memory_either_side_of_the_arena(oracle):
    """Batches of up to 131072 messages from pageable host memory go through the engine's page-locked arena
    (rafthip.hip small_take) instead of DMA copies, piece by piece as each fits: the same answers either side of the
    message limit, with a payload larger than the arena, and call after call on one engine."""
    import progress_util as pu
    from oracle import pyoracle as po
    rng = np.random.default_rng(1900)
    S, G, cap = 5, 3001, 4
    ids, term, ctxs, _, _, _ = make_state(rng, S, G)
    cw, cl = wu.ctx_words(ctxs)
    pl = pu.random_planes(rng, S, G, cap, 0)
    last = pl["match"].max(axis=0) + np.uint64(1 << 20)
    first = np.zeros(G, dtype=np.uint64)
    active, acks = pu.mask_planes(S, G)
    with raft_amd.Engine(G, S) as e:
        e.progress_config(cap, 0)
        e.upload_match(pl["match"])
        e.upload_progress(pl["pstate"], pl["next"], pl["pending"])
        e.upload_inflights(pl["infl_meta"], pl["infl_bytes"], pl["ring_index"], pl["ring_bytes"])
        e.upload_active(active)
        e.upload_acks(acks)
        e.upload_term_range(np.zeros(G, dtype=np.uint64), last)
        e.upload_first_index(first)
        e.upload_ids(ids)
        e.upload_term(term)
        e.upload_read_ctx(cw, cl)
        for n, fat in [(131073, 0), (131072, 0), (1, 0), (100, 0), (3600, 2), (100, 0), (9000, 0), (4500, 3)]:
            blobs, groups = wu.random_batch(rng, n, G, ids, term, ctxs, bad_frac=0.1)
            for i in range(0, n, fat or n + 1):  # fat > 0: unknown fields that push the payload past the arena's 16 MiB
                if fat:
                    pad = bytes(rng.integers(0, 256, size=int(rng.integers(9000, 15000)), dtype=np.uint8))
                    blobs[i] = wu.tag(99, 2) + wu.varint(len(pad)) + pad + blobs[i]
            data, offsets = wu.pack(blobs)
            assert (len(data) > (16 << 20)) == bool(fat)
            want_msgs, want_aux, want_status, want_counts = oracle.decode_wire_msgs(ids, term, cw, cl, data, offsets,
                                                                                    groups)
            msgs, status, counts, act = e.step_wire(data, offsets, groups)
            assert np.array_equal(status, want_status) and np.array_equal(counts, want_counts)
            assert np.array_equal(msgs, want_msgs)
            want_act = po.step_progress(pl, cap, 0, active, acks, last, first, want_msgs, want_aux)
            assert np.array_equal(act, want_act)
            assert np.array_equal(e.read_match(), pl["match"])
            assert np.array_equal(e.read_progress()["next"], pl["next"])
            # the decode alone (messages through the arena as well), on a copy of the state
            wm, wa, wk = pl["match"].copy(), active.copy(), acks.copy()
            m2, s2, c2 = oracle.step_wire_msgs(wm, wa, wk, ids, term, cw, cl, data, offsets, groups)
            msgs, status, counts = e.ingest_wire(data, offsets, groups)
            assert np.array_equal(msgs, m2) and np.array_equal(status, s2) and np.array_equal(counts, c2)
            assert np.array_equal(e.read_match(), wm)
            pl["match"][:] = wm
            active[:], acks[:] = wa, wk
            assert np.array_equal(e.read_active(), active) and np.array_equal(e.read_acks(), acks)


class EngineVoteDriver:
    """tests/golden/scenarios.json's wire_vote_sequences on the engine: the scenario's group sits among G others."""

    def __init__(self, e, voters, G, g):
        self.e, self.voters, self.G, self.g = e, voters, G, g
        S = len(voters)
        ids = np.zeros((S, G), dtype=np.uint64)
        ids[:, g] = voters
        e.set_uniform_config(S)
        e.upload_ids(ids)
        e.upload_term(np.zeros(G, dtype=np.uint64))
        e.upload_read_ctx(np.zeros(G, dtype=np.uint64), np.zeros(G, dtype=np.uint8))
        e.upload_campaign(np.zeros(G, dtype=np.uint8))

    def become(self, resp_type, term):
        self.e.upload_campaign(np.array([resp_type], dtype=np.uint8), first=self.g, count=1)
        self.e.upload_term(np.array([term], dtype=np.uint64), first=self.g, count=1)
        self.e.reset_votes(first=self.g, count=1)

    def record(self, pairs):
        m = np.zeros(len(pairs), dtype=capi.MSG_DTYPE)
        m["group"], m["type"] = self.g, capi.MSG_VOTE_RESP
        m["slot"] = [self.voters.index(i) for i, _ in pairs]
        m["reject"] = [0 if v else 1 for _, v in pairs]
        self.e.record_votes(m)

    def wire(self, blob):
        data, offsets = wu.pack([blob])
        msgs, status, counts = self.e.ingest_wire(data, offsets, np.array([self.g], dtype=np.uint32))
        assert counts[status[0]] == 1 and counts.sum() == 1
        assert (msgs["slot"][0] != 0xFF) == (status[0] in (capi.WIRE_APPLIED, capi.WIRE_VOTE))
        return int(status[0])

    def tally(self):
        self.e.tally_votes()
        res, gr, rj = self.e.read_vote_result(first=self.g, count=1)
        return int(res[0]), int(gr[0]), int(rj[0])


def test_vote_responses_from_the_reference_transcripts():
    from test_wire_cpu import _vote_sequences
    for seq in _vote_sequences():
        with raft_amd.Engine(70, len(seq["voters"])) as e:
            drv = EngineVoteDriver(e, seq["voters"], 70, 41)
            wu.replay_vote_sequence(seq, drv)
            votes = e.read_votes()
            votes[:, 41] = 0
            assert not votes.any()  # no other group's cells moved


def test_figure7_election_from_the_wire():
    import figure7_util as f7
    fx = f7.load()
    with raft_amd.Engine(9, len(fx["voters"])) as e:
        drv = EngineVoteDriver(e, fx["voters"], 9, 4)
        drv.become(wu.MSG_VOTE_RESP, fx["term"])
        for v in fx["votes"]:
            if v["from"] == fx["leader"]:
                drv.record([(v["from"], v["granted"])])
            else:
                blob = wu.encode_message(type=wu.MSG_VOTE_RESP, to=fx["leader"], frm=v["from"], term=fx["term"],
                                         reject=not v["granted"])
                assert drv.wire(blob) == capi.WIRE_VOTE
            assert drv.tally() == (wu.VOTE_RESULT[v["result"]], v["granted_total"], v["rejected_total"]), v


@pytest.mark.parametrize("S,to_progress", [(3, False), (5, True), (9, False), (16, True), (33, False), (64, True)])
def test_vote_wire_random_batches(oracle, S, to_progress):
    """Candidates, pre-candidates, leaders and followers side by side: vote responses and MsgAppResp in one batch,
    through rafthip_ingest_wire and through rafthip_step_wire; status, records, counts, vote cells, tallies and the
    leaders' planes against the oracle stepping the same bytes one message at a time."""
    import progress_util as pu
    from oracle import pyoracle as po
    rng = np.random.default_rng(2100 + S)
    G, n = 613, 25000
    ids, term, ctxs, match, active, acks = make_state(rng, S, G)
    term += np.uint64(1)
    campaign = rng.choice(np.array([0, 6, 18], dtype=np.uint8), size=G, p=[0.3, 0.35, 0.35])
    votes = rng.choice(np.array([0, 1, 2], dtype=np.uint8), size=(S, G), p=[0.8, 0.1, 0.1])
    blobs, groups = wu.random_vote_batch(rng, n, G, ids, term)
    data, offsets = wu.pack(blobs)
    cw, cl = wu.ctx_words(ctxs)
    in_mask = np.full(G, (1 << min(S, 7)) - 1, dtype=gu.mask_dtype(S))
    with raft_amd.Engine(G, S) as e:
        e.upload_config(in_mask, np.zeros_like(in_mask))
        e.upload_match(match)
        e.upload_active(active)
        e.upload_acks(acks)
        e.upload_ids(ids)
        e.upload_term(term)
        e.upload_read_ctx(cw, cl)
        e.upload_votes(votes)
        e.upload_campaign(campaign)
        wv = votes.copy()
        if to_progress:
            pl = pu.random_planes(rng, S, G, 4, 0)
            pl["match"] = match.copy()
            pl["next"] = np.maximum(pl["next"], match + np.uint64(1))
            last = match.max(axis=0) + np.uint64(1 << 31)
            first = np.zeros(G, dtype=np.uint64)
            e.progress_config(4, 0)
            e.upload_progress(pl["pstate"], pl["next"], pl["pending"])
            e.upload_inflights(pl["infl_meta"], pl["infl_bytes"], pl["ring_index"], pl["ring_bytes"])
            e.upload_term_range(np.zeros(G, dtype=np.uint64), last)
            e.upload_first_index(first)
            want_msgs, want_aux, want_status, want_counts = oracle.decode_wire_msgs(ids, term, cw, cl, data, offsets, groups,
                                                                                    campaign=campaign, votes=wv)
            msgs, status, counts, act = e.step_wire(data, offsets, groups)
            ev = want_msgs.copy()
            ev["slot"][want_status == capi.WIRE_VOTE] = 0xFF  # a vote response is no event of a Progress
            want_act = po.step_progress(pl, 4, 0, active, acks, last, first, ev, want_aux)
            assert np.array_equal(act, want_act)
            assert (act[want_status == capi.WIRE_VOTE] == capi.ACT_DROPPED).all()
            assert np.array_equal(e.read_progress()["next"], pl["next"]) and np.array_equal(e.read_match(), pl["match"])
        else:
            wm, wa, wk = match.copy(), active.copy(), acks.copy()
            want_msgs, want_status, want_counts = oracle.step_wire_msgs(wm, wa, wk, ids, term, cw, cl, data, offsets, groups,
                                                                        campaign=campaign, votes=wv)
            msgs, status, counts = e.ingest_wire(data, offsets, groups)
            assert np.array_equal(e.read_match(), wm) and np.array_equal(e.read_active(), wa)
        assert np.array_equal(status, want_status) and np.array_equal(counts, want_counts)
        assert np.array_equal(msgs, want_msgs)
        assert counts[capi.WIRE_VOTE] > n // 10 and counts[capi.WIRE_HIGHER_TERM] > 0 and counts[capi.WIRE_OTHER_TYPE] > 0
        assert np.array_equal(e.read_votes(), wv) and (wv != votes).any()
        e.tally_votes()
        res, gr, rj = e.read_vote_result()
        wres, wgr, wrj = oracle.batch_tally_votes(wv, in_mask, np.zeros_like(in_mask))
        assert np.array_equal(res, wres) and np.array_equal(gr, wgr) and np.array_equal(rj, wrj)
        # clearing the campaign plane turns the vote passes off: the same bytes now leave the cells alone
        e.upload_campaign(None)
        _, status2, counts2 = e.ingest_wire(data, offsets, groups)
        assert counts2[capi.WIRE_VOTE] == 0 and np.array_equal(e.read_votes(), wv)


@pytest.mark.parametrize("S,G", [(3, 1), (5, 1000), (9, 4099), (16, 777), (33, 300), (64, 130)])
def test_emit_heartbeats(oracle, S, G):
    """rafthip_emit_heartbeats = bcastHeartbeat -> Message.Marshal, byte for byte against the oracle: ids up to
    2^64-1, empty slots, groups not led here, ReadIndex contexts of every length, a partial range, device outputs."""
    rng = np.random.default_rng(2300 + S)
    ids, term, ctxs, match, _, _ = make_state(rng, S, G)
    term[rng.random(G) < 0.1] = np.uint64((1 << 64) - 1)
    mdt = gu.mask_dtype(S)
    self_slot = rng.integers(0, S, size=G)
    self_mask = (np.uint64(1) << self_slot.astype(np.uint64)).astype(mdt)
    self_mask[rng.random(G) < 0.15] = 0
    committed = match[rng.integers(0, S, size=G), np.arange(G)]
    cw, cl = wu.ctx_words(ctxs)
    want, want_len = oracle.bcast_heartbeat(match, committed, ids, term, cw, cl, self_mask)
    assert want_len.max() <= capi.HEARTBEAT_STRIDE and (G < 10 or ((want_len > 0).any() and (want_len == 0).any()))
    with raft_amd.Engine(G, S) as e:
        e.upload_match(match)
        e.upload_commit(committed)
        e.upload_ids(ids)
        e.upload_term(term)
        e.upload_read_ctx(cw, cl)
        e.upload_self(self_mask)
        got, got_len = e.emit_heartbeats()
        assert np.array_equal(got_len, want_len)
        assert np.array_equal(got, want)
        if G > 10:
            a, n = G // 3, G // 2
            got, got_len = e.emit_heartbeats(first=a, count=n)
            assert np.array_equal(got_len, want_len[:, a:a + n]) and np.array_equal(got, want[:, a:a + n])
            import torch
            db = torch.full((S, G, capi.HEARTBEAT_STRIDE), 0xEE, dtype=torch.uint8, device="cuda")
            dl = torch.full((S, G), 0xEE, dtype=torch.uint8, device="cuda")
            L = raft_amd.load_library()
            assert L.rafthip_emit_heartbeats(e._h, 0, G, db.data_ptr(), dl.data_ptr()) == 0
            e.synchronize()
            assert np.array_equal(db.cpu().numpy(), want) and np.array_equal(dl.cpu().numpy(), want_len)
            # rafthip_emit_heartbeats_due: HeartbeatTick 3, every group a leader somewhere in its heartbeat interval —
            # only the groups whose tick stepped MsgBeat (raft.go:856-861) send, the rest read as groups not led here
            with pytest.raises(raft_amd.RafthipError):
                e.emit_heartbeats(due=True)  # no tick yet
            e.tick_config(10, 3, False)
            hb = rng.integers(0, 3, size=G).astype(np.uint32)
            e.upload_tick_state(np.zeros(G, np.uint32), hb, np.full(G, 10, np.uint32), np.full(G, 2, np.uint8), np.ones(G, np.uint8))
            for k in range(3):
                e.tick(counts=False)
                beat = (hb + k + 1) % 3 == 0
                assert np.array_equal((e.read_tick_actions() & capi.TICK_BEAT) != 0, beat)
                got, got_len = e.emit_heartbeats(due=True)
                assert np.array_equal(got_len, want_len * beat[None, :]) and np.array_equal(got, want * beat[None, :, None])
                assert beat.any() and not beat.all()


def test_emit_heartbeats_on_the_reference_cases():
    """The heartbeats the reference's tests show (heartbeat_commit_cases), emitted by the engine and parsed by the
    protobuf runtime: `1->2 MsgHeartbeat Term:1 Log:0/0 Commit:17` and friends."""
    import json
    import os
    with open(os.path.join(gu.HERE, "golden", "scenarios.json")) as fh:
        cases = json.load(fh)["heartbeat_commit_cases"]
    Msg = wu.pb_message_class()
    for c in cases:
        n = len(c["ids"])
        me = c["ids"].index(c["leader"])
        with raft_amd.Engine(1, n) as e:
            e.upload_match(np.array(c["match"], dtype=np.uint64).reshape(n, 1))
            e.upload_commit(np.array([c["committed"]], dtype=np.uint64))
            e.upload_ids(np.array(c["ids"], dtype=np.uint64).reshape(n, 1))
            e.upload_term(np.array([c["term"]], dtype=np.uint64))
            e.upload_self(np.array([1 << me], dtype=np.uint8))
            out, lens = e.emit_heartbeats()
            assert lens[me, 0] == 0
            for peer, want in c["want"].items():
                s = c["ids"].index(int(peer))
                m = Msg()
                m.ParseFromString(out[s, 0, :lens[s, 0]].tobytes())
                assert (m.type, m.to, getattr(m, "from"), m.term, m.commit) == (8, int(peer), c["leader"], c["term"], want), c["src"]
                assert not m.HasField("context")


def test_read_ctx_longer_than_a_row_holds_is_refused_or_dropped(oracle):
    """A row holds 8 context bytes (include/rafthip.h).  Host input with ctx_len > 8: RAFTHIP_EINVAL, nothing stored.
    Device-resident input: stored as "none pending" — no context in the heartbeats, no ReadIndex ack from a response
    that shares the first 8 bytes."""
    import torch
    S, G = 3, 64
    rng = np.random.default_rng(77)
    ids = np.ascontiguousarray(np.tile(np.arange(1, S + 1, dtype=np.uint64)[:, None], (1, G)))
    term = np.full(G, 2, dtype=np.uint64)
    ctx = rng.integers(1, 1 << 62, size=G, dtype=np.uint64)
    good = np.full(G, 8, dtype=np.uint8)
    bad = good.copy()
    bad[5] = 9
    self_mask = np.ones(G, dtype=np.uint8)
    match = rng.integers(0, 1 << 40, size=(S, G), dtype=np.uint64)
    with raft_amd.Engine(G, S) as e:
        e.upload_match(match)
        e.upload_ids(ids)
        e.upload_term(term)
        e.upload_self(self_mask)
        e.upload_read_ctx(ctx, good)
        with pytest.raises(capi.RafthipError, match="more than 8 bytes"):
            e.upload_read_ctx(ctx, bad)
        hb, lens = e.emit_heartbeats()
        want, want_len = oracle.bcast_heartbeat(match, e.read_commit(), ids, term, ctx, good, self_mask)
        assert np.array_equal(lens, want_len) and np.array_equal(hb, want)  # the refused upload changed nothing
        d_ctx = torch.from_numpy(ctx.view(np.int64)).to("cuda")
        d_len = torch.from_numpy(bad).to("cuda")
        e.upload_read_ctx(d_ctx.data_ptr(), d_len.data_ptr())
        hb, lens = e.emit_heartbeats()
        none5 = good.copy()
        none5[5] = 0
        ctx5 = ctx.copy()
        ctx5[5] = 0
        want, want_len = oracle.bcast_heartbeat(match, e.read_commit(), ids, term, ctx5, none5, self_mask)
        assert np.array_equal(lens, want_len) and np.array_equal(hb, want)
        # a heartbeat response carrying a 9-byte context whose first 8 bytes equal the refused one: no ack
        e.upload_acks(np.zeros(G, dtype=np.uint8))
        blob = oracle.message_marshal(type=9, to=1, frm=2, term=2, context=int(ctx[5]).to_bytes(8, "little") + b"\x01")
        data = np.frombuffer(blob, dtype=np.uint8).copy()
        offsets = np.array([0, len(blob)], dtype=np.uint64)
        _, status, _ = e.ingest_wire(data, offsets, np.array([5], dtype=np.uint32))
        assert status[0] == capi.WIRE_APPLIED and e.read_acks()[5] == 0
