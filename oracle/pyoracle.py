"""ctypes loader for oracle/liboracle.so (built by oracle/Makefile).

TEST INFRASTRUCTURE: the checker for the HIP engine, never the thing shipped.
Each wrapper names the reference function it restates; the C file carries the
file:line citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VOTE_PENDING, VOTE_LOST, VOTE_WON = 1, 2, 3
VOTE_NAMES = {1: "VotePending", 2: "VoteLost", 3: "VoteWon"}

ACK_DTYPE = np.dtype([("group", "<u4"), ("slot", "<u4"), ("index", "<u8")])


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("quorum_oracle.c", "progress_oracle.c")]
    if force or not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(src) for src in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        u64p, vp, u8p = C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint8)
        L.oracle_majority_committed_index.restype = C.c_uint64
        L.oracle_majority_committed_index.argtypes = [u64p, C.c_int, u64p, u64p, C.c_int]
        L.oracle_alt_majority_committed_index.restype = C.c_uint64
        L.oracle_alt_majority_committed_index.argtypes = [u64p, C.c_int, u64p, u64p, C.c_int]
        L.oracle_joint_committed_index.restype = C.c_uint64
        L.oracle_joint_committed_index.argtypes = [u64p, C.c_int, u64p, C.c_int, u64p, u64p, C.c_int]
        L.oracle_majority_vote_result.restype = C.c_int
        L.oracle_majority_vote_result.argtypes = [u64p, C.c_int, u64p, u64p, C.c_int]
        L.oracle_joint_vote_result.restype = C.c_int
        L.oracle_joint_vote_result.argtypes = [u64p, C.c_int, u64p, C.c_int, u64p, u64p, C.c_int]
        L.oracle_maybe_update.restype = C.c_int
        L.oracle_maybe_update.argtypes = [u64p, u64p, C.c_uint64]
        L.oracle_log_maybe_commit.restype = C.c_uint64
        L.oracle_log_maybe_commit.argtypes = [C.c_uint64] * 4
        L.oracle_batch_committed.restype = None
        L.oracle_batch_committed.argtypes = [C.c_uint64, C.c_int, vp, C.c_uint64, vp, vp, C.c_int,
                                             C.c_int, vp, C.c_int]
        L.oracle_batch_tally_votes.restype = None
        L.oracle_batch_tally_votes.argtypes = [C.c_uint64, C.c_int, vp, C.c_uint64, vp, vp, vp,
                                               C.c_int, C.c_int, vp, vp, vp]
        L.oracle_batch_quorum_active.restype = None
        L.oracle_batch_quorum_active.argtypes = [C.c_uint64, C.c_int, vp, vp, vp, vp, C.c_int,
                                                 C.c_int, vp]
        L.oracle_batch_ack_quorum.restype = None
        L.oracle_batch_ack_quorum.argtypes = [C.c_uint64, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp]
        L.oracle_scatter_acks.restype = C.c_uint64
        L.oracle_scatter_acks.argtypes = [C.c_uint64, C.c_int, vp, C.c_uint64, vp, C.c_uint64]
        L.oracle_raftlog_maybe_commit.restype = C.c_int
        L.oracle_raftlog_maybe_commit.argtypes = [C.c_uint64, C.c_uint64, vp, C.c_uint64, u64p, C.c_uint64,
                                                  C.c_uint64]
        L.oracle_committed_entry_in_current_term.restype = C.c_int
        L.oracle_committed_entry_in_current_term.argtypes = [C.c_uint64, C.c_uint64, vp, C.c_uint64, C.c_uint64,
                                                             C.c_uint64]
        L.oracle_batch_maybe_commit.restype = None
        L.oracle_batch_maybe_commit.argtypes = [C.c_uint64, C.c_int, vp, C.c_uint64, vp, vp, C.c_int, C.c_int,
                                                vp, vp, vp, vp, vp, vp, vp]
        L.oracle_step_leader_msgs.restype = None
        L.oracle_step_leader_msgs.argtypes = [C.c_uint64, C.c_int, vp, C.c_uint64, vp, vp, C.c_int, vp, vp,
                                              C.c_uint64, vp]
        L.oracle_message_unmarshal.restype = C.c_int
        L.oracle_message_unmarshal.argtypes = [vp, C.c_uint64, vp]
        L.oracle_step_wire_msgs.restype = None
        L.oracle_step_wire_msgs.argtypes = [C.c_uint64, C.c_int, vp, C.c_uint64, vp, vp, C.c_int, vp, C.c_uint64,
                                            vp, vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp, vp,
                                            vp, vp, C.c_uint64]
        L.oracle_progress_new.restype = vp
        L.oracle_progress_new.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint64]
        L.oracle_progress_free.restype = None
        L.oracle_progress_free.argtypes = [vp]
        L.oracle_progress_get.restype = None
        L.oracle_progress_get.argtypes = [vp, vp]
        L.oracle_inflights_get.restype = None
        L.oracle_inflights_get.argtypes = [vp, vp, vp, vp]
        L.oracle_inflights_set_start.restype = None
        L.oracle_inflights_set_start.argtypes = [vp, C.c_int]
        L.oracle_inflights_add.restype = C.c_int
        L.oracle_inflights_add.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.oracle_inflights_free_le.restype = None
        L.oracle_inflights_free_le.argtypes = [vp, C.c_uint64]
        L.oracle_inflights_reset.restype = None
        L.oracle_inflights_reset.argtypes = [vp]
        L.oracle_progress_maybe_update.restype = C.c_int
        L.oracle_progress_maybe_update.argtypes = [vp, C.c_uint64]
        L.oracle_progress_maybe_decr_to.restype = C.c_int
        L.oracle_progress_maybe_decr_to.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.oracle_progress_become.restype = None
        L.oracle_progress_become.argtypes = [vp, C.c_int, C.c_uint64]
        L.oracle_step_progress.restype = None
        L.oracle_step_progress.argtypes = [C.c_uint64, C.c_int, C.c_uint64] + [vp] * 8 + [C.c_int, C.c_uint64, vp, vp,
                                           C.c_int, vp, vp, vp, vp, C.c_uint64, C.c_uint32, vp]
        L.oracle_record_votes.restype = C.c_uint64
        L.oracle_record_votes.argtypes = [C.c_uint64, C.c_int, vp, C.c_uint64, vp, C.c_uint64]
        L.oracle_decode_wire_msgs.restype = None
        L.oracle_decode_wire_msgs.argtypes = [C.c_uint64, C.c_int, vp, C.c_uint64, vp, vp, vp, vp, C.c_uint64, vp, vp,
                                              C.c_uint64, vp, vp, vp, vp, vp, vp, C.c_uint64]
        L.oracle_batch_tick.restype = None
        L.oracle_batch_tick.argtypes = [C.c_uint64, C.c_int, vp, vp, vp, vp, vp, C.c_uint32, C.c_uint32, C.c_int,
                                        vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]
        L.oracle_batch_check_quorum.restype = None
        L.oracle_batch_check_quorum.argtypes = [C.c_uint64, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]
        L.oracle_batch_heartbeat_commit.restype = None
        L.oracle_batch_heartbeat_commit.argtypes = [C.c_uint64, C.c_int, vp, C.c_uint64, vp, vp, C.c_uint64]
        L.oracle_bench_create.restype = C.c_void_p
        L.oracle_bench_create.argtypes = [C.c_uint64, C.c_int, vp, C.c_uint64]
        L.oracle_bench_run.restype = C.c_int
        L.oracle_bench_run.argtypes = [C.c_void_p, vp, C.c_int]
        L.oracle_selfcheck_committed.restype = C.c_uint64
        L.oracle_selfcheck_committed.argtypes = [C.c_uint64, C.c_uint64]
        L.oracle_bench_leaf_ns.restype = C.c_double
        L.oracle_bench_leaf_ns.argtypes = [C.c_int, C.c_uint64, C.c_uint64, u64p]
        L.oracle_bench_soa5.restype = C.c_int
        L.oracle_bench_soa5.argtypes = [C.c_uint64, vp, C.c_uint64, vp, C.c_int]
        L.oracle_bench_destroy.restype = None
        L.oracle_bench_destroy.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _arr(xs):
    a = np.ascontiguousarray(np.asarray(list(xs), dtype=np.uint64))
    return a, a.ctypes.data_as(C.POINTER(C.c_uint64)), len(a)


def _map(m):
    ids, pi, n = _arr(m.keys())
    vals, pv, _ = _arr(int(v) for v in m.values())
    return (ids, vals), pi, pv, n


# ---- per-group, id-level (the reference's own shapes) ----------------------

def majority_committed_index(cfg, l):
    """MajorityConfig(cfg).CommittedIndex(mapAckIndexer(l))."""
    c, pc, n = _arr(cfg)
    keep, pi, pv, ln = _map(l)
    return int(lib().oracle_majority_committed_index(pc, n, pi, pv, ln))


def alt_majority_committed_index(cfg, l):
    """alternativeMajorityCommittedIndex (quorum/quick_test.go)."""
    c, pc, n = _arr(cfg)
    keep, pi, pv, ln = _map(l)
    return int(lib().oracle_alt_majority_committed_index(pc, n, pi, pv, ln))


def joint_committed_index(cfg0, cfg1, l):
    """JointConfig{cfg0,cfg1}.CommittedIndex(mapAckIndexer(l))."""
    c0, p0, n0 = _arr(cfg0)
    c1, p1, n1 = _arr(cfg1)
    keep, pi, pv, ln = _map(l)
    return int(lib().oracle_joint_committed_index(p0, n0, p1, n1, pi, pv, ln))


def majority_vote_result(cfg, votes):
    """MajorityConfig(cfg).VoteResult(votes); votes: id -> bool."""
    c, pc, n = _arr(cfg)
    keep, pi, pv, vn = _map(votes)
    return int(lib().oracle_majority_vote_result(pc, n, pi, pv, vn))


def joint_vote_result(cfg0, cfg1, votes):
    """JointConfig{cfg0,cfg1}.VoteResult(votes)."""
    c0, p0, n0 = _arr(cfg0)
    c1, p1, n1 = _arr(cfg1)
    keep, pi, pv, vn = _map(votes)
    return int(lib().oracle_joint_vote_result(p0, n0, p1, n1, pi, pv, vn))


def maybe_update(match, nxt, n):
    """Progress{Match,Next}.MaybeUpdate(n) -> (updated, match, next)."""
    m, x = C.c_uint64(match), C.c_uint64(nxt)
    ok = lib().oracle_maybe_update(C.byref(m), C.byref(x), n)
    return bool(ok), m.value, x.value


def log_maybe_commit(committed, max_index, term_of_max, term):
    """raftLog.maybeCommit guard -> new committed."""
    return int(lib().oracle_log_maybe_commit(committed, max_index, term_of_max, term))


# ---- batched SoA views ------------------------------------------------------

def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _mask_bytes(*masks):
    for m in masks:
        if m is not None:
            return m.dtype.itemsize
    return 1


def _chk(a, dtype=None):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    if dtype is not None:
        assert a.dtype == dtype, (a.dtype, dtype)
    return a


def batch_committed(match, in_mask=None, out_mask=None, uniform_voters=0, threads=1):
    """ProgressTracker.Committed for every group. match: u64 [S][G]."""
    match = _chk(match, np.uint64)
    S, G = match.shape
    out = np.empty(G, dtype=np.uint64)
    lib().oracle_batch_committed(G, S, _vp(match), match.strides[0] // 8, _vp(_chk(in_mask)),
                                 _vp(_chk(out_mask)), _mask_bytes(in_mask, out_mask),
                                 uniform_voters, _vp(out), threads)
    return out


def batch_tally_votes(votes, in_mask=None, out_mask=None, learner_mask=None, uniform_voters=0):
    """ProgressTracker.TallyVotes for every group -> (result, granted, rejected)."""
    votes = _chk(votes, np.uint8)
    S, G = votes.shape
    res = np.empty(G, dtype=np.uint8)
    gr = np.empty(G, dtype=np.uint8)
    rj = np.empty(G, dtype=np.uint8)
    lib().oracle_batch_tally_votes(G, S, _vp(votes), votes.strides[0], _vp(_chk(in_mask)),
                                   _vp(_chk(out_mask)), _vp(_chk(learner_mask)),
                                   _mask_bytes(in_mask, out_mask, learner_mask), uniform_voters,
                                   _vp(res), _vp(gr), _vp(rj))
    return res, gr, rj


def batch_quorum_active(active, nslots, in_mask=None, out_mask=None, learner_mask=None,
                        uniform_voters=0):
    """ProgressTracker.QuorumActive for every group -> u8 bool."""
    active = _chk(active)
    G = active.shape[0]
    out = np.empty(G, dtype=np.uint8)
    lib().oracle_batch_quorum_active(G, nslots, _vp(active), _vp(_chk(in_mask)),
                                     _vp(_chk(out_mask)), _vp(_chk(learner_mask)),
                                     active.dtype.itemsize, uniform_voters, _vp(out))
    return out


def batch_ack_quorum(acks, nslots, in_mask=None, out_mask=None, uniform_voters=0):
    """Voters.VoteResult(readOnly acks) for every group -> VoteResult u8."""
    acks = _chk(acks)
    G = acks.shape[0]
    out = np.empty(G, dtype=np.uint8)
    lib().oracle_batch_ack_quorum(G, nslots, _vp(acks), _vp(_chk(in_mask)), _vp(_chk(out_mask)),
                                  acks.dtype.itemsize, uniform_voters, _vp(out))
    return out


def scatter_acks(match, recs):
    """Apply MsgAppResp acks in order (Progress.MaybeUpdate); mutates match. Returns #skipped."""
    match = _chk(match, np.uint64)
    recs = _chk(recs, ACK_DTYPE)
    S, G = match.shape
    return int(lib().oracle_scatter_acks(G, S, _vp(match), match.strides[0] // 8, _vp(recs),
                                         len(recs)))


MSG_DTYPE = np.dtype([("group", "<u4"), ("slot", "u1"), ("type", "u1"), ("reject", "u1"), ("flags", "u1"),
                      ("index", "<u8")])


def raftlog_maybe_commit(first_index, snap_term, terms, committed, max_index, term):
    """raftLog.maybeCommit(maxIndex, term) on an explicit log -> (advanced, committed)."""
    t = np.ascontiguousarray(np.asarray(terms, dtype=np.uint64))
    c = C.c_uint64(committed)
    ok = lib().oracle_raftlog_maybe_commit(first_index, snap_term, _vp(t), len(t), C.byref(c), max_index, term)
    return bool(ok), c.value


def raftlog_term(first_index, snap_term, terms, i):
    """raftLog.term(i) (log.go:381-407) -> (term, err) with err 0 ok / 1 ErrCompacted / 2 ErrUnavailable."""
    t = np.ascontiguousarray(np.asarray(terms, dtype=np.uint64))
    err = C.c_int()
    f = lib().oracle_raftlog_term
    f.restype = C.c_uint64
    f.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]
    term = f(first_index, snap_term, t.ctypes.data_as(C.POINTER(C.c_uint64)), len(t), i, C.byref(err))
    return int(term), err.value


def find_conflict_by_term(first_index, snap_term, terms, index, term):
    """raftLog.findConflictByTerm(index, term) on an explicit log -> (guess index, its term or 0)."""
    t = np.ascontiguousarray(np.asarray(terms, dtype=np.uint64))
    out = C.c_uint64(0)
    f = lib().oracle_find_conflict_by_term
    f.restype = C.c_uint64
    f.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    return int(f(first_index, snap_term, _vp(t), len(t), index, term, C.byref(out))), out.value


def committed_entry_in_current_term(first_index, snap_term, terms, committed, term):
    """raft.committedEntryInCurrentTerm on an explicit log."""
    t = np.ascontiguousarray(np.asarray(terms, dtype=np.uint64))
    return bool(lib().oracle_committed_entry_in_current_term(first_index, snap_term, _vp(t), len(t), committed, term))


def batch_maybe_commit(match, logs, term, committed, in_mask=None, out_mask=None, uniform_voters=0):
    """raft.maybeCommit for every group.  logs = (log_off[G+1], log_terms, log_first[G],
    log_snap_term[G]).  Returns (new committed, advanced u8)."""
    match = _chk(match, np.uint64)
    S, G = match.shape
    off, terms, first, snap = (_chk(np.ascontiguousarray(a), np.uint64) for a in logs)
    term = _chk(np.ascontiguousarray(term), np.uint64)
    com = np.array(committed, dtype=np.uint64, copy=True)
    adv = np.zeros(G, dtype=np.uint8)
    lib().oracle_batch_maybe_commit(G, S, _vp(match), match.strides[0] // 8, _vp(_chk(in_mask)),
                                    _vp(_chk(out_mask)), _mask_bytes(in_mask, out_mask), uniform_voters,
                                    _vp(off), _vp(terms), _vp(first), _vp(snap), _vp(term), _vp(com), _vp(adv))
    return com, adv


def step_leader_msgs(match, active, acks, last_index, msgs):
    """stepLeader bookkeeping for msgs in order; mutates match/active/acks/last_index.
    Returns (dropped, ignored)."""
    match = _chk(match, np.uint64)
    msgs = _chk(msgs, MSG_DTYPE)
    S, G = match.shape
    counts = np.zeros(2, dtype=np.uint64)
    lib().oracle_step_leader_msgs(G, S, _vp(match), match.strides[0] // 8, _vp(_chk(active)), _vp(_chk(acks)),
                                  active.dtype.itemsize, _vp(_chk(last_index, np.uint64)), _vp(msgs),
                                  len(msgs), _vp(counts))
    return int(counts[0]), int(counts[1])


class _PbMsg(C.Structure):
    _fields_ = [("type", C.c_int32), ("to", C.c_uint64), ("from_", C.c_uint64), ("term", C.c_uint64),
                ("log_term", C.c_uint64), ("index", C.c_uint64), ("commit", C.c_uint64),
                ("reject_hint", C.c_uint64), ("vote", C.c_uint64), ("reject", C.c_int), ("has_ctx", C.c_int),
                ("ctx_off", C.c_uint64), ("ctx_len", C.c_uint64), ("nested", C.c_int)]


def message_unmarshal(data):
    """raftpb.Message.Unmarshal on `data` (bytes).  Returns (rc, fields): rc 0 = nil error, < 0 = the
    error the reference returns; fields = dict of the scalar fields, context bytes (None = nil), and
    `nested` (a non-empty entries / snapshot / responses payload was present, not decoded)."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(0, dtype=np.uint8)
    m = _PbMsg()
    rc = int(lib().oracle_message_unmarshal(_vp(buf) if len(buf) else None, len(buf), C.byref(m)))
    if rc:
        return rc, None
    ctx = bytes(buf[m.ctx_off:m.ctx_off + m.ctx_len]) if m.has_ctx else None
    return 0, {"type": m.type, "to": m.to, "from": m.from_, "term": m.term, "logTerm": m.log_term, "index": m.index,
               "commit": m.commit, "rejectHint": m.reject_hint, "vote": m.vote, "reject": bool(m.reject),
               "context": ctx, "nested": bool(m.nested)}


def _campaign_args(campaign, votes, S, G):
    if campaign is None:
        return None, None, 0
    campaign = _chk(campaign, np.uint8)
    votes = _chk(votes, np.uint8)
    assert campaign.shape == (G,) and votes.shape == (S, G)
    return _vp(campaign), _vp(votes), votes.strides[0]


def step_wire_msgs(match, active, acks, ids, term, read_ctx, read_ctx_len, data, offsets, groups, campaign=None,
                   votes=None):
    """Serialized raftpb.Message values stepped in order; mutates match/active/acks (and votes [S][G] for the
    groups whose campaign[g] names a vote response type).  ids [S][G] = ProgressMap key per slot.
    Returns (out_msgs, out_status, counts[10])."""
    match = _chk(match, np.uint64)
    ids = _chk(ids, np.uint64)
    S, G = match.shape
    assert ids.shape == (S, G)
    offsets = _chk(np.ascontiguousarray(offsets), np.uint64)
    groups = _chk(np.ascontiguousarray(groups), np.uint32)
    n = len(groups)
    assert len(offsets) == n + 1
    data = _chk(np.ascontiguousarray(data), np.uint8)
    out_msgs = np.zeros(n, dtype=MSG_DTYPE)
    out_status = np.zeros(n, dtype=np.uint8)
    counts = np.zeros(10, dtype=np.uint64)
    lib().oracle_step_wire_msgs(G, S, _vp(match), match.strides[0] // 8, _vp(_chk(active)), _vp(_chk(acks)),
                                active.dtype.itemsize, _vp(ids), ids.strides[0] // 8, _vp(_chk(term, np.uint64)),
                                _vp(_chk(read_ctx, np.uint64)), _vp(_chk(read_ctx_len, np.uint8)),
                                _vp(data) if len(data) else None, len(data), _vp(offsets), _vp(groups), n,
                                _vp(out_msgs), _vp(out_status), _vp(counts), *_campaign_args(campaign, votes, S, G))
    return out_msgs, out_status, counts


def record_votes(votes, msgs):
    """RecordVote for a stream of vote responses in order (first vote of a cell sticks); mutates votes [S][G]."""
    votes = _chk(votes, np.uint8)
    S, G = votes.shape
    msgs = _chk(msgs, MSG_DTYPE)
    return int(lib().oracle_record_votes(G, S, _vp(votes), votes.strides[0], _vp(msgs), len(msgs)))


def decode_wire_msgs(ids, term, read_ctx, read_ctx_len, data, offsets, groups, campaign=None, votes=None):
    """Decode + classify only: (records, aux = RejectHint of rejections, status, counts[10]).  A rejection
    with LogTerm > 0 is status 8 and record type 0xFE (the host steps it).  Vote responses are polled into
    votes as in step_wire_msgs."""
    ids = _chk(ids, np.uint64)
    S, G = ids.shape
    offsets = _chk(np.ascontiguousarray(offsets), np.uint64)
    groups = _chk(np.ascontiguousarray(groups), np.uint32)
    n = len(groups)
    data = _chk(np.ascontiguousarray(data), np.uint8)
    out_msgs = np.zeros(n, dtype=MSG_DTYPE)
    aux = np.zeros(n, dtype=np.uint64)
    status = np.zeros(n, dtype=np.uint8)
    counts = np.zeros(10, dtype=np.uint64)
    lib().oracle_decode_wire_msgs(G, S, _vp(ids), ids.strides[0] // 8, _vp(_chk(term, np.uint64)),
                                  _vp(_chk(read_ctx, np.uint64)), _vp(_chk(read_ctx_len, np.uint8)),
                                  _vp(data) if len(data) else None, len(data), _vp(offsets), _vp(groups), n,
                                  _vp(out_msgs), _vp(aux), _vp(status), _vp(counts),
                                  *_campaign_args(campaign, votes, S, G))
    return out_msgs, aux, status, counts


def message_marshal(type=0, to=0, frm=0, term=0, logTerm=0, index=0, commit=0, reject=False, rejectHint=0,
                    context=None, vote=0):
    """Message.Marshal for a message without entries / snapshot / responses -> bytes."""
    f = lib().oracle_message_marshal
    f.restype = C.c_int64
    f.argtypes = [C.c_int32] + [C.c_uint64] * 6 + [C.c_int, C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                                  C.c_uint64]
    out = np.zeros(256 + (len(context) if context else 0), dtype=np.uint8)
    n = f(type, to, frm, term, logTerm, index, commit, int(reject), rejectHint, context,
          len(context) if context is not None else 0, vote, _vp(out), len(out))
    assert n > 0, n
    return out[:n].tobytes()


def bcast_heartbeat(match, committed, ids, term, read_ctx, read_ctx_len, self_mask, msg_stride=72):
    """bcastHeartbeat for every group -> (uint8 [S][G][msg_stride] zero-padded messages, uint8 [S][G] lengths)."""
    match = _chk(match, np.uint64)
    ids = _chk(ids, np.uint64)
    S, G = match.shape
    assert ids.shape == (S, G)
    out = np.zeros((S, G, msg_stride), dtype=np.uint8)
    lens = np.zeros((S, G), dtype=np.uint8)
    f = lib().oracle_bcast_heartbeat
    f.restype = None
    f.argtypes = [C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
    self_mask = _chk(self_mask)
    f(G, S, _vp(match), match.strides[0] // 8, _vp(_chk(committed, np.uint64)), _vp(ids), ids.strides[0] // 8,
      _vp(_chk(term, np.uint64)), _vp(_chk(read_ctx, np.uint64)), _vp(_chk(read_ctx_len, np.uint8)), _vp(self_mask),
      self_mask.dtype.itemsize, _vp(out), msg_stride, _vp(lens))
    return out, lens


def batch_tick(election_elapsed, heartbeat_elapsed, randomized_timeout, state, flags, election_timeout,
               heartbeat_timeout, check_quorum, nslots, active, self_mask, in_mask=None, out_mask=None,
               learner_mask=None, uniform_voters=0):
    """RawNode.Tick for every group (raft.go:823-862): mutates the two elapsed planes, state, flags and active in
    place; returns the action byte per group."""
    G = len(state)
    for a, dt in ((election_elapsed, np.uint32), (heartbeat_elapsed, np.uint32), (randomized_timeout, np.uint32),
                  (state, np.uint8), (flags, np.uint8)):
        _chk(a, dt)
        assert len(a) == G
    act = np.empty(G, dtype=np.uint8)
    lib().oracle_batch_tick(G, nslots, _vp(election_elapsed), _vp(heartbeat_elapsed), _vp(randomized_timeout),
                            _vp(state), _vp(flags), election_timeout, heartbeat_timeout, int(bool(check_quorum)),
                            _vp(_chk(active)), _vp(_chk(in_mask)), _vp(_chk(out_mask)), _vp(_chk(learner_mask)),
                            active.dtype.itemsize, uniform_voters, _vp(_chk(self_mask)), _vp(act))
    return act


def tick_heard_from_leader(election_elapsed, groups):
    """stepFollower's `r.electionElapsed = 0` on MsgApp / MsgHeartbeat / MsgSnap (raft.go:1681-1692) for a list of
    groups, in place; ids outside the plane are skipped and counted."""
    g = np.asarray(groups, dtype=np.int64)
    ok = g < len(election_elapsed)
    election_elapsed[g[ok]] = 0
    return int((~ok).sum())


def batch_check_quorum(active, nslots, self_mask, in_mask=None, out_mask=None, learner_mask=None,
                       uniform_voters=0):
    """MsgCheckQuorum for every group: returns QuorumActive u8; mutates active (all but self cleared)."""
    active = _chk(active)
    G = active.shape[0]
    out = np.empty(G, dtype=np.uint8)
    lib().oracle_batch_check_quorum(G, nslots, _vp(active), _vp(_chk(in_mask)), _vp(_chk(out_mask)),
                                    _vp(_chk(learner_mask)), active.dtype.itemsize, uniform_voters,
                                    _vp(_chk(self_mask)), _vp(out))
    return out


def batch_heartbeat_commit(match, committed):
    """sendHeartbeat's commit = min(Match, committed) for every (slot, group)."""
    match = _chk(match, np.uint64)
    S, G = match.shape
    out = np.empty((S, G), dtype=np.uint64)
    lib().oracle_batch_heartbeat_commit(G, S, _vp(match), match.strides[0] // 8,
                                        _vp(_chk(committed, np.uint64)), _vp(out), G)
    return out


def selfcheck_committed(cases, seed=0x5EED):
    """Main vs alternative CommittedIndex on `cases` random inputs, in C. Returns #mismatches."""
    return int(lib().oracle_selfcheck_committed(cases, seed))


def bench_leaf_ns(voters, iters=20_000_000, seed=0x5EED0001):
    """BenchmarkMajorityConfig_CommittedIndex/voters=V restated: ns per op, one thread."""
    sink = C.c_uint64()
    return float(lib().oracle_bench_leaf_ns(voters, iters, seed, C.byref(sink)))


def bench_soa5(match, threads=0):
    """Best-CPU figure: SoA planes + median-of-5 network on the host cores -> (out, threads used)."""
    match = _chk(match, np.uint64)
    assert match.shape[0] >= 5
    out = np.empty(match.shape[1], dtype=np.uint64)
    used = lib().oracle_bench_soa5(match.shape[1], _vp(match), match.strides[0] // 8, _vp(out), threads)
    return out, int(used)


class CpuBaseline:
    """bench.py cpu_baseline leg: per-group id set + id-keyed map, parallel loop."""

    def __init__(self, match, V):
        match = _chk(match, np.uint64)
        self.G = match.shape[1]
        self.h = lib().oracle_bench_create(self.G, V, _vp(match), match.strides[0] // 8)
        self.out = np.empty(self.G, dtype=np.uint64)

    def run(self, threads):
        return lib().oracle_bench_run(self.h, _vp(self.out), threads)

    def close(self):
        if self.h:
            lib().oracle_bench_destroy(self.h)
            self.h = None


# ---- tracker.Progress / tracker.Inflights (progress_oracle.c) --------------------------------

STATE_PROBE, STATE_REPLICATE, STATE_SNAPSHOT = 0, 1, 2
STATE_NAMES = {"StateProbe": 0, "StateReplicate": 1, "StateSnapshot": 2}
EV_APP_RESP, EV_HEARTBEAT_RESP, EV_UNREACHABLE, EV_SNAP_STATUS = 4, 9, 10, 11
EV_APPENDED, EV_SENT_APP, EV_SENT_SNAP, EV_NEEDS_HOST = 0xF0, 0xF1, 0xF2, 0xFE
ACT_SEND_APPEND, ACT_UPDATED, ACT_WAS_PAUSED, ACT_CAUGHT_UP, ACT_ERROR, ACT_DEFERRED, ACT_DROPPED = 1, 2, 4, 8, 32, 64, 128
STEP_LIMIT = 64  # events per Progress per batch the engine steps (more: deferred)


class Progress:
    """One tracker.Progress (tracker/progress.go:30-98) with its Inflights, by handle."""

    def __init__(self, state=0, match=0, next=0, pending=0, paused=False, inflight_size=256, inflight_max_bytes=0):
        self._h = lib().oracle_progress_new(state, match, next, pending, int(paused), inflight_size, inflight_max_bytes)
        self._size = inflight_size

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_progress_free(self._h)
            self._h = None

    def get(self):
        out = np.zeros(7, dtype=np.uint64)
        lib().oracle_progress_get(self._h, _vp(out))
        return {"state": int(out[0]), "match": int(out[1]), "next": int(out[2]), "pending": int(out[3]),
                "paused": bool(out[4]), "is_paused": bool(out[5])}

    def inflights(self):
        out = np.zeros(5, dtype=np.uint64)
        idx = np.zeros(max(self._size, 1), dtype=np.uint64)
        by = np.zeros(max(self._size, 1), dtype=np.uint64)
        lib().oracle_inflights_get(self._h, _vp(out), _vp(idx), _vp(by))
        return {"start": int(out[0]), "count": int(out[1]), "bytes": int(out[2]), "size": int(out[3]),
                "full": bool(out[4]), "index": idx[:self._size].tolist(), "sizes": by[:self._size].tolist()}

    def set_start(self, start):
        lib().oracle_inflights_set_start(self._h, start)

    def add(self, index, nbytes):
        return int(lib().oracle_inflights_add(self._h, index, nbytes))

    def free_le(self, to):
        lib().oracle_inflights_free_le(self._h, to)

    def reset_inflights(self):
        lib().oracle_inflights_reset(self._h)

    def maybe_update(self, n):
        return bool(lib().oracle_progress_maybe_update(self._h, n))

    def maybe_decr_to(self, rejected, hint):
        return bool(lib().oracle_progress_maybe_decr_to(self._h, rejected, hint))

    def become(self, state, snapshoti=0):
        lib().oracle_progress_become(self._h, state, snapshoti)


def progress_planes(S, G, cap):
    """Zeroed [slot][group] planes of the per-peer replication state (every Progress in
    StateProbe, Match 0, Next 0, empty Inflights of capacity cap)."""
    return {"match": np.zeros((S, G), dtype=np.uint64), "next": np.zeros((S, G), dtype=np.uint64),
            "pstate": np.zeros((S, G), dtype=np.uint8), "pending": np.zeros((S, G), dtype=np.uint64),
            "infl_meta": np.zeros((S, G), dtype=np.uint32), "infl_bytes": np.zeros((S, G), dtype=np.uint64),
            "ring_index": np.zeros((max(cap, 1), S, G), dtype=np.uint64),
            "ring_bytes": np.zeros((max(cap, 1), S, G), dtype=np.uint64)}


def step_progress(pl, cap, max_bytes, active, acks, last_index, first_index, msgs, aux=None, limit=STEP_LIMIT):
    """stepLeader's Progress branches for a batch of events, in stream order per Progress;
    mutates the planes, active, acks, last_index.  Returns the per-event action bytes."""
    S, G = pl["match"].shape
    for k in ("match", "next", "pstate", "pending", "infl_meta", "infl_bytes", "ring_index", "ring_bytes"):
        assert pl[k].flags["C_CONTIGUOUS"] and pl[k].shape[-2:] == (S, G), k
    assert pl["ring_index"].shape[0] == max(cap, 1)
    msgs = _chk(msgs)
    n = len(msgs)
    if aux is not None:
        aux = _chk(aux, np.uint64)
        assert len(aux) == n
    actions = np.zeros(n, dtype=np.uint8)
    lib().oracle_step_progress(G, S, G, _vp(pl["match"]), _vp(pl["next"]), _vp(pl["pstate"]), _vp(pl["pending"]),
                               _vp(pl["infl_meta"]), _vp(pl["infl_bytes"]), _vp(pl["ring_index"]),
                               _vp(pl["ring_bytes"]), cap, max_bytes, _vp(_chk(active)), _vp(_chk(acks)),
                               active.dtype.itemsize, _vp(_chk(last_index, np.uint64)),
                               _vp(_chk(first_index, np.uint64)), _vp(msgs), _vp(aux) if aux is not None else None,
                               n, limit, _vp(actions))
    return actions


SEND_NONE, SEND_APPEND, SEND_EMPTY, SEND_SNAPSHOT = 0, 1, 2, 3


def plan_appends(pl, cap, max_bytes, rows, self_mask, active, first_index, last_index, send_if_empty=True):
    """bcastAppend as a plan: uint8 [S][G], what maybeSendAppend(to, sendIfEmpty) decides for every Progress row
    (rows: slot bitmask of the rows that exist per group, None = all)."""
    S, G = pl["next"].shape
    plan = np.zeros((S, G), dtype=np.uint8)
    f = lib().oracle_plan_appends
    f.restype = None
    f.argtypes = [C.c_uint64, C.c_int, C.c_uint64] + [C.c_void_p] * 4 + [C.c_int, C.c_uint64] + [C.c_void_p] * 3 + [
        C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    self_mask, active = _chk(self_mask), _chk(active)
    f(G, S, pl["next"].strides[0] // 8, _vp(_chk(pl["next"], np.uint64)), _vp(_chk(pl["pstate"], np.uint8)),
      _vp(_chk(pl["infl_meta"], np.uint32)), _vp(_chk(pl["infl_bytes"], np.uint64)), cap, max_bytes,
      _vp(_chk(rows)) if rows is not None else None, _vp(self_mask), _vp(active), active.dtype.itemsize,
      _vp(_chk(first_index, np.uint64)), _vp(_chk(last_index, np.uint64)), int(bool(send_if_empty)), _vp(plan))
    return plan
