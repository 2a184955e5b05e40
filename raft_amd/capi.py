"""ctypes binding of include/rafthip.h — one Python method per C entry point.

Plumbing only: argument marshalling, error codes → exceptions.  All compute
happens in librafthip.so on the GPU; a missing library raises RafthipError at
load time (there is deliberately no fallback).
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ACK_DTYPE = np.dtype([("group", "<u4"), ("slot", "<u4"), ("index", "<u8")])
PAIR_DTYPE = np.dtype([("group", "<u8"), ("commit", "<u8")])
MSG_DTYPE = np.dtype([("group", "<u4"), ("slot", "u1"), ("type", "u1"), ("reject", "u1"), ("flags", "u1"),
                      ("index", "<u8")])
MSG_APP_RESP, MSG_HEARTBEAT_RESP, MSG_APPENDED, MSGF_READ_CTX = 4, 9, 0xF0, 1
PLANE_HEARTBEAT = 3
FLAG_WIDE_BITONIC = 1
FLAG_SCATTER_ATOMIC, FLAG_SCATTER_PARTITION = 2, 4
FLAG_STEP_LINKED, FLAG_STEP_PARTITION = 8, 16

PLANE_MATCH, PLANE_COMMIT, PLANE_VOTES = 0, 1, 2
VOTE_PENDING, VOTE_LOST, VOTE_WON = 1, 2, 3

# every symbol include/rafthip.h declares (tests check the .so exports them all)
SYMBOLS = [
    "rafthip_abi_version", "rafthip_last_error", "rafthip_device_count", "rafthip_create",
    "rafthip_destroy", "rafthip_synchronize", "rafthip_mask_bytes", "rafthip_stats",
    "rafthip_set_uniform_config", "rafthip_upload_config", "rafthip_upload_match",
    "rafthip_reset_match", "rafthip_read_match", "rafthip_upload_votes", "rafthip_reset_votes",
    "rafthip_upload_active", "rafthip_upload_acks", "rafthip_upload_commit",
    "rafthip_scatter_acks", "rafthip_commit", "rafthip_commit_changed", "rafthip_read_commit",
    "rafthip_tally_votes", "rafthip_read_vote_result", "rafthip_quorum_active",
    "rafthip_read_quorum_active", "rafthip_ack_quorum", "rafthip_read_ack_quorum",
    "rafthip_plane", "rafthip_batch_committed_index", "rafthip_batch_vote_result",
    "rafthip_ingest_msgs", "rafthip_commit_advance", "rafthip_upload_term_range",
    "rafthip_heartbeat_commit", "rafthip_upload_self", "rafthip_check_quorum",
    "rafthip_read_active", "rafthip_read_acks", "rafthip_read_term_range",
    "rafthip_committed_in_term", "rafthip_read_committed_in_term",
    "rafthip_upload_ids", "rafthip_upload_term", "rafthip_upload_read_ctx", "rafthip_ingest_wire",
    "rafthip_progress_config", "rafthip_upload_progress", "rafthip_read_progress", "rafthip_upload_inflights",
    "rafthip_read_inflights", "rafthip_upload_first_index", "rafthip_step_progress", "rafthip_step_wire",
    "rafthip_merge_acks", "rafthip_host_alloc", "rafthip_host_free",
    "rafthip_record_votes", "rafthip_read_votes", "rafthip_step_dense", "rafthip_upload_campaign",
    "rafthip_emit_heartbeats", "rafthip_plan_appends", "rafthip_commit_changed_static",
    "rafthip_comm_unique_id", "rafthip_comm_create", "rafthip_comm_create_all", "rafthip_comm_destroy",
    "rafthip_comm_info", "rafthip_commit_allgather", "rafthip_commit_allgather_many", "rafthip_commit_many",
    "rafthip_commit_allgather_dense", "rafthip_commit_allgather_dense_many", "rafthip_commit_report_many",
    "rafthip_init_progress_row",
    "rafthip_tick_config", "rafthip_upload_tick_state", "rafthip_read_tick_state", "rafthip_tick",
    "rafthip_read_tick_actions", "rafthip_read_tick_events", "rafthip_emit_heartbeats_due",
    "rafthip_tick_heard_from_leader", "rafthip_tick_many",
]
COMM_ID_BYTES = 128
# RawNode.Tick (rafthip.h RAFTHIP_TICK_*): StateType of raft.go:48-53, action bits, flag bits
STATE_FOLLOWER, STATE_CANDIDATE, STATE_LEADER, STATE_PRE_CANDIDATE = 0, 1, 2, 3
TICK_HUP, TICK_CHECK_QUORUM, TICK_STEP_DOWN, TICK_ABORT_TRANSFER, TICK_BEAT = 1, 2, 4, 8, 16
TICK_ACTIONS = 5
TICKF_PROMOTABLE, TICKF_TRANSFER = 1, 2
TICK_EVENT_DTYPE = np.dtype([("group", "<u4"), ("action", "<u4")])
MSG_VOTE_RESP, MSG_PRE_VOTE_RESP = 6, 18

STATE_PROBE, STATE_REPLICATE, STATE_SNAPSHOT, PROGRESS_PAUSED = 0, 1, 2, 4
MSG_UNREACHABLE, MSG_SNAP_STATUS, MSG_SENT_APP, MSG_SENT_SNAP = 10, 11, 0xF1, 0xF2
ACT_SEND_APPEND, ACT_UPDATED, ACT_WAS_PAUSED, ACT_CAUGHT_UP, ACT_ERROR, ACT_DEFERRED, ACT_DROPPED = 1, 2, 4, 8, 32, 64, 128
STEP_LIMIT = 64


WIRE_STATUSES = 10
(WIRE_APPLIED, WIRE_OTHER_TYPE, WIRE_NO_PROGRESS, WIRE_STALE_TERM, WIRE_HIGHER_TERM, WIRE_NESTED, WIRE_MALFORMED,
 WIRE_BAD_GROUP, WIRE_NEEDS_LOG, WIRE_VOTE) = range(10)
MSG_NEEDS_HOST = 0xFE
HEARTBEAT_STRIDE = 72
SEND_NONE, SEND_APPEND, SEND_EMPTY, SEND_SNAPSHOT = 0, 1, 2, 3
MSGF_READ_CTX, MSGF_HAS_CTX = 1, 2


class RafthipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rafthip error %d: %s" % (code, msg))
        self.code = code


class _Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("num_groups", C.c_uint64),
                ("num_slots", C.c_uint32), ("flags", C.c_uint32), ("stream", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("commit_launches", "groups_evaluated", "acks_applied",
                                           "acks_rejected", "changed_reported", "bytes_uploaded",
                                           "bytes_downloaded", "advanced_reported", "msgs_ingested",
                                           "msgs_dropped", "msgs_ignored")]


def library_path():
    return os.environ.get("RAFTHIP_LIBRARY", os.path.join(_HERE, "lib", "librafthip.so"))


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so.7; if this library
    pulls in /opt/rocm's copy first and torch is imported later, the process holds two runtimes and
    the second one finds no GPU ("No HIP GPUs are available").  When torch is installed but not yet
    imported, load its runtime first so that librafthip.so binds to the same one either way
    (RAFTHIP_SYSTEM_HIP=1 keeps the system runtime: for processes that never import torch)."""
    if "torch" in sys.modules or os.environ.get("RAFTHIP_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        for d in (spec.submodule_search_locations or []) if spec else []:
            cand = os.path.join(d, "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
                return
    except (ImportError, OSError, ValueError):
        pass


def load_library():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RafthipError(-2, "%s not found: build it with `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    _share_torch_hip_runtime()
    L = C.CDLL(path)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    sig = {
        "rafthip_abi_version": (u32, []),
        "rafthip_last_error": (C.c_char_p, []),
        "rafthip_device_count": (i32, [C.POINTER(C.c_int)]),
        "rafthip_create": (i32, [C.POINTER(_Config), C.POINTER(vp)]),
        "rafthip_destroy": (i32, [vp]),
        "rafthip_synchronize": (i32, [vp]),
        "rafthip_mask_bytes": (i32, [vp]),
        "rafthip_stats": (i32, [vp, C.POINTER(Stats)]),
        "rafthip_set_uniform_config": (i32, [vp, u32]),
        "rafthip_upload_config": (i32, [vp, u64, u64, vp, vp, vp]),
        "rafthip_upload_match": (i32, [vp, u64, u64, vp, u64]),
        "rafthip_reset_match": (i32, [vp, u64, u64]),
        "rafthip_read_match": (i32, [vp, u64, u64, vp, u64]),
        "rafthip_upload_votes": (i32, [vp, u64, u64, vp, u64]),
        "rafthip_reset_votes": (i32, [vp, u64, u64]),
        "rafthip_upload_active": (i32, [vp, u64, u64, vp]),
        "rafthip_upload_acks": (i32, [vp, u64, u64, vp]),
        "rafthip_upload_commit": (i32, [vp, u64, u64, vp]),
        "rafthip_scatter_acks": (i32, [vp, vp, u64]),
        "rafthip_commit": (i32, [vp]),
        "rafthip_commit_changed": (i32, [vp, vp, u64, C.POINTER(u64)]),
        "rafthip_commit_many": (i32, [C.POINTER(vp), u32]),
        "rafthip_commit_changed_static": (i32, [vp, vp, u64, C.POINTER(u64)]),
        "rafthip_read_commit": (i32, [vp, u64, u64, vp]),
        "rafthip_tally_votes": (i32, [vp]),
        "rafthip_read_vote_result": (i32, [vp, u64, u64, vp, vp, vp]),
        "rafthip_quorum_active": (i32, [vp]),
        "rafthip_read_quorum_active": (i32, [vp, u64, u64, vp]),
        "rafthip_ack_quorum": (i32, [vp]),
        "rafthip_read_ack_quorum": (i32, [vp, u64, u64, vp]),
        "rafthip_plane": (i32, [vp, i32, C.POINTER(vp), C.POINTER(u64)]),
        "rafthip_batch_committed_index": (i32, [i32, u64, u32, vp, vp, vp, vp]),
        "rafthip_batch_vote_result": (i32, [i32, u64, u32, vp, vp, vp, vp]),
        "rafthip_ingest_msgs": (i32, [vp, vp, u64]),
        "rafthip_commit_advance": (i32, [vp, vp, u64, C.POINTER(u64)]),
        "rafthip_upload_term_range": (i32, [vp, u64, u64, vp, vp]),
        "rafthip_heartbeat_commit": (i32, [vp, u64, u64, vp, u64]),
        "rafthip_upload_self": (i32, [vp, u64, u64, vp]),
        "rafthip_check_quorum": (i32, [vp]),
        "rafthip_tick_config": (i32, [vp, C.c_uint32, C.c_uint32, C.c_int]),
        "rafthip_upload_tick_state": (i32, [vp, u64, u64, vp, vp, vp, vp, vp]),
        "rafthip_read_tick_state": (i32, [vp, u64, u64, vp, vp, vp, vp, vp]),
        "rafthip_tick": (i32, [vp, vp]),
        "rafthip_tick_many": (i32, [vp, C.c_uint32, vp]),
        "rafthip_tick_heard_from_leader": (i32, [vp, vp, u64, C.POINTER(u64)]),
        "rafthip_read_tick_actions": (i32, [vp, u64, u64, vp]),
        "rafthip_read_tick_events": (i32, [vp, C.c_uint32, vp, u64, C.POINTER(u64)]),
        "rafthip_read_active": (i32, [vp, u64, u64, vp]),
        "rafthip_read_acks": (i32, [vp, u64, u64, vp]),
        "rafthip_read_term_range": (i32, [vp, u64, u64, vp, vp]),
        "rafthip_committed_in_term": (i32, [vp]),
        "rafthip_read_committed_in_term": (i32, [vp, u64, u64, vp]),
        "rafthip_upload_ids": (i32, [vp, u64, u64, vp, u64]),
        "rafthip_upload_term": (i32, [vp, u64, u64, vp]),
        "rafthip_upload_read_ctx": (i32, [vp, u64, u64, vp, vp]),
        "rafthip_upload_campaign": (i32, [vp, u64, u64, vp]),
        "rafthip_emit_heartbeats": (i32, [vp, u64, u64, vp, vp]),
        "rafthip_emit_heartbeats_due": (i32, [vp, u64, u64, vp, vp]),
        "rafthip_plan_appends": (i32, [vp, u64, u64, i32, vp, vp, u64]),
        "rafthip_ingest_wire": (i32, [vp, vp, vp, vp, u64, vp, vp, vp]),
        "rafthip_progress_config": (i32, [vp, u32, u64]),
        "rafthip_upload_progress": (i32, [vp, u64, u64, vp, vp, vp, u64]),
        "rafthip_read_progress": (i32, [vp, u64, u64, vp, vp, vp, vp, vp, u64]),
        "rafthip_init_progress_row": (i32, [vp, u64, u32, C.c_uint8, u64, u64]),
        "rafthip_upload_inflights": (i32, [vp, u64, u64, vp, vp, vp, vp, u64]),
        "rafthip_read_inflights": (i32, [vp, u64, u64, vp, vp, u64]),
        "rafthip_upload_first_index": (i32, [vp, u64, u64, vp]),
        "rafthip_step_progress": (i32, [vp, vp, vp, u64, vp]),
        "rafthip_step_wire": (i32, [vp, vp, vp, vp, u64, vp, vp, vp, vp]),
        "rafthip_merge_acks": (i32, [vp, u64, u64, vp, u64]),
        "rafthip_step_dense": (i32, [vp, u64, u64, vp, u64, vp, vp, vp, u64]),
        "rafthip_record_votes": (i32, [vp, vp, u64]),
        "rafthip_read_votes": (i32, [vp, u64, u64, vp, u64]),
        "rafthip_comm_unique_id": (i32, [vp]),
        "rafthip_comm_create": (i32, [vp, vp, i32, i32, u64, u64, C.POINTER(vp)]),
        "rafthip_comm_create_all": (i32, [C.POINTER(vp), i32, C.POINTER(u64), u64, C.POINTER(vp)]),
        "rafthip_comm_destroy": (i32, [vp]),
        "rafthip_comm_info": (i32, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(u64)]),
        "rafthip_commit_allgather": (i32, [vp, i32, vp, u64, vp, C.POINTER(u64)]),
        "rafthip_commit_allgather_dense": (i32, [vp, i32, vp, u64, C.POINTER(u64)]),
        "rafthip_commit_report_many": (i32, [C.POINTER(vp), u32, i32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]),
        "rafthip_commit_allgather_dense_many": (i32, [C.POINTER(vp), i32, i32, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]),
        "rafthip_commit_allgather_many": (i32, [C.POINTER(vp), i32, i32, C.POINTER(vp), C.POINTER(u64),
                                                C.POINTER(vp), C.POINTER(u64)]),
        "rafthip_host_alloc": (i32, [C.POINTER(vp), u64]),
        "rafthip_host_free": (i32, [vp]),
    }
    assert sorted(sig) == sorted(SYMBOLS)
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise RafthipError(rc, load_library().rafthip_last_error().decode("utf-8", "replace"))


def mask_dtype(nslots):
    return np.dtype(np.uint8 if nslots <= 8 else np.uint16 if nslots <= 16
                    else np.uint32 if nslots <= 32 else np.uint64)


def _ptr(a):
    """Host numpy array, torch tensor (host or device), raw int address, or None → void*."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        if not a.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return C.c_void_p(a.data_ptr())
    raise TypeError("unsupported buffer type %r" % type(a))


class PinnedArray:
    """A numpy array over page-locked host memory (rafthip_host_alloc); free() or use as a context manager."""

    def __init__(self, shape, dtype):
        self._L = load_library()
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        _check(self._L.rafthip_host_alloc(C.byref(p), max(n, 1)))
        self._p = p
        self.array = np.frombuffer((C.c_char * max(n, 1)).from_address(p.value), dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self._p is not None:
            self.array = None
            _check(self._L.rafthip_host_free(self._p))
            self._p = None

    def __enter__(self):
        return self.array

    def __exit__(self, *exc):
        self.free()


class Engine:
    """One engine per GPU: G groups × S slots of device-resident state."""

    def __init__(self, num_groups, num_slots, device=0, stream=None, flags=0):
        self._h = None
        L = load_library()
        cfg = _Config(C.sizeof(_Config), device, num_groups, num_slots, flags, stream)
        h = C.c_void_p()
        _check(L.rafthip_create(C.byref(cfg), C.byref(h)))
        self._L, self._h = L, h
        self.G, self.S, self.device = num_groups, num_slots, device
        self.mask_dtype = mask_dtype(num_slots)
        assert L.rafthip_mask_bytes(h) == self.mask_dtype.itemsize

    def close(self):
        if getattr(self, "_h", None):
            self._L.rafthip_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _range(self, first, count):
        return first, (self.G - first if count is None else count)

    # configuration
    def set_uniform_config(self, voters):
        _check(self._L.rafthip_set_uniform_config(self._h, voters))

    def upload_config(self, in_mask, out_mask, learner_mask=None, first=0, count=None):
        first, count = self._range(first, count)
        _check(self._L.rafthip_upload_config(self._h, first, count, _ptr(in_mask), _ptr(out_mask),
                                             _ptr(learner_mask)))

    # state
    def upload_match(self, match, first=0, count=None, src_stride=None):
        first, count = self._range(first, count)
        if src_stride is None:
            src_stride = match.shape[1] if hasattr(match, "shape") else count
        _check(self._L.rafthip_upload_match(self._h, first, count, _ptr(match), src_stride))

    def reset_match(self, first=0, count=None):
        first, count = self._range(first, count)
        _check(self._L.rafthip_reset_match(self._h, first, count))

    def read_match(self, first=0, count=None):
        first, count = self._range(first, count)
        out = np.empty((self.S, count), dtype=np.uint64)
        _check(self._L.rafthip_read_match(self._h, first, count, _ptr(out), count))
        return out

    def upload_votes(self, votes, first=0, count=None, src_stride=None):
        first, count = self._range(first, count)
        if src_stride is None:
            src_stride = votes.shape[1] if hasattr(votes, "shape") else count
        _check(self._L.rafthip_upload_votes(self._h, first, count, _ptr(votes), src_stride))

    def reset_votes(self, first=0, count=None):
        first, count = self._range(first, count)
        _check(self._L.rafthip_reset_votes(self._h, first, count))

    def upload_active(self, bits, first=0, count=None):
        first, count = self._range(first, count)
        _check(self._L.rafthip_upload_active(self._h, first, count, _ptr(bits)))

    def upload_acks(self, bits, first=0, count=None):
        first, count = self._range(first, count)
        _check(self._L.rafthip_upload_acks(self._h, first, count, _ptr(bits)))

    def upload_commit(self, commit, first=0, count=None):
        first, count = self._range(first, count)
        _check(self._L.rafthip_upload_commit(self._h, first, count, _ptr(commit)))

    # compute
    def scatter_acks(self, recs, n=None):
        if n is None:
            n = len(recs)
        _check(self._L.rafthip_scatter_acks(self._h, _ptr(recs), n))

    def merge_acks(self, acks, first=0, count=None, stride=None):
        """acks: u64 [S][count] plane (numpy, torch or raw device pointer with stride), 0 = no ack."""
        first, count = self._range(first, count)
        _check(self._L.rafthip_merge_acks(self._h, first, count, _ptr(acks), count if stride is None else stride))

    def read_votes(self, first=0, count=None):
        first, count = self._range(first, count)
        out = np.empty((self.S, count), dtype=np.uint8)
        _check(self._L.rafthip_read_votes(self._h, first, count, _ptr(out), count))
        return out

    def record_votes(self, msgs, n=None):
        if n is None:
            n = len(msgs)
        _check(self._L.rafthip_record_votes(self._h, _ptr(msgs), n))

    def commit(self):
        _check(self._L.rafthip_commit(self._h))

    def commit_changed(self, max_pairs=None):
        """Returns (pairs[PAIR_DTYPE], n_changed)."""
        if max_pairs is None:
            max_pairs = self.G
        out = np.empty(max_pairs, dtype=PAIR_DTYPE)
        n = C.c_uint64()
        _check(self._L.rafthip_commit_changed(self._h, _ptr(out) if max_pairs else None, max_pairs,
                                              C.byref(n)))
        return out[:min(n.value, max_pairs)], n.value

    def commit_changed_static(self, dst, max_pairs):
        """rafthip_commit_changed_static: pairs land at `dst` (device or page-locked host address, room
        for max_pairs pairs); allocates nothing.  Returns the total number of changed groups."""
        n = C.c_uint64()
        _check(self._L.rafthip_commit_changed_static(self._h, _ptr(dst) if max_pairs else None, max_pairs, C.byref(n)))
        return n.value

    def commit_report_into(self, dst, max_pairs, advance=False):
        """commit_changed / commit_advance with the pairs landing at `dst` (host or device
        address, room for max_pairs pairs).  Returns the total number of pairs."""
        n = C.c_uint64()
        fn = self._L.rafthip_commit_advance if advance else self._L.rafthip_commit_changed
        _check(fn(self._h, _ptr(dst) if max_pairs else None, max_pairs, C.byref(n)))
        return n.value

    def commit_advance(self, max_pairs=None):
        """raft.maybeCommit for every group. Returns (pairs[PAIR_DTYPE], n_advanced)."""
        if max_pairs is None:
            max_pairs = self.G
        out = np.empty(max_pairs, dtype=PAIR_DTYPE)
        n = C.c_uint64()
        _check(self._L.rafthip_commit_advance(self._h, _ptr(out) if max_pairs else None, max_pairs,
                                              C.byref(n)))
        return out[:min(n.value, max_pairs)], n.value

    def upload_term_range(self, term_first, term_last, first=0, count=None):
        first, count = self._range(first, count)
        _check(self._L.rafthip_upload_term_range(self._h, first, count, _ptr(term_first), _ptr(term_last)))

    def read_term_range(self, first=0, count=None):
        first, count = self._range(first, count)
        tf, tl = np.empty(count, dtype=np.uint64), np.empty(count, dtype=np.uint64)
        _check(self._L.rafthip_read_term_range(self._h, first, count, _ptr(tf), _ptr(tl)))
        return tf, tl

    def ingest_msgs(self, msgs, n=None):
        if n is None:
            n = len(msgs)
        _check(self._L.rafthip_ingest_msgs(self._h, _ptr(msgs), n))

    def upload_ids(self, ids, first=0, count=None):
        """ids [S][count] uint64: the raft id behind each Progress row, 0 = empty."""
        first, count = self._range(first, count)
        assert ids.dtype == np.uint64 and ids.shape == (self.S, count) and ids.flags["C_CONTIGUOUS"]
        _check(self._L.rafthip_upload_ids(self._h, first, count, _ptr(ids), count))

    def upload_term(self, term, first=0, count=None):
        first, count = self._range(first, count)
        _check(self._L.rafthip_upload_term(self._h, first, count, _ptr(term)))

    def upload_read_ctx(self, ctx, ctx_len, first=0, count=None):
        first, count = self._range(first, count)
        _check(self._L.rafthip_upload_read_ctx(self._h, first, count, _ptr(ctx), _ptr(ctx_len)))

    def upload_campaign(self, resp_type, first=0, count=None):
        """resp_type u8 per group: MSG_PRE_VOTE_RESP / MSG_VOTE_RESP / 0; None clears every group."""
        first, count = self._range(first, count)
        _check(self._L.rafthip_upload_campaign(self._h, first, count, _ptr(resp_type)))

    def ingest_wire(self, data, offsets, groups, n=None, want_msgs=True, want_status=True, want_counts=True):
        """Serialized raftpb.Message values -> (out_msgs, out_status, counts[WIRE_STATUSES]); data / offsets / groups
        are numpy arrays or raw device pointers (then pass n)."""
        if n is None:
            n = len(groups)
        msgs = np.zeros(n, dtype=MSG_DTYPE) if want_msgs else None
        status = np.zeros(n, dtype=np.uint8) if want_status else None
        counts = np.zeros(WIRE_STATUSES, dtype=np.uint64) if want_counts else None
        _check(self._L.rafthip_ingest_wire(self._h, _ptr(data), _ptr(offsets), _ptr(groups), n, _ptr(msgs),
                                           _ptr(status), _ptr(counts)))
        return msgs, status, counts

    def progress_config(self, max_inflight, max_inflight_bytes=0):
        _check(self._L.rafthip_progress_config(self._h, max_inflight, max_inflight_bytes))
        self.inflight_cap = max_inflight

    def upload_progress(self, state, next, pending=None, first=0, count=None):
        """state u8 [S][count] (STATE_* | PROGRESS_PAUSED), next / pending u64 [S][count]; empties the Inflights."""
        first, count = self._range(first, count)
        assert state.shape == (self.S, count) and next.shape == (self.S, count)
        _check(self._L.rafthip_upload_progress(self._h, first, count, _ptr(state), _ptr(next), _ptr(pending), count))

    def init_progress_row(self, group, slot, state=STATE_PROBE, next=1, pending=0):
        """Changer.initProgress for one new peer: the row alone is re-created, Inflights emptied."""
        _check(self._L.rafthip_init_progress_row(self._h, group, slot, state, next, pending))

    def upload_inflights(self, infl_meta, infl_bytes, ring_index, ring_bytes, first=0, count=None):
        first, count = self._range(first, count)
        assert ring_index.shape == (max(self.inflight_cap, 1), self.S, count) == ring_bytes.shape
        _check(self._L.rafthip_upload_inflights(self._h, first, count, _ptr(infl_meta), _ptr(infl_bytes),
                                                _ptr(ring_index), _ptr(ring_bytes), count))

    def upload_first_index(self, first_index, first=0, count=None):
        first, count = self._range(first, count)
        _check(self._L.rafthip_upload_first_index(self._h, first, count, _ptr(first_index)))

    def read_progress(self, first=0, count=None, rings=True):
        """The per-peer planes as a dict of [slot][group] arrays (rings: [entry][slot][group])."""
        first, count = self._range(first, count)
        S, cap = self.S, max(self.inflight_cap, 1)
        pl = {"pstate": np.empty((S, count), dtype=np.uint8), "next": np.empty((S, count), dtype=np.uint64),
              "pending": np.empty((S, count), dtype=np.uint64), "infl_meta": np.empty((S, count), dtype=np.uint32),
              "infl_bytes": np.empty((S, count), dtype=np.uint64)}
        _check(self._L.rafthip_read_progress(self._h, first, count, _ptr(pl["pstate"]), _ptr(pl["next"]),
                                             _ptr(pl["pending"]), _ptr(pl["infl_meta"]), _ptr(pl["infl_bytes"]), count))
        if rings:
            pl["ring_index"] = np.empty((cap, S, count), dtype=np.uint64)
            pl["ring_bytes"] = np.empty((cap, S, count), dtype=np.uint64)
            _check(self._L.rafthip_read_inflights(self._h, first, count, _ptr(pl["ring_index"]), _ptr(pl["ring_bytes"]),
                                                  count))
        return pl

    def step_progress(self, msgs, aux=None, n=None, want_actions=True):
        """Events (MSG_DTYPE records or a device pointer) -> action bytes (numpy), in stream order per Progress."""
        if n is None:
            n = len(msgs)
        actions = np.zeros(n, dtype=np.uint8) if want_actions else None
        _check(self._L.rafthip_step_progress(self._h, _ptr(msgs), _ptr(aux), n, _ptr(actions)))
        return actions

    def step_dense(self, acks=None, hb_mask=None, hb_ctx_mask=None, first=0, count=None, want_actions=True):
        """Dense tick: acks u64 [S][count] (index + 1, 0 = none), heartbeat-response slot masks [count]."""
        first, count = self._range(first, count)
        actions = np.zeros((self.S, count), dtype=np.uint8) if want_actions else None
        _check(self._L.rafthip_step_dense(self._h, first, count, _ptr(acks), count, _ptr(hb_mask), _ptr(hb_ctx_mask),
                                          _ptr(actions), count))
        return actions

    def step_wire(self, data, offsets, groups, n=None):
        """Serialized messages -> (decoded records, status, counts[WIRE_STATUSES], action bytes) through the Progress stepper."""
        if n is None:
            n = len(groups)
        msgs = np.zeros(n, dtype=MSG_DTYPE)
        status = np.zeros(n, dtype=np.uint8)
        counts = np.zeros(WIRE_STATUSES, dtype=np.uint64)
        actions = np.zeros(n, dtype=np.uint8)
        _check(self._L.rafthip_step_wire(self._h, _ptr(data), _ptr(offsets), _ptr(groups), n, _ptr(msgs), _ptr(status),
                                         _ptr(counts), _ptr(actions)))
        return msgs, status, counts, actions

    def heartbeat_commit(self, first=0, count=None, download=True):
        first, count = self._range(first, count)
        out = np.empty((self.S, count), dtype=np.uint64) if download else None
        _check(self._L.rafthip_heartbeat_commit(self._h, first, count, _ptr(out), count))
        return out

    def plan_appends(self, send_if_empty=True, first=0, count=None, want_next=True):
        """bcastAppend as a plan -> (uint8 [S][count] SEND_*, uint64 [S][count] Next or None)."""
        first, count = self._range(first, count)
        plan = np.empty((self.S, count), dtype=np.uint8)
        nxt = np.empty((self.S, count), dtype=np.uint64) if want_next else None
        _check(self._L.rafthip_plan_appends(self._h, first, count, int(bool(send_if_empty)), _ptr(plan), _ptr(nxt), count))
        return plan, nxt

    def emit_heartbeats(self, first=0, count=None, due=False):
        """bcastHeartbeat as bytes -> (uint8 [S][count][HEARTBEAT_STRIDE] zero-padded messages, uint8 [S][count] lengths).
        due: only for the groups that stepped MsgBeat in the last tick (rafthip_emit_heartbeats_due)."""
        first, count = self._range(first, count)
        out = np.empty((self.S, count, HEARTBEAT_STRIDE), dtype=np.uint8)
        lens = np.empty((self.S, count), dtype=np.uint8)
        fn = self._L.rafthip_emit_heartbeats_due if due else self._L.rafthip_emit_heartbeats
        _check(fn(self._h, first, count, _ptr(out), _ptr(lens)))
        return out, lens

    def committed_in_term(self, first=0, count=None):
        """raft.committedEntryInCurrentTerm for every group -> u8."""
        first, count = self._range(first, count)
        _check(self._L.rafthip_committed_in_term(self._h))
        out = np.empty(count, dtype=np.uint8)
        _check(self._L.rafthip_read_committed_in_term(self._h, first, count, _ptr(out)))
        return out

    def upload_self(self, bits, first=0, count=None):
        first, count = self._range(first, count)
        _check(self._L.rafthip_upload_self(self._h, first, count, _ptr(bits)))

    def check_quorum(self):
        _check(self._L.rafthip_check_quorum(self._h))

    def tick_config(self, election_tick, heartbeat_tick, check_quorum):
        _check(self._L.rafthip_tick_config(self._h, election_tick, heartbeat_tick, int(bool(check_quorum))))

    def upload_tick_state(self, election_elapsed=None, heartbeat_elapsed=None, randomized_timeout=None, role=None,
                          flags=None, first=0, count=None):
        """Any plane may be None (left alone).  uint32 timers, uint8 role / flags."""
        first, count = self._range(first, count)
        planes = []
        for a, dt in ((election_elapsed, np.uint32), (heartbeat_elapsed, np.uint32), (randomized_timeout, np.uint32),
                      (role, np.uint8), (flags, np.uint8)):
            if a is None:
                planes.append(None)
                continue
            a = np.ascontiguousarray(a, dtype=dt)
            if len(a) != count:
                raise ValueError("tick plane of %d entries for %d groups" % (len(a), count))
            planes.append(a)
        _check(self._L.rafthip_upload_tick_state(self._h, first, count, *[None if a is None else _ptr(a) for a in planes]))

    def read_tick_state(self, first=0, count=None):
        """(election_elapsed, heartbeat_elapsed, randomized_timeout, role, flags)."""
        first, count = self._range(first, count)
        outs = [np.empty(count, dtype=np.uint32) for _ in range(3)] + [np.empty(count, dtype=np.uint8) for _ in range(2)]
        _check(self._L.rafthip_read_tick_state(self._h, first, count, *[_ptr(a) for a in outs]))
        return tuple(outs)

    def tick(self, counts=True):
        """RawNode.Tick for every group.  Returns the per-action group counts (TICK_ACTIONS words), or None."""
        if not counts:
            _check(self._L.rafthip_tick(self._h, None))
            return None
        out = np.zeros(TICK_ACTIONS, dtype=np.uint64)
        _check(self._L.rafthip_tick(self._h, _ptr(out)))
        return out

    def tick_heard_from_leader(self, groups):
        """electionElapsed = 0 for the listed groups (numpy uint32 / device tensor / address + count tuple).
        Returns how many ids lay outside the engine."""
        if isinstance(groups, tuple):
            ptr, n = groups
        else:
            if isinstance(groups, np.ndarray):
                groups = np.ascontiguousarray(groups, dtype=np.uint32)
            ptr, n = groups, len(groups)
        bad = C.c_uint64()
        _check(self._L.rafthip_tick_heard_from_leader(self._h, _ptr(ptr) if n else None, n, C.byref(bad)))
        return bad.value

    def read_tick_events(self, mask, max_events=None, dst=None):
        """(events[TICK_EVENT_DTYPE] in group order, total); with dst (device / page-locked address) only the total."""
        if max_events is None:
            max_events = self.G
        n = C.c_uint64()
        if dst is not None:
            _check(self._L.rafthip_read_tick_events(self._h, mask, _ptr(dst) if max_events else None, max_events, C.byref(n)))
            return n.value
        out = np.empty(max_events, dtype=TICK_EVENT_DTYPE)
        _check(self._L.rafthip_read_tick_events(self._h, mask, _ptr(out) if max_events else None, max_events, C.byref(n)))
        return out[:min(n.value, max_events)], n.value

    def read_tick_actions(self, first=0, count=None):
        first, count = self._range(first, count)
        out = np.empty(count, dtype=np.uint8)
        _check(self._L.rafthip_read_tick_actions(self._h, first, count, _ptr(out)))
        return out

    def read_active(self, first=0, count=None):
        first, count = self._range(first, count)
        out = np.empty(count, dtype=self.mask_dtype)
        _check(self._L.rafthip_read_active(self._h, first, count, _ptr(out)))
        return out

    def read_acks(self, first=0, count=None):
        first, count = self._range(first, count)
        out = np.empty(count, dtype=self.mask_dtype)
        _check(self._L.rafthip_read_acks(self._h, first, count, _ptr(out)))
        return out

    def read_commit(self, first=0, count=None):
        first, count = self._range(first, count)
        out = np.empty(count, dtype=np.uint64)
        _check(self._L.rafthip_read_commit(self._h, first, count, _ptr(out)))
        return out

    def tally_votes(self):
        _check(self._L.rafthip_tally_votes(self._h))

    def read_vote_result(self, first=0, count=None, counts=True):
        first, count = self._range(first, count)
        res = np.empty(count, dtype=np.uint8)
        gr = np.empty(count, dtype=np.uint8) if counts else None
        rj = np.empty(count, dtype=np.uint8) if counts else None
        _check(self._L.rafthip_read_vote_result(self._h, first, count, _ptr(res), _ptr(gr), _ptr(rj)))
        return (res, gr, rj) if counts else res

    def quorum_active(self):
        _check(self._L.rafthip_quorum_active(self._h))

    def read_quorum_active(self, first=0, count=None):
        first, count = self._range(first, count)
        out = np.empty(count, dtype=np.uint8)
        _check(self._L.rafthip_read_quorum_active(self._h, first, count, _ptr(out)))
        return out

    def ack_quorum(self):
        _check(self._L.rafthip_ack_quorum(self._h))

    def read_ack_quorum(self, first=0, count=None):
        first, count = self._range(first, count)
        out = np.empty(count, dtype=np.uint8)
        _check(self._L.rafthip_read_ack_quorum(self._h, first, count, _ptr(out)))
        return out

    def synchronize(self):
        _check(self._L.rafthip_synchronize(self._h))

    def stats(self):
        s = Stats()
        _check(self._L.rafthip_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in Stats._fields_}

    def plane(self, which):
        """(device address, stride in elements) of an engine plane."""
        p, st = C.c_void_p(), C.c_uint64()
        _check(self._L.rafthip_plane(self._h, which, C.byref(p), C.byref(st)))
        return p.value, st.value


class EngineSet:
    """A fixed list of engines for rafthip_commit_many (the handle array is built once)."""

    def __init__(self, engines):
        self.engines = list(engines)
        self._L = load_library()
        self._arr = (C.c_void_p * len(self.engines))(*[e._h for e in self.engines])

    def commit(self, n=None):
        """One launch over the first n engines (default: all)."""
        _check(self._L.rafthip_commit_many(self._arr, len(self.engines) if n is None else n))

    def tick(self, counts=True, n=None):
        """rafthip_tick_many: RawNode.Tick for every group of the first n engines (default: all) in one launch.
        Returns the per-action counts, one row of TICK_ACTIONS per engine, or None."""
        n = len(self.engines) if n is None else n
        if not counts:
            _check(self._L.rafthip_tick_many(self._arr, n, None))
            return None
        out = np.zeros((n, TICK_ACTIONS), dtype=np.uint64)
        _check(self._L.rafthip_tick_many(self._arr, n, _ptr(out)))
        return out

    def commit_report(self, advance=False, max_pairs=None, dsts=None):
        """rafthip_commit_report_many: commit_changed / commit_advance of every engine in one call.  Returns
        (list of numpy PAIR_DTYPE arrays — or None when `dsts`, device / host addresses with room for max_pairs[i]
        pairs each, were given —, counts)."""
        n = len(self.engines)
        if max_pairs is None:
            max_pairs = [e.G for e in self.engines]
        elif np.isscalar(max_pairs):
            max_pairs = [int(max_pairs)] * n
        outs = None
        if dsts is None:
            outs = [np.empty(m, dtype=PAIR_DTYPE) for m in max_pairs]
            dsts = [o.ctypes.data if m else None for o, m in zip(outs, max_pairs)]
        darr = (C.c_void_p * n)(*dsts)
        marr = (C.c_uint64 * n)(*max_pairs)
        counts = (C.c_uint64 * n)()
        _check(self._L.rafthip_commit_report_many(self._arr, n, int(bool(advance)), darr, marr, counts))
        counts = [int(c) for c in counts]
        if outs is not None:
            outs = [o[:min(c, m)] for o, c, m in zip(outs, counts, max_pairs)]
        return outs, counts


def comm_unique_id():
    """rafthip_comm_unique_id -> 128 bytes to hand to every rank (rank 0 calls this)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load_library().rafthip_comm_unique_id(buf))
    return buf.raw


class Comm:
    """One rank of the path's exchange step (rafthip_comm_*): binds an Engine to a rank of an RCCL
    communicator.  Creation is collective over the ranks."""

    def __init__(self, engine, comm_id, rank, nranks, group_base=0, max_pairs_per_rank=0):
        self._h = None
        self._L = load_library()
        self.engine = engine
        idbuf = C.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        h = C.c_void_p()
        _check(self._L.rafthip_comm_create(engine._h, idbuf, rank, nranks, group_base, max_pairs_per_rank, C.byref(h)))
        self._h = h
        r, n, cap = C.c_int(), C.c_int(), C.c_uint64()
        _check(self._L.rafthip_comm_info(h, C.byref(r), C.byref(n), C.byref(cap)))
        self.rank, self.nranks, self.max_pairs_per_rank = r.value, n.value, cap.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.rafthip_comm_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def commit_allgather(self, advance=False, max_pairs=None, dst=None):
        """Commit pass + all-gather.  Returns (pairs of the whole node as numpy PAIR_DTYPE sorted by global
        group id — or None when `dst`, a device / host address with room for max_pairs pairs, was given —,
        per-rank counts, total)."""
        if max_pairs is None:
            max_pairs = self.nranks * self.max_pairs_per_rank
        counts = np.zeros(self.nranks, dtype=np.uint64)
        total = C.c_uint64()
        out = None
        if dst is None:
            out = np.empty(max_pairs, dtype=PAIR_DTYPE)
            dst = out
        _check(self._L.rafthip_commit_allgather(self._h, int(bool(advance)), _ptr(dst) if max_pairs else None,
                                                max_pairs, _ptr(counts), C.byref(total)))
        if out is not None:
            got = int(np.minimum(counts, self.max_pairs_per_rank).sum())
            out = out[:min(got, max_pairs)]
        return out, counts, total.value


    def dense_width(self):
        """Elements per rank in the dense exchange: the largest commit-plane stride of any rank."""
        width = C.c_uint64()
        _check(self._L.rafthip_commit_allgather_dense(self._h, 0, None, 0, C.byref(width)))
        return width.value

    def commit_allgather_dense(self, advance=False, dst=None, dst_elems=None):
        """Commit pass + all-gather of the commit planes.  Returns (planes as a numpy uint64 [nranks][width] array —
        or None when `dst`, a device / host address with room for dst_elems elements, was given —, width)."""
        width = C.c_uint64()
        out = None
        if dst is None:
            dst_elems = self.nranks * self.dense_width()
            out = np.empty(dst_elems, dtype=np.uint64)
            dst = out
        _check(self._L.rafthip_commit_allgather_dense(self._h, int(bool(advance)), _ptr(dst), int(dst_elems), C.byref(width)))
        if out is not None:
            out = out.reshape(self.nranks, width.value)
        return out, width.value


class CommSet:
    """All n ranks of one communicator driven by one thread (rafthip_commit_allgather_many /
    _dense_many).  `CommSet.loopback(engines, group_bases)` builds the ranks on ONE GPU through the
    library's internal loopback transport (raft_amd/csrc/rafthip_internal.h — a test seam, not part of
    include/rafthip.h): the exchange code is the shipped one, only the gather is device copies."""

    def __init__(self, handles, engines):
        self._L = load_library()
        self._hs = list(handles)
        self.engines = list(engines)
        self.n = len(self._hs)
        self._arr = (C.c_void_p * self.n)(*self._hs)
        r, n, cap = C.c_int(), C.c_int(), C.c_uint64()
        self.caps = []
        for h in self._hs:
            _check(self._L.rafthip_comm_info(h, C.byref(r), C.byref(n), C.byref(cap)))
            assert n.value == self.n
            self.caps.append(cap.value)
        self.max_pairs_per_rank = self.caps[0]

    @classmethod
    def loopback(cls, engines, group_bases, max_pairs_per_rank=None):
        L = load_library()
        fn = L.rafthip_internal_comm_create_loopback
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p)]
        n = len(engines)
        earr = (C.c_void_p * n)(*[e._h for e in engines])
        barr = (C.c_uint64 * n)(*[int(b) for b in group_bases])
        marr = None if max_pairs_per_rank is None else (C.c_uint64 * n)(*[int(m) for m in max_pairs_per_rank])
        out = (C.c_void_p * n)()
        _check(fn(earr, n, barr, marr, out))
        return cls([C.c_void_p(h) for h in out], engines)

    def close(self):
        for h in getattr(self, "_hs", []):
            self._L.rafthip_comm_destroy(h)
        self._hs = []

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def commit_allgather(self, advance=False, max_pairs=None, dsts=None):
        """The exchange on every rank.  Returns (per rank: gathered pairs as numpy PAIR_DTYPE — or None when
        `dsts` (addresses) were given —, per rank: counts[nranks], per rank: total)."""
        n = self.n
        if max_pairs is None:
            max_pairs = [n * self.max_pairs_per_rank] * n
        elif np.isscalar(max_pairs):
            max_pairs = [int(max_pairs)] * n
        outs = None
        if dsts is None:
            outs = [np.empty(m, dtype=PAIR_DTYPE) for m in max_pairs]
            dsts = [o.ctypes.data if m else None for o, m in zip(outs, max_pairs)]
        counts = [np.zeros(n, dtype=np.uint64) for _ in range(n)]
        darr = (C.c_void_p * n)(*dsts)
        marr = (C.c_uint64 * n)(*max_pairs)
        carr = (C.c_void_p * n)(*[c.ctypes.data for c in counts])
        totals = (C.c_uint64 * n)()
        _check(self._L.rafthip_commit_allgather_many(self._arr, n, int(bool(advance)), darr, marr, carr, totals))
        if outs is not None:
            outs = [o[:min(int(np.minimum(c, self.max_pairs_per_rank).sum()), m)] for o, c, m in zip(outs, counts, max_pairs)]
        return outs, counts, [int(t) for t in totals]

    def dense_width(self):
        w = (C.c_uint64 * self.n)()
        _check(self._L.rafthip_commit_allgather_dense_many(self._arr, self.n, 0, None, None, w))
        return int(w[0])

    def commit_allgather_dense(self, advance=False, dsts=None, dst_elems=None):
        """The dense exchange on every rank.  Returns (per rank: planes [nranks][width] or None with `dsts`, width)."""
        n = self.n
        width = self.dense_width()
        outs = None
        if dsts is None:
            outs = [np.empty(n * width, dtype=np.uint64) for _ in range(n)]
            dsts = [o.ctypes.data for o in outs]
            dst_elems = [n * width] * n
        darr = (C.c_void_p * n)(*dsts)
        earr = (C.c_uint64 * n)(*dst_elems)
        w = (C.c_uint64 * n)()
        _check(self._L.rafthip_commit_allgather_dense_many(self._arr, n, int(bool(advance)), darr, earr, w))
        assert all(int(x) == width for x in w)
        if outs is not None:
            outs = [o.reshape(n, width) for o in outs]
        return outs, width


def batch_committed_index(match, in_mask, out_mask=None, device=0, out=None):
    """quorum.BatchCommittedIndex: match u64 [S][G], slot masks [G] → u64 [G] (`out`: a result array to reuse)."""
    S, G = match.shape
    if out is None:
        out = np.empty(G, dtype=np.uint64)
    elif out.dtype != np.uint64 or out.shape != (G,) or not out.flags.c_contiguous:
        raise ValueError("out must be a contiguous uint64 array of %d elements" % G)
    _check(load_library().rafthip_batch_committed_index(device, G, S, _ptr(match), _ptr(in_mask),
                                                        _ptr(out_mask), _ptr(out)))
    return out


def batch_vote_result(votes, in_mask, out_mask=None, device=0):
    """quorum.BatchVoteResult: vote cells u8 [S][G], slot masks [G] → VoteResult u8 [G]."""
    S, G = votes.shape
    out = np.empty(G, dtype=np.uint8)
    _check(load_library().rafthip_batch_vote_result(device, G, S, _ptr(votes), _ptr(in_mask),
                                                    _ptr(out_mask), _ptr(out)))
    return out
