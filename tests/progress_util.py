"""Shared helpers of the Progress / Inflights tests: the
fixture tests/golden/progress.json as event records, random event streams, and
a runner that drives any `step(msgs, aux) -> actions` + `state() -> dict`
pair (the oracle on the CPU, the engine on the GPU) through a scenario."""
import json
import os

import numpy as np

import golden_util as gu

HERE = os.path.dirname(os.path.abspath(__file__))
MSG_DTYPE = np.dtype([("group", "<u4"), ("slot", "u1"), ("type", "u1"), ("reject", "u1"), ("flags", "u1"),
                      ("index", "<u8")])


class po:  # the event / action / state numbers (= oracle.pyoracle's and raft_amd.capi's; tests/test_progress_cpu.py
    # checks both) — kept here so that the generators below load neither the oracle nor the engine
    STATE_PROBE, STATE_REPLICATE, STATE_SNAPSHOT = 0, 1, 2
    STATE_NAMES = {"StateProbe": 0, "StateReplicate": 1, "StateSnapshot": 2}
    EV_APP_RESP, EV_HEARTBEAT_RESP, EV_UNREACHABLE, EV_SNAP_STATUS = 4, 9, 10, 11
    EV_APPENDED, EV_SENT_APP, EV_SENT_SNAP, EV_NEEDS_HOST = 0xF0, 0xF1, 0xF2, 0xFE
    ACT_SEND_APPEND, ACT_UPDATED, ACT_WAS_PAUSED, ACT_CAUGHT_UP, ACT_ERROR, ACT_DEFERRED, ACT_DROPPED = 1, 2, 4, 8, 32, 64, 128
    STEP_LIMIT = 64
    SEND_NONE, SEND_APPEND, SEND_EMPTY, SEND_SNAPSHOT = 0, 1, 2, 3

    @staticmethod
    def progress_planes(S, G, cap):
        return {"match": np.zeros((S, G), dtype=np.uint64), "next": np.zeros((S, G), dtype=np.uint64),
                "pstate": np.zeros((S, G), dtype=np.uint8), "pending": np.zeros((S, G), dtype=np.uint64),
                "infl_meta": np.zeros((S, G), dtype=np.uint32), "infl_bytes": np.zeros((S, G), dtype=np.uint64),
                "ring_index": np.zeros((max(cap, 1), S, G), dtype=np.uint64),
                "ring_bytes": np.zeros((max(cap, 1), S, G), dtype=np.uint64)}

EV = {"app_resp": po.EV_APP_RESP, "heartbeat_resp": po.EV_HEARTBEAT_RESP, "unreachable": po.EV_UNREACHABLE,
      "snap_status": po.EV_SNAP_STATUS, "appended": po.EV_APPENDED, "sent_app": po.EV_SENT_APP,
      "sent_snap": po.EV_SENT_SNAP}
ACT = {"send_append": po.ACT_SEND_APPEND, "updated": po.ACT_UPDATED, "was_paused": po.ACT_WAS_PAUSED,
       "caught_up": po.ACT_CAUGHT_UP, "error": po.ACT_ERROR, "deferred": po.ACT_DEFERRED, "dropped": po.ACT_DROPPED}


def load_fixture():
    """The generated tables (progress_tables.json, lifted from the reference's Go tests by make_golden.py) and
    the hand-reconstructed scenarios (progress.json) as one dict."""
    with open(os.path.join(HERE, "golden", "progress_tables.json")) as fh:
        fx = json.load(fh)
    with open(os.path.join(HERE, "golden", "progress.json")) as fh:
        fx["scenarios"] = json.load(fh)["scenarios"]
    return fx


def event(step, group=0, slot=0):
    """One fixture step -> (record, aux)."""
    m = np.zeros(1, dtype=MSG_DTYPE)
    m["group"], m["slot"], m["type"] = group, slot, EV[step["ev"]]
    m["reject"] = bool(step.get("reject", False))
    m["index"] = step.get("index", 0)
    aux = 0
    if step["ev"] == "sent_app":
        m["flags"] = 1 if step["entries"] else 0
        aux = step.get("bytes", 0)
    elif step.get("reject"):
        aux = step.get("hint", 0)
    return m, np.array([aux], dtype=np.uint64)


def generated_steps(kind, cap):
    """The loops of raft_flow_control_test.go:63-104 / :108-152 as fixture steps."""
    steps = []
    paused = {"is_paused": True}
    for k in range(cap):  # fill the window: the first MsgApp carries entries 1..2, then one entry each
        steps.append({"ev": "sent_app", "entries": True, "index": 2 + k, "bytes": 8})
    steps[-1]["expect"] = dict(paused, inflight=cap, full=True)
    if kind == "move_forward":
        for tt in range(2, cap):
            steps.append({"ev": "app_resp", "index": tt, "expect": {"is_paused": False, "inflight": cap - 1}})
            steps.append({"ev": "sent_app", "entries": True, "index": cap + tt, "bytes": 8, "expect": dict(paused, inflight=cap)})
            for i in range(tt):
                steps.append({"ev": "app_resp", "index": i, "actions": [], "expect": dict(paused, inflight=cap)})
    else:
        for tt in range(1, 5):
            for _ in range(tt):
                steps.append({"ev": "heartbeat_resp", "actions": ["send_append"], "expect": {"paused": False}})
                steps.append({"ev": "sent_app", "entries": False, "bytes": 0, "expect": dict(paused, inflight=cap)})
            steps.append({"ev": "heartbeat_resp", "actions": ["send_append"]})
            steps.append({"ev": "sent_app", "entries": False, "bytes": 0, "expect": dict(paused, inflight=cap)})
    return steps


def scenario_steps(sc):
    return generated_steps(sc["generate"], sc["cap"]) if "generate" in sc else sc["steps"]


def init_planes(sc, S=1, G=1, slot=0, group=0):
    pl = po.progress_planes(S, G, sc["cap"])
    init = sc["init"]
    pl["match"][slot, group] = init["match"]
    pl["next"][slot, group] = init["next"]
    pl["pstate"][slot, group] = po.STATE_NAMES[init["state"]] | (4 if init.get("paused") else 0)
    pl["pending"][slot, group] = init.get("pending", 0)
    return pl


def cell_state(pl, cap, max_bytes, slot=0, group=0):
    ps = int(pl["pstate"][slot, group])
    state, paused = ps & 3, bool(ps & 4)
    count = int(pl["infl_meta"][slot, group]) >> 16
    nbytes = int(pl["infl_bytes"][slot, group])
    return {"state": state, "match": int(pl["match"][slot, group]), "next": int(pl["next"][slot, group]),
            "pending": int(pl["pending"][slot, group]), "paused": paused,
            "is_paused": True if state == po.STATE_SNAPSHOT else paused, "inflight": count,
            "full": count == cap or (max_bytes != 0 and nbytes >= max_bytes)}


def check_expect(step, got, where):
    for k, v in step.get("expect", {}).items():
        want = po.STATE_NAMES[v] if k == "state" else v
        assert got[k] == want, (where, k, got[k], want, got)


def check_actions(step, act, where):
    if "actions" in step:
        want = 0
        for a in step["actions"]:
            want |= ACT[a]
        assert act == want, (where, act, want)


def sent_plan(step, src):
    """A send record of a fixture scenario is a send the reference performed (maybeSendAppend, raft.go:600-666):
    the plan computed from the state just before it must name that send.  None = no statement (other events;
    raft_snap_test.go calls BecomeSnapshot by hand, :76,:98,:121, without going through maybeSendAppend)."""
    if step["ev"] == "sent_app":
        return po.SEND_APPEND if step["entries"] else po.SEND_EMPTY
    if step["ev"] == "sent_snap" and not src.startswith("raft_snap_test.go"):
        return po.SEND_SNAPSHOT
    return None


def random_planes(rng, S, G, cap, max_bytes):
    """Consistent random Progress state: Next > Match, Inflights rings holding ascending indexes > Match."""
    pl = po.progress_planes(S, G, cap)
    pl["match"][:] = rng.integers(0, 1 << 20, size=(S, G), dtype=np.uint64)
    pl["match"][rng.random((S, G)) < 0.1] = 0
    state = rng.choice(np.array([0, 1, 2], dtype=np.uint8), size=(S, G), p=[0.3, 0.6, 0.1])
    paused = (rng.random((S, G)) < 0.3).astype(np.uint8)
    pl["pstate"][:] = state | (paused << 2)
    pl["next"][:] = pl["match"] + np.uint64(1) + rng.integers(0, 6, size=(S, G)).astype(np.uint64)
    pl["pending"][:] = np.where(state == 2, pl["match"] + rng.integers(0, 9, size=(S, G)).astype(np.uint64), 0)
    if cap > 0:
        count = np.where(state == 1, rng.integers(0, cap + 1, size=(S, G)), 0)
        start = np.where(count > 0, rng.integers(0, cap, size=(S, G)), 0)
        pl["infl_meta"][:] = (start | (count << 16)).astype(np.uint32)
        base = pl["match"] + np.uint64(1)
        for k in range(cap):  # logical position k of the ring sits at (start + k) % cap
            pos = (start + k) % cap
            live = k < count
            idx = base + np.uint64(2 * k) + rng.integers(0, 2, size=(S, G)).astype(np.uint64)
            by = rng.integers(0, 64, size=(S, G)).astype(np.uint64)
            ss, gg = np.nonzero(live)
            pl["ring_index"][pos[ss, gg], ss, gg] = idx[ss, gg]
            pl["ring_bytes"][pos[ss, gg], ss, gg] = by[ss, gg]
            pl["infl_bytes"][ss, gg] += by[ss, gg]
        pl["next"][:] = np.where(count > 0, base + np.uint64(2 * cap + 2), pl["next"])
    return pl


def random_events(rng, n, S, G, pl, hot_cells=0):
    """n events over random cells: mostly responses near each cell's Match / Next, some sends,
    snapshot traffic, rejections, unknown types and out-of-range cells.  hot_cells > 0 aims a
    share of the stream at that many cells so that several events meet on one Progress."""
    msgs = np.zeros(n, dtype=MSG_DTYPE)
    g = rng.integers(0, G, size=n)
    s = rng.integers(0, S, size=n)
    if hot_cells:
        hot = rng.random(n) < 0.3
        hg, hs = rng.integers(0, G, size=hot_cells), rng.integers(0, S, size=hot_cells)
        pick = rng.integers(0, hot_cells, size=n)
        g = np.where(hot, hg[pick], g)
        s = np.where(hot, hs[pick], s)
    types = np.array([po.EV_APP_RESP, po.EV_HEARTBEAT_RESP, po.EV_UNREACHABLE, po.EV_SNAP_STATUS, po.EV_APPENDED,
                      po.EV_SENT_APP, po.EV_SENT_SNAP, 3, 0xEE], dtype=np.uint8)
    msgs["type"] = rng.choice(types, size=n, p=[0.42, 0.15, 0.03, 0.04, 0.03, 0.25, 0.03, 0.03, 0.02])
    msgs["group"], msgs["slot"] = g, s
    match, nxt = pl["match"][s, g], pl["next"][s, g]
    near = np.where(rng.random(n) < 0.5, match, nxt) + rng.integers(0, 12, size=n).astype(np.uint64)
    near = near - np.minimum(near, rng.integers(0, 4, size=n).astype(np.uint64))
    msgs["index"] = near
    msgs["reject"] = rng.random(n) < 0.15
    rej = (msgs["type"] == po.EV_APP_RESP) & (msgs["reject"] != 0)
    # a rejection names Next-1 (probe) or something above Match (replicate) most of the time
    msgs["index"] = np.where(rej & (rng.random(n) < 0.7), nxt - np.uint64(1), msgs["index"])
    msgs["flags"] = rng.random(n) < 0.6
    aux = rng.integers(0, 200, size=n).astype(np.uint64)
    aux = np.where(rej, np.minimum(match + rng.integers(0, 5, size=n).astype(np.uint64), msgs["index"]), aux)
    bad = rng.random(n) < 0.02
    msgs["group"] = np.where(bad & (rng.random(n) < 0.5), G + rng.integers(0, 3, size=n), msgs["group"])
    msgs["slot"] = np.where(bad & (msgs["group"] < G), min(S, 255), msgs["slot"])
    return msgs, np.ascontiguousarray(aux)


def mask_planes(S, G):
    mdt = gu.mask_dtype(S)
    return np.zeros(G, dtype=mdt), np.zeros(G, dtype=mdt)
