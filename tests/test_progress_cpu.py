"""CPU tests of the Progress / Inflights oracle (oracle/progress_oracle.c) against the
reference's own tables and scenarios, tests/golden/progress.json."""
import numpy as np
import pytest

import progress_util as pu
from oracle import pyoracle as po

FX = pu.load_fixture()
ST = po.STATE_NAMES


def test_is_paused():
    for r in FX["is_paused"]["rows"]:
        assert po.Progress(ST[r["state"]], paused=r["paused"]).get()["is_paused"] == r["w"], r


def test_resume():
    fx = FX["resume"]
    p = po.Progress(next=fx["next"], paused=fx["paused"])
    p.maybe_decr_to(1, 1)
    assert p.get()["paused"] is False
    p = po.Progress(next=p.get()["next"], paused=True)
    p.maybe_update(2)
    assert p.get()["paused"] is False


def test_become():
    for r in FX["become_probe"]["rows"]:
        p = po.Progress(ST[r["state"]], r["match"], r["next"], r["pending"])
        p.become(po.STATE_PROBE)
        got = p.get()
        assert (got["state"], got["match"], got["next"]) == (po.STATE_PROBE, r["match"], r["wnext"]), r
    r = FX["become_replicate"]
    p = po.Progress(ST[r["state"]], r["match"], r["next"])
    p.become(po.STATE_REPLICATE)
    assert (p.get()["state"], p.get()["match"], p.get()["next"]) == (po.STATE_REPLICATE, r["match"], r["wnext"])
    r = FX["become_snapshot"]
    p = po.Progress(ST[r["state"]], r["match"], r["next"])
    p.become(po.STATE_SNAPSHOT, r["snapshoti"])
    assert (p.get()["state"], p.get()["match"], p.get()["pending"]) == (po.STATE_SNAPSHOT, r["match"], r["wpending"])


def test_maybe_update_and_decr():
    fx = FX["maybe_update"]
    for r in fx["rows"]:
        p = po.Progress(match=fx["prev_match"], next=fx["prev_next"])
        assert p.maybe_update(r["update"]) == r["wok"]
        assert (p.get()["match"], p.get()["next"]) == (r["wm"], r["wn"]), r
    for r in FX["maybe_decr"]["rows"]:
        p = po.Progress(ST[r["state"]], r["m"], r["n"])
        assert p.maybe_decr_to(r["rejected"], r["last"]) == r["w"], r
        assert (p.get()["match"], p.get()["next"]) == (r["m"], r["wn"]), r


def _check_ring(p, want):
    got = p.inflights()
    for k in ("start", "count", "bytes", "index", "sizes"):
        assert got[k] == want[k], (k, got, want)


def test_inflights_add_and_free():
    for case in FX["inflights_add"]["cases"]:
        p = po.Progress(inflight_size=case["size"])
        p.set_start(case["start"])
        for st in case["steps"]:
            for i in range(*st["add"]):
                assert p.add(i, 100 + i) == 0
            _check_ring(p, st["want"])
    fx = FX["inflights_free_to"]
    p = po.Progress(inflight_size=fx["size"])
    for st in fx["steps"]:
        if "add" in st:
            for i in range(*st["add"]):
                assert p.add(i, 100 + i) == 0
        else:
            p.free_le(st["free_le"])
            _check_ring(p, st["want"])


def test_inflights_full_and_reset():
    for r in FX["inflights_full"]["rows"]:
        p = po.Progress(inflight_size=r["size"], inflight_max_bytes=r["maxBytes"])

        def add_until_full(begin, end):
            for i in range(begin, end):
                assert not p.inflights()["full"], (r, i)
                assert p.add(i, 100 + i) == 0
            assert p.inflights()["full"], (r, end)

        add_until_full(0, r["fullAt"])
        p.free_le(r["freeLE"])
        add_until_full(r["fullAt"], r["againAt"])
        assert p.add(100, 1024) == -1  # the reference panics
    fx = FX["inflights_reset"]
    p = po.Progress(inflight_size=fx["size"], inflight_max_bytes=fx["maxBytes"])
    index = 0
    for _ in range(fx["epochs"]):
        p.reset_inflights()
        for _ in range(fx["adds_per_epoch"]):
            assert not p.inflights()["full"]
            index += 1
            p.add(index, fx["add_bytes"])
        p.free_le(index - fx["keep"])
        assert not p.inflights()["full"] and p.inflights()["count"] == fx["keep"]
    p.free_le(index)
    assert p.inflights()["count"] == 0


@pytest.mark.parametrize("k", range(len(FX["scenarios"])))
def test_scenarios_step_by_step_and_in_one_batch(k):
    sc = FX["scenarios"][k]
    steps = pu.scenario_steps(sc)
    cap, mb = sc["cap"], sc["max_bytes"]
    pl = pu.init_planes(sc)
    active, acks = pu.mask_planes(1, 1)
    last = np.array([sc["last_index"]], dtype=np.uint64)
    first = np.array([sc["first_index"]], dtype=np.uint64)
    recs, auxs = [], []
    def plan(send_if_empty=True):
        # the scenario's Progress in slot 0 of a two-slot group led from slot 1
        two = {k: np.concatenate([pl[k], np.zeros_like(pl[k])]) for k in ("next", "pstate", "infl_meta", "infl_bytes")}
        return int(po.plan_appends(two, cap, mb, None, np.array([2], dtype=np.uint8), active, first, last, send_if_empty)[0, 0])

    for i, st in enumerate(steps):
        want = pu.sent_plan(st, sc["src"])
        if want is not None:  # the reference sent exactly this from the state the scenario is in
            assert plan() == want, (sc["src"], i, plan(), want)
            if want == po.SEND_EMPTY:
                assert plan(send_if_empty=False) == po.SEND_NONE
        m, aux = pu.event(st)
        recs.append(m)
        auxs.append(aux)
        act = po.step_progress(pl, cap, mb, active, acks, last, first, m, aux)
        pu.check_actions(st, int(act[0]), (sc["src"], i))
        got = pu.cell_state(pl, cap, mb)
        pu.check_expect(st, got, (sc["src"], i))
        if got["is_paused"]:
            assert plan() == po.SEND_NONE, (sc["src"], i)  # maybeSendAppend returns at :602-604
    # the same events as ONE batch end in the same state (stream order per Progress), while
    # the appended records of the batch apply first
    if len(steps) <= po.STEP_LIMIT and not any(st["ev"] == "appended" for st in steps):
        pl2 = pu.init_planes(sc)
        a2, k2 = pu.mask_planes(1, 1)
        last2 = np.array([sc["last_index"]], dtype=np.uint64)
        po.step_progress(pl2, cap, mb, a2, k2, last2, first, np.concatenate(recs), np.concatenate(auxs))
        for key in pl:
            assert np.array_equal(pl[key], pl2[key]), key
        assert np.array_equal(active, a2)


def test_batch_equals_one_by_one_on_random_streams():
    """A batch = its events stepped one at a time in stream order, whenever no Progress is
    named more than STEP_LIMIT times and the appended records come first."""
    rng = np.random.default_rng(11)
    S, G, cap, mb = 5, 300, 4, 150
    pl = pu.random_planes(rng, S, G, cap, mb)
    msgs, aux = pu.random_events(rng, 6000, S, G, pl, hot_cells=20)
    order = np.argsort(msgs["type"] != po.EV_APPENDED, kind="stable")  # appended first, rest in stream order
    msgs, aux = msgs[order], aux[order]
    last = rng.integers(0, 1 << 20, size=G).astype(np.uint64)
    first = rng.integers(0, 1 << 19, size=G).astype(np.uint64)
    a1, k1 = pu.mask_planes(S, G)
    pl1 = {k: v.copy() for k, v in pl.items()}
    l1 = last.copy()
    act1 = po.step_progress(pl1, cap, mb, a1, k1, l1, first, msgs, aux, limit=1 << 30)
    a2, k2 = pu.mask_planes(S, G)
    pl2 = {k: v.copy() for k, v in pl.items()}
    l2 = last.copy()
    act2 = np.concatenate([po.step_progress(pl2, cap, mb, a2, k2, l2, first, msgs[i:i + 1], aux[i:i + 1])
                           for i in range(len(msgs))])
    assert np.array_equal(act1, act2)
    for key in pl:
        assert np.array_equal(pl1[key], pl2[key]), key
    assert np.array_equal(a1, a2) and np.array_equal(k1, k2) and np.array_equal(l1, l2)
    seen = np.bincount(act1, minlength=256)
    for bit in (po.ACT_SEND_APPEND, po.ACT_UPDATED, po.ACT_UPDATED | po.ACT_WAS_PAUSED, po.ACT_ERROR, po.ACT_DROPPED):
        assert seen[bit] > 0, bit
    assert len(set(pl1["pstate"].ravel().tolist())) >= 5


def test_deferral_is_all_or_nothing_per_progress():
    rng = np.random.default_rng(12)
    S, G, cap = 2, 3, 4
    pl = pu.random_planes(rng, S, G, cap, 0)
    before = {k: v.copy() for k, v in pl.items()}
    n = 70
    msgs = np.zeros(n, dtype=pu.MSG_DTYPE)
    msgs["type"] = po.EV_HEARTBEAT_RESP
    msgs["group"][:66] = 1  # 66 events on Progress (slot 0, group 1): over the limit of 64
    msgs["group"][66:] = 2
    a, k = pu.mask_planes(S, G)
    act = po.step_progress(pl, cap, 0, a, k, np.zeros(G, dtype=np.uint64), np.zeros(G, dtype=np.uint64), msgs)
    assert (act[:66] == po.ACT_DEFERRED).all() and (act[66:] & po.ACT_DEFERRED == 0).all()
    assert pl["pstate"][0, 1] == before["pstate"][0, 1] and int(a[1]) == 0 and int(a[2]) == 1


def test_a_host_only_event_blocks_the_rest_of_its_progress():
    rng = np.random.default_rng(13)
    S, G, cap = 2, 4, 4
    pl = pu.random_planes(rng, S, G, cap, 0)
    before = {k: v.copy() for k, v in pl.items()}
    msgs = np.zeros(6, dtype=pu.MSG_DTYPE)
    msgs["type"] = [po.EV_HEARTBEAT_RESP, po.EV_NEEDS_HOST, po.EV_HEARTBEAT_RESP, po.EV_HEARTBEAT_RESP, po.EV_UNREACHABLE,
                    po.EV_HEARTBEAT_RESP]
    msgs["group"] = [1, 1, 1, 2, 1, 1]
    msgs["slot"] = [0, 0, 0, 0, 1, 0]
    a, k = pu.mask_planes(S, G)
    act = po.step_progress(pl, cap, 0, a, k, np.zeros(G, dtype=np.uint64), np.zeros(G, dtype=np.uint64), msgs)
    D = po.ACT_DEFERRED
    assert [bool(x & D) for x in act] == [False, True, True, False, False, True]
    assert (pl["pstate"][0, 1] & 4) == 0  # the first heartbeat response was stepped
    assert pl["next"][0, 1] == before["next"][0, 1]


def test_shared_numbers_agree():
    from raft_amd import capi
    for name in ("STATE_PROBE", "STATE_REPLICATE", "STATE_SNAPSHOT", "ACT_SEND_APPEND", "ACT_UPDATED", "ACT_WAS_PAUSED",
                 "ACT_CAUGHT_UP", "ACT_ERROR", "ACT_DEFERRED", "ACT_DROPPED", "STEP_LIMIT"):
        assert getattr(pu.po, name) == getattr(po, name) == getattr(capi, name), name
    for name in ("EV_APP_RESP", "EV_HEARTBEAT_RESP", "EV_UNREACHABLE", "EV_SNAP_STATUS", "EV_APPENDED", "EV_SENT_APP",
                 "EV_SENT_SNAP", "EV_NEEDS_HOST", "STATE_NAMES"):
        assert getattr(pu.po, name) == getattr(po, name), name
    assert pu.MSG_DTYPE == po.MSG_DTYPE == capi.MSG_DTYPE
    assert (capi.MSG_UNREACHABLE, capi.MSG_SNAP_STATUS, capi.MSG_SENT_APP, capi.MSG_SENT_SNAP, capi.MSG_NEEDS_HOST) == (
        po.EV_UNREACHABLE, po.EV_SNAP_STATUS, po.EV_SENT_APP, po.EV_SENT_SNAP, po.EV_NEEDS_HOST)
