"""GPU parity for the per-peer replication state:
rafthip_step_progress = tracker.Progress + tracker.Inflights driven by
stepLeader's branches (raft.go:1333-1586), in stream order per Progress.
Bit-exact vs oracle/progress_oracle.c, which tests/test_progress_cpu.py pins to
the reference's own tables and scenarios; through the C ABI."""
import numpy as np
import pytest

import progress_util as pu
import raft_amd
from oracle import pyoracle as po
from raft_amd import capi

pytestmark = pytest.mark.gpu

FX = pu.load_fixture()
# the two forms of the stepper (include/rafthip.h): K9's global lists, K9p's partition + per-tile lists
FORMS = {"linked": capi.FLAG_STEP_LINKED, "partitioned": capi.FLAG_STEP_PARTITION}
KEYS = ("pstate", "next", "pending", "infl_meta", "infl_bytes", "ring_index", "ring_bytes")


def upload_state(e, pl, cap, max_bytes, active, acks, last, first):
    e.progress_config(cap, max_bytes)
    e.upload_match(pl["match"])
    e.upload_progress(pl["pstate"], pl["next"], pl["pending"])
    e.upload_inflights(pl["infl_meta"], pl["infl_bytes"], pl["ring_index"], pl["ring_bytes"])
    e.upload_active(active)
    e.upload_acks(acks)
    e.upload_term_range(np.zeros(len(last), dtype=np.uint64), last)
    e.upload_first_index(first)


def assert_state(e, pl, active, acks, last):
    got = e.read_progress()
    for k in KEYS:
        assert np.array_equal(got[k], pl[k]), k
    assert np.array_equal(e.read_match(), pl["match"])
    assert np.array_equal(e.read_active(), active)
    assert np.array_equal(e.read_acks(), acks)
    assert np.array_equal(e.read_term_range()[1], last)


def test_constants_match_the_oracle():
    for name in ("ACT_SEND_APPEND", "ACT_UPDATED", "ACT_WAS_PAUSED", "ACT_CAUGHT_UP", "ACT_ERROR", "ACT_DEFERRED",
                 "ACT_DROPPED", "STEP_LIMIT", "STATE_PROBE", "STATE_REPLICATE", "STATE_SNAPSHOT"):
        assert getattr(capi, name) == getattr(po, name), name
    assert (capi.MSG_UNREACHABLE, capi.MSG_SNAP_STATUS, capi.MSG_SENT_APP, capi.MSG_SENT_SNAP) == (
        po.EV_UNREACHABLE, po.EV_SNAP_STATUS, po.EV_SENT_APP, po.EV_SENT_SNAP)


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("k", range(len(FX["scenarios"])))
def test_reference_scenarios(k, form):
    """tests/golden/progress.json on the engine: every step as its own batch (expected
    state and actions from the reference's tests), then the whole scenario as one batch."""
    sc = FX["scenarios"][k]
    steps = pu.scenario_steps(sc)
    cap, mb = sc["cap"], sc["max_bytes"]
    S, G, slot, group = 3, 5, 1, 3  # the scenario's Progress sits in the middle of a small engine
    pl = pu.init_planes(sc, S, G, slot, group)
    active, acks = pu.mask_planes(S, G)
    last = np.full(G, sc["last_index"], dtype=np.uint64)
    first = np.full(G, sc["first_index"], dtype=np.uint64)
    recs, auxs = [], []
    with raft_amd.Engine(G, S, flags=FORMS[form]) as e:
        upload_state(e, pl, cap, mb, active, acks, last, first)
        self_mask = np.zeros(G, dtype=active.dtype)
        self_mask[group] = 1  # the scenario's group is led from slot 0
        e.upload_self(self_mask)
        e.set_uniform_config(S)
        for i, st in enumerate(steps):
            want = pu.sent_plan(st, sc["src"])
            if want is not None:  # a send the reference performed from this state (maybeSendAppend, raft.go:600-666)
                plan, nxt = e.plan_appends(first=group, count=1)
                assert int(plan[slot, 0]) == want and int(plan[0, 0]) == capi.SEND_NONE, (sc["src"], i, plan[:, 0])
                assert int(nxt[slot, 0]) == int(e.read_progress()["next"][slot, group])
                if want == capi.SEND_EMPTY:
                    assert int(e.plan_appends(send_if_empty=False, first=group, count=1)[0][slot, 0]) == capi.SEND_NONE
            m, aux = pu.event(st, group, slot)
            recs.append(m)
            auxs.append(aux)
            act = e.step_progress(m, aux)
            pu.check_actions(st, int(act[0]), (sc["src"], i))
            got = e.read_progress()
            got["match"] = e.read_match()
            cell = pu.cell_state(got, cap, mb, slot, group)
            pu.check_expect(st, cell, (sc["src"], i))
            if cell["is_paused"]:
                assert int(e.plan_appends(first=group, count=1, want_next=False)[0][slot, 0]) == capi.SEND_NONE
        final = e.read_progress()
        final["match"] = e.read_match()
        if len(steps) <= capi.STEP_LIMIT and not any(st["ev"] == "appended" for st in steps):
            upload_state(e, pl, cap, mb, active, acks, last, first)
            e.step_progress(np.concatenate(recs), np.concatenate(auxs))
            again = e.read_progress()
            for key in KEYS:
                assert np.array_equal(again[key], final[key]), key
            assert np.array_equal(e.read_match(), final["match"])
        # untouched cells stay untouched
        other = np.ones((S, G), dtype=bool)
        other[slot, group] = False
        assert np.array_equal(final["next"][other], pl["next"][other])


@pytest.mark.parametrize("S,cap,mb,hot", [(3, 4, 0, 0), (5, 4, 150, 25), (9, 1, 0, 10), (16, 8, 300, 40), (33, 3, 0, 5),
                                           (64, 2, 60, 5), (5, 0, 0, 10)])
@pytest.mark.parametrize("form", FORMS)
def test_random_streams(S, cap, mb, hot, form):
    random_streams(S, cap, mb, hot, form, 2003, 40000)


@pytest.mark.parametrize("S,cap,mb,hot", [(5, 4, 150, 25), (9, 1, 0, 10), (33, 3, 0, 5)])
def test_random_streams_over_several_buckets_of_tiles(S, cap, mb, hot):
    """The partitioned form with more than one bucket of 128 tiles: 296 tiles of 1024 groups (592 of 512 at S = 33)."""
    random_streams(S, cap, mb, hot, "partitioned", 300007, 400000)


def test_a_batch_piled_onto_one_tile_falls_back_to_the_lists():
    """More than 2^17 events in one tile of 1024 groups: the partitioned form's scan raises its flag on the device and
    K9's link and apply passes step the batch (the grid-stride forms behind the flag)."""
    random_streams(9, 4, 0, 0, "partitioned", 1000, 140000)


@pytest.mark.parametrize("form", FORMS)
def test_state_carries_over_many_batches(form):
    """Twelve batches in a row on one engine, every one generated from the state the previous ones left (so acks
    free what earlier sends put in flight, probes follow rejections, snapshots end): actions and every plane
    against the oracle after each batch.  The partitioned form reuses its scratch — tile counts that must be zero
    again, record buffers, the sorted runs — from batch to batch."""
    rng = np.random.default_rng(1700)
    S, G, cap, mb, n = 5, 100003, 4, 300, 150000
    pl = pu.random_planes(rng, S, G, cap, mb)
    last = pl["match"].max(axis=0) + np.uint64(64)
    first = np.zeros(G, dtype=np.uint64)
    active, acks = pu.mask_planes(S, G)
    seen = np.zeros(256, dtype=np.int64)
    with raft_amd.Engine(G, S, flags=FORMS[form]) as e:
        upload_state(e, pl, cap, mb, active, acks, last, first)
        for batch in range(12):
            msgs, aux = pu.random_events(rng, n + 1000 * batch, S, G, pl, hot_cells=batch % 3 * 10)
            act = e.step_progress(msgs, aux)
            want = po.step_progress(pl, cap, mb, active, acks, last, first, msgs, aux)
            assert np.array_equal(act, want), batch
            assert_state(e, pl, active, acks, last)
            seen += np.bincount(want, minlength=256)
    assert seen[capi.ACT_UPDATED] > 0 and seen[capi.ACT_SEND_APPEND] > 0 and seen[capi.ACT_DEFERRED] > 0


def random_streams(S, cap, mb, hot, form, G, n):
    rng = np.random.default_rng(1500 + S)
    pl = pu.random_planes(rng, S, G, cap, mb)
    msgs, aux = pu.random_events(rng, n, S, G, pl, hot_cells=hot)
    last = rng.integers(0, 1 << 20, size=G).astype(np.uint64)
    first = rng.integers(0, 1 << 19, size=G).astype(np.uint64)
    active, acks = pu.mask_planes(S, G)
    active[:] = rng.integers(0, 4, size=G)
    with raft_amd.Engine(G, S, flags=FORMS[form]) as e:
        upload_state(e, pl, cap, mb, active, acks, last, first)
        act = e.step_progress(msgs, aux)
        want = po.step_progress(pl, cap, mb, active, acks, last, first, msgs, aux)
        assert np.array_equal(act, want)
        assert_state(e, pl, active, acks, last)
        seen = np.bincount(want, minlength=256)
        assert seen[capi.ACT_UPDATED] > 0 and seen[capi.ACT_SEND_APPEND] > 0 and seen[capi.ACT_DROPPED] > 0
        # a second batch on top (state carried on the device), device-resident inputs, no aux
        import torch
        msgs2, _ = pu.random_events(rng, n // 2, S, G, pl, hot_cells=hot)
        d = torch.from_numpy(msgs2.view(np.uint8)).cuda()
        dact = torch.zeros(len(msgs2), dtype=torch.uint8, device="cuda")
        L = raft_amd.load_library()
        assert L.rafthip_step_progress(e._h, d.data_ptr(), None, len(msgs2), dact.data_ptr()) == 0
        want2 = po.step_progress(pl, cap, mb, active, acks, last, first, msgs2, None)
        e.synchronize()
        assert np.array_equal(dact.cpu().numpy(), want2)
        assert_state(e, pl, active, acks, last)


@pytest.mark.parametrize("form", FORMS)
def test_long_lists_are_deferred_whole(form):
    rng = np.random.default_rng(1600)
    S, G, cap = 4, 50, 4
    pl = pu.random_planes(rng, S, G, cap, 0)
    n = 3000
    msgs, aux = pu.random_events(rng, n, S, G, pl)
    msgs["group"][:900] = 7   # 900 events on two Progress rows of group 7 ...
    msgs["slot"][:900] = rng.integers(0, 2, size=900)
    msgs["group"][900:964] = 8  # ... and exactly STEP_LIMIT on one of group 8
    msgs["slot"][900:964] = 3
    msgs["type"][900:964] = capi.MSG_HEARTBEAT_RESP
    keep = ~((msgs["group"] == 8) & (msgs["slot"] == 3))
    keep[900:964] = True
    msgs, aux = msgs[keep], np.ascontiguousarray(aux[keep])
    last = np.full(G, 1 << 19, dtype=np.uint64)
    first = np.zeros(G, dtype=np.uint64)
    active, acks = pu.mask_planes(S, G)
    with raft_amd.Engine(G, S, flags=FORMS[form]) as e:
        upload_state(e, pl, cap, 0, active, acks, last, first)
        act = e.step_progress(msgs, aux)
        want = po.step_progress(pl, cap, 0, active, acks, last, first, msgs, aux)
        assert np.array_equal(act, want)
        assert_state(e, pl, active, acks, last)
        g7 = (msgs["group"] == 7) & (msgs["slot"] < 2) & (want != capi.ACT_DROPPED) & (msgs["type"] != capi.MSG_APPENDED)
        assert (want[g7] == capi.ACT_DEFERRED).all() and g7.sum() > 2 * capi.STEP_LIMIT
        assert (want[(msgs["group"] == 8) & (msgs["slot"] == 3)] & capi.ACT_DEFERRED == 0).all()
        # the host resubmits what was deferred, in order, in slices the engine takes
        idx = np.flatnonzero(want == capi.ACT_DEFERRED)
        for lo in range(0, len(idx), 60):
            part = idx[lo:lo + 60]
            a = e.step_progress(np.ascontiguousarray(msgs[part]), np.ascontiguousarray(aux[part]))
            w = po.step_progress(pl, cap, 0, active, acks, last, first, np.ascontiguousarray(msgs[part]),
                                 np.ascontiguousarray(aux[part]))
            assert np.array_equal(a, w) and (w & capi.ACT_DEFERRED == 0).all()
        assert_state(e, pl, active, acks, last)


def test_requires_config_and_rejects_bad_arguments():
    with raft_amd.Engine(8, 3) as e:
        m = np.zeros(1, dtype=raft_amd.MSG_DTYPE)
        with pytest.raises(raft_amd.RafthipError):
            e.step_progress(m)
        e.progress_config(4, 0)
        e.progress_config(4, 0)  # same limits: fine
        with pytest.raises(raft_amd.RafthipError):
            e.progress_config(5, 0)
        assert len(e.step_progress(m[:0])) == 0


@pytest.mark.parametrize("form", ["auto", "linked"])
def test_leader_tick_with_progress_full_size(form):
    """cfg5 shape: 2^20 groups x (7 voters + 2 learners), one tick = 1M sends + 3M responses
    through rafthip_step_progress, then the commit pass; every plane vs the oracle.  A batch of this
    size (2^22 events over 9.4 x 10^6 Progress rows) takes the partitioned form by itself ("auto")."""
    G, S, V, cap = 1 << 20, 9, 7, 4
    rng = np.random.default_rng(0x5EED0008)
    pl = pu.random_planes(rng, S, G, cap, 0)
    n = 4 * G
    msgs, aux = pu.random_events(rng, n, S, G, pl)
    last = pl["match"].max(axis=0) + np.uint64(16)
    first = np.maximum(pl["match"].min(axis=0), 8) - np.uint64(8)
    active, acks = pu.mask_planes(S, G)
    with raft_amd.Engine(G, S, flags=FORMS.get(form, 0)) as e:
        e.set_uniform_config(V)
        upload_state(e, pl, cap, 0, active, acks, last, first)
        act = e.step_progress(msgs, aux)
        e.commit()
        want = po.step_progress(pl, cap, 0, active, acks, last, first, msgs, aux)
        assert np.array_equal(act, want)
        assert_state(e, pl, active, acks, last)
        assert np.array_equal(e.read_commit(), po.batch_committed(pl["match"], uniform_voters=V, threads=0))
        assert (want == capi.ACT_DEFERRED).sum() == 0 and (want & capi.ACT_UPDATED).any()


def dense_as_events(acks, hb, ctx, S, G, first=0):
    """The event stream rafthip_step_dense stands for: every MsgAppResp, then every MsgHeartbeatResp."""
    ss, gg = np.nonzero(acks)
    a = np.zeros(len(ss), dtype=raft_amd.MSG_DTYPE)
    a["slot"], a["group"], a["type"], a["index"] = ss, gg + first, capi.MSG_APP_RESP, acks[ss, gg] - np.uint64(1)
    bits = (hb.astype(np.uint64)[None, :] >> np.arange(S, dtype=np.uint64)[:, None]) & np.uint64(1)
    cbits = (ctx.astype(np.uint64)[None, :] >> np.arange(S, dtype=np.uint64)[:, None]) & np.uint64(1)
    hs, hg = np.nonzero(bits)
    h = np.zeros(len(hs), dtype=raft_amd.MSG_DTYPE)
    h["slot"], h["group"], h["type"], h["flags"] = hs, hg + first, capi.MSG_HEARTBEAT_RESP, cbits[hs, hg]
    return np.concatenate([a, h]), (ss, gg), (hs, hg)


@pytest.mark.parametrize("S,cap,mb,first_off", [(3, 4, 0, 0), (7, 2, 90, 5), (9, 4, 0, 0), (16, 3, 0, 17), (40, 2, 0, 0), (64, 1, 0, 3)])
def test_dense_step_equals_the_event_stream(S, cap, mb, first_off):
    """rafthip_step_dense (ack plane + heartbeat-response masks) = the same responses as records through
    the oracle: every MsgAppResp, then every MsgHeartbeatResp; states, Inflights, bit planes, actions."""
    rng = np.random.default_rng(2100 + S)
    G = 2501
    pl = pu.random_planes(rng, S, G, cap, mb)
    mdt = pu.gu.mask_dtype(S)
    last = pl["match"].max(axis=0) + rng.integers(0, 3, size=G).astype(np.uint64)
    firsti = rng.integers(0, 1 << 19, size=G).astype(np.uint64)
    active, acks_bits = pu.mask_planes(S, G)
    active[:] = rng.integers(0, 4, size=G)
    cnt = G - first_off - 2
    full = np.uint64((1 << S) - 1) if S < 64 else np.uint64(pu.gu.U64MAX)
    with raft_amd.Engine(G, S) as e:
        upload_state(e, pl, cap, mb, active, acks_bits, last, firsti)
        for rnd in range(3):
            near = np.where(rng.random((S, cnt)) < 0.5, pl["match"][:, first_off:first_off + cnt],
                            pl["next"][:, first_off:first_off + cnt]) + rng.integers(0, 6, size=(S, cnt)).astype(np.uint64)
            plane = np.where(rng.random((S, cnt)) < (0.6 if rnd else 0.05), near + np.uint64(1), 0).astype(np.uint64)
            hb = (rng.integers(0, 1 << 63, size=cnt, dtype=np.uint64) * np.uint64(2) + np.uint64(rnd & 1)) & full
            hb = np.where(rng.random(cnt) < 0.7, hb, 0).astype(mdt)
            ctx = (hb & rng.integers(0, 1 << 63, size=cnt, dtype=np.uint64).astype(mdt)).astype(mdt)
            plane = np.ascontiguousarray(plane)
            act = e.step_dense(plane, np.ascontiguousarray(hb), np.ascontiguousarray(ctx), first=first_off, count=cnt)
            ev, (ss, gg), (hs, hg) = dense_as_events(plane, hb, ctx, S, cnt, first_off)
            want = po.step_progress(pl, cap, mb, active, acks_bits, last, firsti, ev, None, limit=1 << 30)
            wact = np.zeros((S, cnt), dtype=np.uint8)
            wact[ss, gg] |= want[:len(ss)]
            wact[hs, hg] |= want[len(ss):]
            assert np.array_equal(act, wact), rnd
            assert_state(e, pl, active, acks_bits, last)
        assert (wact & capi.ACT_UPDATED).any() and (wact & capi.ACT_SEND_APPEND).any()
        # device-resident inputs, no actions wanted, acks only
        import torch
        d = torch.from_numpy(plane.view(np.int64)).cuda()
        L = raft_amd.load_library()
        assert L.rafthip_step_dense(e._h, first_off, cnt, d.data_ptr(), cnt, None, None, None, 0) == 0
        ev, _, _ = dense_as_events(plane, np.zeros(cnt, dtype=mdt), np.zeros(cnt, dtype=mdt), S, cnt, first_off)
        po.step_progress(pl, cap, mb, active, acks_bits, last, firsti, ev, None, limit=1 << 30)
        assert_state(e, pl, active, acks_bits, last)


@pytest.mark.parametrize("S,cap,mb,uniform", [(3, 4, 0, True), (5, 2, 100, False), (9, 1, 0, True), (16, 8, 300, False),
                                              (33, 3, 0, False), (64, 2, 60, True), (5, 0, 0, False)])
def test_plan_appends_random(S, cap, mb, uniform):
    """rafthip_plan_appends = maybeSendAppend's decision for every Progress row (raft.go:600-666, bcastAppend
    :689-696) against the oracle: every state, paused or not, full and empty windows, Next around the compacted
    prefix and past the log's end, inactive peers, rows outside the config, groups not led here; both
    sendIfEmpty values; a partial range; device outputs with a row pitch wider than the range."""
    import golden_util as gu
    from test_gpu_parity import rand_masks
    rng = np.random.default_rng(2500 + S)
    G = 3011
    pl = pu.random_planes(rng, S, G, cap, mb)
    last = pl["match"].max(axis=0) + rng.integers(0, 6, size=G).astype(np.uint64)
    first = np.where(rng.random(G) < 0.5, 0, pl["match"].min(axis=0) + rng.integers(0, 8, size=G).astype(np.uint64)).astype(np.uint64)
    mdt = gu.mask_dtype(S)
    full = mdt((1 << S) - 1 if S < 64 else gu.U64MAX)
    active = rand_masks(rng, S, G, joint_frac=0, empty_frac=0.1)[0]
    self_mask = (np.uint64(1) << rng.integers(0, S, size=G).astype(np.uint64)).astype(mdt)
    self_mask[rng.random(G) < 0.1] = 0
    with raft_amd.Engine(G, S) as e:
        upload_state(e, pl, cap, mb, active, np.zeros(G, dtype=mdt), last, first)
        e.upload_self(self_mask)
        if uniform:
            e.set_uniform_config(max(S - 2, 1))
            rows = None
        else:
            m0, m1 = rand_masks(rng, S, G)
            lrn = (~(m0 | m1)) & full & rand_masks(rng, S, G, joint_frac=0, empty_frac=0.3)[0]
            e.upload_config(m0, m1, lrn)
            rows = m0 | m1 | lrn
        seen = np.zeros(4, dtype=np.int64)
        for sie in (True, False):
            want = po.plan_appends(pl, cap, mb, rows, self_mask, active, first, last, sie)
            plan, nxt = e.plan_appends(send_if_empty=sie)
            assert np.array_equal(plan, want) and np.array_equal(nxt, pl["next"])
            seen += np.bincount(want.ravel(), minlength=4)
            a, n = G // 3, G // 2
            plan, nxt = e.plan_appends(send_if_empty=sie, first=a, count=n)
            assert np.array_equal(plan, want[:, a:a + n]) and np.array_equal(nxt, pl["next"][:, a:a + n])
        assert (seen > 0).all(), seen
        import torch
        dp = torch.full((S, G + 5), 0xEE, dtype=torch.uint8, device="cuda")
        dn = torch.zeros((S, G + 5), dtype=torch.int64, device="cuda")
        L = raft_amd.load_library()
        assert L.rafthip_plan_appends(e._h, 0, G, 1, dp.data_ptr(), dn.data_ptr(), G + 5) == 0
        e.synchronize()
        want = po.plan_appends(pl, cap, mb, rows, self_mask, active, first, last, True)
        assert np.array_equal(dp.cpu().numpy()[:, :G], want) and (dp.cpu().numpy()[:, G:] == 0xEE).all()
        assert np.array_equal(dn.cpu().numpy().view(np.uint64)[:, :G], pl["next"])
        # the plan reads, it never writes
        assert_state(e, pl, active, np.zeros(G, dtype=mdt), last)
