"""GPU parity of K12, rafthip_tick (RawNode.Tick for every group, raft.go:823-862), through the C ABI: the tick
loops of the reference's own tests (tests/tick_scenarios.py), then random state over every mask width, uniform and
joint configs with learners, ragged group counts, many ticks in a row — every plane, the action bytes and the counts
against the oracle after each tick."""
import numpy as np
import pytest

import golden_util as gu
import raft_amd
import tick_scenarios as ts
from test_gpu_parity import rand_masks

pytestmark = pytest.mark.gpu


class EngineTicker:
    """tick_scenarios' ticker interface over one engine."""

    def __init__(self, G, S, election_tick, heartbeat_tick, check_quorum, uniform_voters=0, masks=None):
        self.e = raft_amd.Engine(G, S)
        self.G, self.S = G, S
        self.mdt = self.e.mask_dtype
        if masks:
            self.e.upload_config(*masks)
        else:
            self.e.set_uniform_config(uniform_voters)
        self.e.tick_config(election_tick, heartbeat_tick, check_quorum)
        self.e.upload_active(np.zeros(G, self.mdt))
        self.e.upload_self(np.zeros(G, self.mdt))

    def load(self, el=None, hb=None, rt=None, role=None, flags=None, active=None, self_mask=None):
        self.e.upload_tick_state(el, hb, rt, role, flags)
        if active is not None:
            self.e.upload_active(np.ascontiguousarray(active, dtype=self.mdt))
        if self_mask is not None:
            self.e.upload_self(np.ascontiguousarray(self_mask, dtype=self.mdt))

    def heard(self, groups):
        return self.e.tick_heard_from_leader(np.asarray(groups, dtype=np.uint32))

    def tick(self):
        counts = self.e.tick()
        act = self.e.read_tick_actions()
        assert np.array_equal(counts, ts.counts_of(act))
        return act, counts

    def state(self):
        return self.e.read_tick_state() + (self.e.read_active(),)

    def close(self):
        self.e.close()


def check_events(e, act, mask, cap=None):
    """rafthip_read_tick_events against the action plane: the groups with a bit of `mask`, in group order, each with
    its whole action byte; a destination shorter than the list gets the head of it and the full count."""
    want = np.flatnonzero((act & np.uint8(mask & 0xFF)) != 0)
    ev, n = e.read_tick_events(mask, max_events=cap)
    assert n == len(want)
    keep = len(want) if cap is None else min(cap, len(want))
    assert np.array_equal(ev["group"], want[:keep]) and np.array_equal(ev["action"], act[want[:keep]])


@pytest.mark.parametrize("scenario", ts.SCENARIOS, ids=lambda f: f.__name__)
def test_reference_tick_loops_on_the_engine(scenario):
    made = []

    def mk(**kw):
        made.append(EngineTicker(**kw))
        return made[-1]
    try:
        scenario(mk)
    finally:
        for t in made:
            t.close()


@pytest.mark.parametrize("S", [1, 3, 5, 8, 9, 16, 17, 32, 33, 64])
@pytest.mark.parametrize("config", ["uniform", "joint"])
def test_tick_random_state_every_mask_width(oracle, S, config):
    rng = np.random.default_rng(7000 + S + (100 if config == "joint" else 0))
    G = [1, 5, 1023, 1025, 4097, 20000][S % 6]
    et, ht, cq = 5, 2, True
    mdt = gu.mask_dtype(S)
    full = (1 << S) - 1

    def bits():
        if S < 64:
            return rng.integers(0, full + 1, size=G, dtype=np.uint64).astype(mdt)
        return (rng.integers(0, 1 << 63, size=G, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=G, dtype=np.uint64)).astype(mdt)

    masks = None
    V = max(1, S - 1)
    if config == "joint":
        m0, m1 = rand_masks(rng, S, G, joint_frac=0.5, empty_frac=0.05)
        learners = (~(m0 | m1)) & mdt(full if S < 64 else gu.U64MAX)
        learners = np.where(rng.random(G) < 0.5, learners, mdt(0)).astype(mdt)
        masks = (m0, m1, learners)
    o = ts.OracleTicker(oracle, G, S, et, ht, cq, uniform_voters=0 if masks else V, masks=masks)
    t = EngineTicker(G, S, et, ht, cq, uniform_voters=0 if masks else V, masks=masks)
    try:
        self_mask = (np.uint64(1) << rng.integers(0, S, G).astype(np.uint64)).astype(mdt)
        self_mask[::7] = 0  # a leader without a Progress row of its own
        init = dict(el=rng.integers(0, 2 * et, G), hb=rng.integers(0, ht + 1, G), rt=rng.integers(et, 2 * et, G),
                    role=rng.integers(0, 4, G), flags=rng.integers(0, 4, G), active=bits() & bits(), self_mask=self_mask)
        o.load(**init)
        t.load(**init)
        total = np.zeros(5, dtype=np.uint64)
        for k in range(3 * et + 1):
            want_act, want_counts = o.tick()
            act, counts = t.tick()
            assert np.array_equal(act, want_act), (k, np.flatnonzero(act != want_act)[:8])
            assert np.array_equal(counts, want_counts)
            for name, a, b in zip(("el", "hb", "rt", "role", "flags", "active"), t.state(), o.state()):
                assert np.array_equal(a, b), (k, name, np.flatnonzero(a != b)[:8])
            total += counts
            # stepFollower: some groups heard from their leader (duplicates, ids outside the engine, a device-resident list)
            heard = rng.integers(0, G + 3, size=int(rng.integers(0, 2 * G + 2))).astype(np.uint32)
            if k % 2:
                import torch
                d = torch.from_numpy(heard.view(np.int32)).cuda()
                bad = t.e.tick_heard_from_leader((d.data_ptr(), len(heard)))
            else:
                bad = t.e.tick_heard_from_leader(heard)
            assert bad == o.heard(heard) == np.count_nonzero(heard >= G)
            check_events(t.e, act, [0xFF, ts.HUP | ts.STEP_DOWN, ts.ABORT_TRANSFER, ts.BEAT, 0][k % 5], cap=[None, 3, 0][k % 3])
            # between ticks: responses arrive (RecentActive, raft.go:1337,1528), a few followers win elections
            a = o.state()[5] | (bits() & bits())
            role = o.state()[3]
            role[rng.random(G) < 0.05] = ts.LEADER
            flags = o.state()[4] | (rng.random(G) < 0.03).astype(np.uint8) * np.uint8(ts.TRANSFER)
            for x in (o, t):
                x.load(active=a, role=role, flags=flags)
        if G >= 1000:
            assert np.all(total > 0), total  # every action occurred
    finally:
        t.close()


def test_tick_without_check_quorum_needs_no_active_bits_and_counts_are_optional(oracle):
    G, S = 3000, 3
    rng = np.random.default_rng(7300)
    with raft_amd.Engine(G, S) as e:
        with pytest.raises(raft_amd.RafthipError):
            e.tick()  # no rafthip_tick_config yet
        for bad in ((10, 0), (1, 1), (3, 5)):  # Config.validate, raft.go:296-302
            with pytest.raises(raft_amd.RafthipError):
                e.tick_config(bad[0], bad[1], False)
        e.set_uniform_config(3)
        e.tick_config(10, 3, False)
        role = rng.integers(0, 4, G).astype(np.uint8)
        flags = rng.integers(0, 4, G).astype(np.uint8)
        rt = rng.integers(10, 20, G).astype(np.uint32)
        e.upload_tick_state(None, None, rt, role, flags)
        o = ts.OracleTicker(oracle, G, S, 10, 3, False, uniform_voters=3)
        o.load(rt=rt, role=role, flags=flags)
        for k in range(25):
            want, _ = o.tick()
            e.tick(counts=(k % 2 == 0))
            assert np.array_equal(e.read_tick_actions(), want), k
        for a, b in zip(e.read_tick_state(), o.state()[:5]):
            assert np.array_equal(a, b)
        e.tick_config(10, 3, True)
        with pytest.raises(raft_amd.RafthipError):
            e.tick()  # CheckQuorum without RecentActive bits


def test_tick_a_million_groups(oracle):
    """2^20 groups x 9 slots (7 voters + 2 learners, cfg5's shape), steady state: every group a leader or follower
    in a random phase of its timers, twelve ticks."""
    G, S, V, et, ht = 1 << 20, 9, 7, 10, 1
    rng = np.random.default_rng(7400)
    mdt = gu.mask_dtype(S)
    o = ts.OracleTicker(oracle, G, S, et, ht, True, uniform_voters=V)
    t = EngineTicker(G, S, et, ht, True, uniform_voters=V)
    try:
        init = dict(el=rng.integers(0, et, G), hb=np.zeros(G), rt=rng.integers(et, 2 * et, G),
                    role=np.where(rng.random(G) < 0.4, ts.LEADER, ts.FOLLOWER), flags=np.full(G, ts.PROMOTABLE),
                    active=rng.integers(0, 1 << S, G).astype(mdt), self_mask=np.ones(G, mdt))
        o.load(**init)
        t.load(**init)
        for k in range(12):
            want_act, want_counts = o.tick()
            counts = t.e.tick()
            assert np.array_equal(counts, want_counts), k
            assert np.array_equal(t.e.read_tick_actions(), want_act), k
            check_events(t.e, want_act, ts.HUP | ts.STEP_DOWN | ts.ABORT_TRANSFER)
        for name, a, b in zip(("el", "hb", "rt", "role", "flags", "active"), t.state(), o.state()):
            assert np.array_equal(a, b), name
        # the records straight into device memory and into page-locked host memory
        import torch
        want = np.flatnonzero((want_act & np.uint8(ts.BEAT)) != 0)
        d = torch.zeros(2 * G, dtype=torch.int32, device="cuda")
        assert t.e.read_tick_events(ts.BEAT, max_events=G, dst=d.data_ptr()) == len(want)
        t.e.synchronize()
        got = d.cpu().numpy().view(raft_amd.capi.TICK_EVENT_DTYPE)[:len(want)]
        assert np.array_equal(got["group"], want) and np.all(got["action"] & ts.BEAT)
        with raft_amd.PinnedArray(1000, raft_amd.capi.TICK_EVENT_DTYPE) as pin:
            assert t.e.read_tick_events(ts.BEAT, max_events=1000, dst=pin.ctypes.data) == len(want)
            assert np.array_equal(pin["group"], want[:1000])
    finally:
        t.close()


@pytest.mark.parametrize("S", [5, 9])
def test_tick_many_shards_in_one_launch(oracle, S):
    """rafthip_tick_many: the shards of several engines in one launch — ragged sizes, uniform and masked configs, each
    shard with its own tick constants — every shard against its own oracle after each of eight ticks; then engines of
    another mask width in the list (the call ticks them one by one) and engines on streams of their own."""
    import torch
    rng = np.random.default_rng(7600 + S)
    mdt = gu.mask_dtype(S)
    sizes = [1, 1023, 1024, 1025, 4097, 9000, 5, 2048, 3333, 4096, 100, 7777, 1, 2, 3, 4, 5000, 600, 70]  # > 16: two launches
    streams = [torch.cuda.Stream() for _ in range(3)]
    shards = []
    try:
        for i, G in enumerate(sizes):
            masks = None
            V = 1 + i % S
            if i % 3 == 0:
                m0, m1 = rand_masks(rng, S, G, joint_frac=0.5, empty_frac=0.05)
                masks = (m0, m1, ((~(m0 | m1)) & mdt((1 << S) - 1)).astype(mdt))
            et, ht, cq = 3 + i % 4, 1 + i % 2, i % 5 != 0
            o = ts.OracleTicker(oracle, G, S, et, ht, cq, uniform_voters=0 if masks else V, masks=masks)
            t = EngineTicker(G, S, et, ht, cq, uniform_voters=0 if masks else V, masks=masks)
            if i % 4 == 1:  # an engine on a stream of its own
                t.close()
                t.e = raft_amd.Engine(G, S, stream=streams[i % 3].cuda_stream)
                if masks:
                    t.e.upload_config(*masks)
                else:
                    t.e.set_uniform_config(V)
                t.e.tick_config(et, ht, cq)
                t.e.upload_active(np.zeros(G, mdt))
                t.e.upload_self(np.zeros(G, mdt))
            init = dict(el=rng.integers(0, 2 * et, G), hb=rng.integers(0, ht + 1, G), rt=rng.integers(et, 2 * et, G),
                        role=rng.integers(0, 4, G), flags=rng.integers(0, 4, G),
                        active=rng.integers(0, 1 << S, G).astype(mdt) & rng.integers(0, 1 << S, G).astype(mdt),
                        self_mask=(1 << rng.integers(0, S, G)).astype(mdt))
            o.load(**init)
            t.load(**init)
            shards.append((o, t))
        es = raft_amd.capi.EngineSet([t.e for _, t in shards])
        for k in range(8):
            counts = es.tick(counts=(k != 3))
            for i, (o, t) in enumerate(shards):
                want_act, want_counts = o.tick()
                assert np.array_equal(t.e.read_tick_actions(), want_act), (k, i)
                if counts is not None:
                    assert np.array_equal(counts[i], want_counts), (k, i)
                for name, a, b in zip(("el", "hb", "rt", "role", "flags", "active"), t.state(), o.state()):
                    assert np.array_equal(a, b), (k, i, name)
                a = o.state()[5] | (rng.integers(0, 1 << S, o.G).astype(mdt) & rng.integers(0, 1 << S, o.G).astype(mdt))
                o.load(active=a)
                t.load(active=a)
        # another mask width in the list: ticked one by one, same results
        G2, S2 = 3000, 20
        o2 = ts.OracleTicker(oracle, G2, S2, 4, 1, True, uniform_voters=11)
        t2 = EngineTicker(G2, S2, 4, 1, True, uniform_voters=11)
        shards.append((o2, t2))
        init = dict(el=rng.integers(0, 8, G2), hb=np.zeros(G2), rt=rng.integers(4, 8, G2), role=rng.integers(0, 4, G2),
                    flags=rng.integers(0, 4, G2), active=rng.integers(0, 1 << S2, G2).astype(np.uint32),
                    self_mask=np.ones(G2, np.uint32))
        o2.load(**init)
        t2.load(**init)
        es = raft_amd.capi.EngineSet([t.e for _, t in shards[:3]] + [t2.e])
        counts = es.tick()
        for i, (o, t) in enumerate(shards[:3] + [(o2, t2)]):
            want_act, want_counts = o.tick()
            assert np.array_equal(t.e.read_tick_actions(), want_act) and np.array_equal(counts[i], want_counts), i
    finally:
        for _, t in shards:
            t.close()
