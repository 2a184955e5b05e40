"""The oracle's RawNode.Tick (oracle_batch_tick, raft.go:823-862) against the tick loops of the reference's own
tests (tests/tick_scenarios.py cites each), and against a line-by-line Python restatement on random state."""
import numpy as np
import pytest

import tick_scenarios as ts


def oracle_factory(oracle):
    def mk(G, S, election_tick, heartbeat_tick, check_quorum, uniform_voters=0, masks=None):
        return ts.OracleTicker(oracle, G, S, election_tick, heartbeat_tick, check_quorum, uniform_voters, masks)
    return mk


@pytest.mark.parametrize("scenario", ts.SCENARIOS, ids=lambda f: f.__name__)
def test_reference_tick_loops_on_the_oracle(oracle, scenario):
    scenario(oracle_factory(oracle))


def py_tick(el, hb, rt, role, flags, et, ht, cq, active, self_mask, qa):
    """tickElection / tickHeartbeat for one group, in the reference's statement order.  qa: QuorumActive()."""
    act = 0
    if role == ts.LEADER:
        hb += 1
        el += 1
        if el >= et:
            el = 0
            if cq:
                act |= ts.CHECK_QUORUM
                if not qa:
                    act |= ts.STEP_DOWN
                    role, el, hb, active = ts.FOLLOWER, 0, 0, 0
                    flags &= ~ts.TRANSFER
                else:
                    active &= self_mask
            if role == ts.LEADER and flags & ts.TRANSFER:
                act |= ts.ABORT_TRANSFER
                flags &= ~ts.TRANSFER
        if role == ts.LEADER and hb >= ht:
            hb = 0
            act |= ts.BEAT
    else:
        el += 1
        if flags & ts.PROMOTABLE and el >= rt:
            el = 0
            act |= ts.HUP
    return el, hb, role, flags, active, act


@pytest.mark.parametrize("S,cq", [(3, True), (5, False), (9, True)])
def test_oracle_tick_against_the_statement_order(oracle, S, cq):
    rng = np.random.default_rng(40 + S)
    G, et, ht = 3000, 6, 2
    t = ts.OracleTicker(oracle, G, S, et, ht, cq, uniform_voters=S)
    mdt = t.mdt
    t.load(el=rng.integers(0, 2 * et, G), hb=rng.integers(0, ht + 1, G), rt=rng.integers(et, 2 * et, G),
           role=rng.integers(0, 4, G), flags=rng.integers(0, 4, G), active=rng.integers(0, 1 << S, G).astype(mdt),
           self_mask=(1 << rng.integers(0, S, G)).astype(mdt))
    for _ in range(3 * et):
        before = t.state()
        self_mask = t.self_mask.copy()
        qa = oracle.batch_quorum_active(before[5].copy(), S, uniform_voters=S)
        act, _ = t.tick()
        after = t.state()
        for g in range(G):
            el, hb, role, flags, active, a = py_tick(int(before[0][g]), int(before[1][g]), int(before[2][g]), int(before[3][g]),
                                                     int(before[4][g]), et, ht, cq, int(before[5][g]), int(self_mask[g]), bool(qa[g]))
            assert (el, hb, role, flags, active, a) == (int(after[0][g]), int(after[1][g]), int(after[3][g]), int(after[4][g]),
                                                        int(after[5][g]), int(act[g])), g
        # responses between ticks: some peers become active again
        t.load(active=after[5] | rng.integers(0, 1 << S, G).astype(mdt))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_tick_shards_by_group_range_without_any_exchange(oracle, world):
    """RawNode.Tick couples no two groups, so the multi-GPU form is the sharding of everything else (raft_amd/multigpu.py
    shard_range): every rank ticks its own contiguous range and the concatenation is the one-engine result — state,
    actions and the sum of the per-rank counts.  (One process here; the ranks' engines are independent by
    construction, there is nothing to rendezvous on.)"""
    from raft_amd import multigpu
    rng = np.random.default_rng(50 + world)
    G, S, et, ht = 10007, 5, 4, 2
    whole = ts.OracleTicker(oracle, G, S, et, ht, True, uniform_voters=S)
    init = dict(el=rng.integers(0, 2 * et, G), hb=rng.integers(0, ht + 1, G), rt=rng.integers(et, 2 * et, G),
                role=rng.integers(0, 4, G), flags=rng.integers(0, 4, G), active=rng.integers(0, 1 << S, G).astype(np.uint8),
                self_mask=(1 << rng.integers(0, S, G)).astype(np.uint8))
    whole.load(**init)
    ranks = []
    for r in range(world):
        first, count = multigpu.shard_range(G, r, world)
        t = ts.OracleTicker(oracle, count, S, et, ht, True, uniform_voters=S)
        t.load(**{k: np.asarray(v)[first:first + count] for k, v in init.items()})
        ranks.append(t)
    for _ in range(2 * et + 1):
        act, counts = whole.tick()
        parts = [t.tick() for t in ranks]
        assert np.array_equal(np.concatenate([a for a, _ in parts]), act)
        assert np.array_equal(sum(c for _, c in parts), counts)
    for k, plane in enumerate(whole.state()):
        assert np.array_equal(np.concatenate([t.state()[k] for t in ranks]), plane)
