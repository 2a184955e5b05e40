"""Randomised sequences of C-ABI calls with random group ranges against a numpy
shadow of every plane: after each compute call the engine's outputs must equal
the oracle evaluated on the shadow.  Catches offset / stride / partial-range
mistakes that whole-plane tests cannot."""
import os

import numpy as np
import pytest

import golden_util as gu
import progress_util as pu
import raft_amd
import wire_util as wu
from oracle import pyoracle as po
from test_gpu_parity import rand_masks, rand_match

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S,G,seed", [(1, 1, 0), (3, 5000, 1), (5, 4097, 2), (8, 12289, 3), (9, 3000, 4), (16, 2500, 5),
                                      (17, 2049, 6), (40, 1500, 7), (64, 700, 8)])
def test_random_call_sequences(oracle, S, G, seed):
    rng = np.random.default_rng(9000 + seed + 1000 * int(os.environ.get("RAFTHIP_FUZZ_ROUND", "0")))  # more rounds: set the variable
    mdt = gu.mask_dtype(S)
    full = mdt((1 << S) - 1 if S < 64 else gu.U64MAX)
    sh = dict(match=np.zeros((S, G), np.uint64), votes=np.zeros((S, G), np.uint8), commit=np.zeros(G, np.uint64),
              m0=np.zeros(G, mdt), m1=np.zeros(G, mdt), ml=np.zeros(G, mdt), active=np.zeros(G, mdt),
              acks=np.zeros(G, mdt), self=np.zeros(G, mdt), tf=np.zeros(G, np.uint64), tl=np.zeros(G, np.uint64))
    checked = [0]
    uniform = [0]  # current uniform voter count, or None when per-group masks are in force

    def rng_range():
        if rng.random() < 0.3:
            return 0, G
        a = int(rng.integers(0, G))
        return a, int(rng.integers(0, G - a + 1))

    def cfg_kw():
        if uniform[0] is None:
            return dict(in_mask=sh["m0"], out_mask=sh["m1"])
        if uniform[0] == 0:  # both halves empty
            return dict(in_mask=np.zeros(G, mdt), out_mask=np.zeros(G, mdt))
        return dict(uniform_voters=uniform[0])

    def lrn_kw():
        kw = cfg_kw()
        if uniform[0] is None:
            kw["learner_mask"] = sh["ml"]
        elif uniform[0] == 0:
            kw["learner_mask"] = np.zeros(G, mdt)
        return kw

    cap, max_bytes = 3, 100
    pl = po.progress_planes(S, G, cap)  # the per-peer replication state; its Match plane IS sh["match"]
    pl["match"] = sh["match"]
    first_index = np.zeros(G, np.uint64)
    PKEYS = ("pstate", "next", "pending", "infl_meta", "infl_bytes", "ring_index", "ring_bytes")
    # the lookup state of the wire path: ids behind the rows, r.Term, pending ReadIndex context, campaign type
    wire = dict(ids=np.zeros((S, G), np.uint64), term=np.zeros(G, np.uint64), cw=np.zeros(G, np.uint64),
                cl=np.zeros(G, np.uint8), camp=np.zeros(G, np.uint8))

    # odd seeds pin rafthip_scatter_acks and the Progress stepper to their partitioned forms (K5p, K9p), so that the
    # scratch the two share — tile counts that must be zero between calls, cursors, the skew flag — lives through a
    # whole random sequence of other calls
    flags = (raft_amd.capi.FLAG_SCATTER_PARTITION | raft_amd.capi.FLAG_STEP_PARTITION) if seed % 2 else 0
    with raft_amd.Engine(G, S, flags=flags) as e:
        e.upload_active(sh["active"])
        e.upload_acks(sh["acks"])
        e.upload_self(sh["self"])
        e.upload_term_range(sh["tf"], sh["tl"])
        e.reset_votes()
        e.progress_config(cap, max_bytes)
        e.upload_ids(wire["ids"])
        e.upload_term(wire["term"])
        e.upload_read_ctx(wire["cw"], wire["cl"])
        e.upload_campaign(wire["camp"])
        # RawNode.Tick (K12) shares RecentActive, the self masks and the config with the calls above.  It is driven
        # from a generator of its own, so the sequences the seeds above produce stay what they were.
        rng2 = np.random.default_rng(77000 + seed)
        tk = dict(el=np.zeros(G, np.uint32), hb=np.zeros(G, np.uint32), rt=np.zeros(G, np.uint32),
                  role=np.zeros(G, np.uint8), flags=np.zeros(G, np.uint8))
        e.tick_config(4, 2, True)
        ticks = 0
        # a second, small engine of the same mask width, so that every other tick goes through rafthip_tick_many
        import tick_scenarios as ts
        G2 = 37
        e2 = raft_amd.Engine(G2, S)
        e2.set_uniform_config(S)
        e2.tick_config(3, 1, True)
        o2 = ts.OracleTicker(oracle, G2, S, 3, 1, True, uniform_voters=S)
        init2 = dict(role=rng2.integers(0, 4, G2), flags=rng2.integers(0, 4, G2), rt=rng2.integers(3, 6, G2),
                     active=np.full(G2, 1, mdt), self_mask=np.full(G2, 1, mdt))
        o2.load(**init2)
        e2.upload_tick_state(None, None, o2.rt, o2.role, o2.flags)
        e2.upload_active(o2.active)
        e2.upload_self(o2.self_mask)
        both = raft_amd.capi.EngineSet([e, e2])
        for step in range(170):
            if step % 5 == 4:
                if rng2.random() < 0.4:  # reset / becomeLeader / a transfer on a range of groups
                    b = int(rng2.integers(0, G))
                    k = int(rng2.integers(0, G - b + 1))
                    part = dict(el=rng2.integers(0, 8, k), hb=rng2.integers(0, 3, k), rt=rng2.integers(4, 8, k),
                                role=rng2.integers(0, 4, k), flags=rng2.integers(0, 4, k))
                    for key in tk:
                        tk[key][b:b + k] = part[key]
                    e.upload_tick_state(tk["el"][b:b + k], tk["hb"][b:b + k], tk["rt"][b:b + k], tk["role"][b:b + k],
                                        tk["flags"][b:b + k], first=b, count=k)
                else:
                    heard = rng2.integers(0, G + 2, size=int(rng2.integers(0, G + 1))).astype(np.uint32)  # stepFollower
                    assert e.tick_heard_from_leader(heard) == po.tick_heard_from_leader(tk["el"], heard), step
                    want = po.batch_tick(tk["el"], tk["hb"], tk["rt"], tk["role"], tk["flags"], 4, 2, True, S, sh["active"],
                                         sh["self"], **lrn_kw())
                    if ticks % 2:
                        counts = e.tick()
                    else:
                        c2 = both.tick()
                        counts = c2[0]
                        want2, want2_counts = o2.tick()
                        assert np.array_equal(e2.read_tick_actions(), want2) and np.array_equal(c2[1], want2_counts), step
                    act = e.read_tick_actions()
                    assert np.array_equal(act, want), (step, np.flatnonzero(act != want)[:8])
                    assert np.array_equal(counts, [np.count_nonzero(want & (1 << b)) for b in range(5)]), step
                    for got, key in zip(e.read_tick_state(), ("el", "hb", "rt", "role", "flags")):
                        assert np.array_equal(got, tk[key]), (step, key)
                    assert np.array_equal(e.read_active(), sh["active"]), step
                    ev, n_ev = e.read_tick_events(0xFF)
                    assert n_ev == np.count_nonzero(want) and np.array_equal(ev["group"], np.flatnonzero(want)), step
                    ticks += 1
            op = int(rng.integers(0, 26))
            checked[0] += 10 <= op < 16
            a, n = rng_range()
            if op == 0:
                part = rand_match(rng, S, max(n, 1))[:, :n]
                pad = np.zeros((S, n + 3), np.uint64)  # source rows wider than the range
                pad[:, :n] = part
                e.upload_match(pad, first=a, count=n, src_stride=n + 3)
                sh["match"][:, a:a + n] = part
            elif op == 1:
                e.reset_match(a, n)
                sh["match"][:, a:a + n] = 0
            elif op == 2:
                m0, m1 = rand_masks(rng, S, max(n, 1))
                m0, m1 = m0[:n], m1[:n]
                ml = (~(m0 | m1)) & full if rng.random() < 0.5 else None
                if uniform[0] is not None:  # a partial upload materialises the uniform config for the rest
                    v = uniform[0]
                    vm = mdt(((1 << v) - 1) if v < 64 else gu.U64MAX) if v else mdt(0)
                    sh["m0"][:], sh["m1"][:] = vm, 0
                    sh["ml"][:] = (full & ~vm) if v else 0
                e.upload_config(m0, m1, ml, first=a, count=n)
                sh["m0"][a:a + n], sh["m1"][a:a + n] = m0, m1
                sh["ml"][a:a + n] = ml if ml is not None else 0
                uniform[0] = None
            elif op == 3:
                v = int(rng.integers(0, S + 1))
                e.set_uniform_config(v)
                uniform[0] = v
            elif op == 4:
                part = rng.integers(0, 3, size=(S, n)).astype(np.uint8)
                e.upload_votes(np.ascontiguousarray(part), first=a, count=n, src_stride=n)
                sh["votes"][:, a:a + n] = part
            elif op == 5:
                e.reset_votes(a, n)
                sh["votes"][:, a:a + n] = 0
            elif op in (6, 7, 8):
                key = ("active", "acks", "self")[op - 6]
                bits, _ = rand_masks(rng, S, max(n, 1), joint_frac=0, empty_frac=0.2)
                bits = bits[:n]
                if key == "self":
                    bits = bits & (-bits.astype(np.int64)).astype(mdt)  # lowest set bit only
                getattr(e, "upload_" + key)(np.ascontiguousarray(bits), first=a, count=n)
                sh[key][a:a + n] = bits
            elif op == 9:
                k = int(rng.integers(0, 400))
                msgs = np.zeros(k, dtype=raft_amd.MSG_DTYPE)
                msgs["group"] = rng.integers(0, G + 2, size=k)
                msgs["slot"] = rng.integers(0, min(S + 1, 255), size=k)
                msgs["type"] = rng.choice(np.array([4, 9, 0xF0, 6], dtype=np.uint8), size=k)
                msgs["reject"] = rng.random(k) < 0.2
                msgs["flags"] = rng.integers(0, 2, size=k)
                msgs["index"] = rng.integers(0, 50, size=k)
                e.ingest_msgs(msgs)
                oracle.step_leader_msgs(sh["match"], sh["active"], sh["acks"], sh["tl"], msgs)
            elif op == 16:
                plane = np.where(rng.random((S, n)) < 0.4, rng.integers(0, 60, size=(S, n)), 0).astype(np.uint64)
                pad = np.zeros((S, n + 2), np.uint64)
                pad[:, :n] = plane
                e.merge_acks(pad, first=a, count=n, stride=n + 2)
                sh["match"][:, a:a + n] = np.maximum(sh["match"][:, a:a + n], plane)
            elif op == 17:
                st = rng.integers(0, 3, size=(S, n)).astype(np.uint8) | (rng.integers(0, 2, size=(S, n)).astype(np.uint8) << 2)
                nx = sh["match"][:, a:a + n] + rng.integers(1, 5, size=(S, n)).astype(np.uint64)
                pn = np.where((st & 3) == 2, rng.integers(0, 60, size=(S, n)), 0).astype(np.uint64)
                e.upload_progress(np.ascontiguousarray(st), np.ascontiguousarray(nx), np.ascontiguousarray(pn), first=a,
                                  count=n)
                pl["pstate"][:, a:a + n], pl["next"][:, a:a + n], pl["pending"][:, a:a + n] = st, nx, pn
                pl["infl_meta"][:, a:a + n] = 0
                pl["infl_bytes"][:, a:a + n] = 0
                fi = rng.integers(0, 30, size=n).astype(np.uint64)
                e.upload_first_index(fi, first=a, count=n)
                first_index[a:a + n] = fi
            elif op == 21:
                from test_gpu_progress import dense_as_events
                near = sh["match"][:, a:a + n] + rng.integers(0, 8, size=(S, n)).astype(np.uint64)
                plane = np.ascontiguousarray(np.where(rng.random((S, n)) < 0.5, near + np.uint64(1), 0).astype(np.uint64))
                hb, _ = rand_masks(rng, S, max(n, 1), joint_frac=0, empty_frac=0.3)
                hb = np.ascontiguousarray(hb[:n])
                ctx = np.ascontiguousarray(hb & rand_masks(rng, S, max(n, 1), joint_frac=0, empty_frac=0.3)[0][:n])
                act = e.step_dense(plane, hb, ctx, first=a, count=n)
                ev, (ss, gg), (hs, hg) = dense_as_events(plane, hb, ctx, S, n, a)
                want = po.step_progress(pl, cap, max_bytes, sh["active"], sh["acks"], sh["tl"], first_index, ev, None,
                                        limit=1 << 30)
                wact = np.zeros((S, n), dtype=np.uint8)
                wact[ss, gg] |= want[:len(ss)]
                wact[hs, hg] |= want[len(ss):]
                assert np.array_equal(act, wact), step
                got = e.read_progress()
                for key in PKEYS:
                    assert np.array_equal(got[key], pl[key]), (step, key)
                assert np.array_equal(e.read_match(), sh["match"]), step
                assert np.array_equal(e.read_active(), sh["active"]), step
            elif op == 22:
                which = int(rng.integers(0, 4))
                if which == 0:
                    part = np.ascontiguousarray(wu.random_ids(rng, S, max(n, 1), empty_frac=0.1)[:, :n] % np.uint64(1 << 40))
                    e.upload_ids(part, first=a, count=n)
                    wire["ids"][:, a:a + n] = part
                elif which == 1:
                    part = rng.integers(0, 4, size=n).astype(np.uint64)
                    e.upload_term(part, first=a, count=n)
                    wire["term"][a:a + n] = part
                elif which == 2:
                    cl = rng.integers(0, 9, size=n).astype(np.uint8)
                    cw = np.where(cl > 0, rng.integers(0, 4, size=n), 0).astype(np.uint64)  # fits any length >= 1
                    e.upload_read_ctx(cw, cl, first=a, count=n)
                    wire["cw"][a:a + n], wire["cl"][a:a + n] = cw, cl
                else:
                    part = rng.choice(np.array([0, 6, 18], dtype=np.uint8), size=n)
                    e.upload_campaign(part, first=a, count=n)
                    wire["camp"][a:a + n] = part
            elif op == 23:
                k = int(rng.integers(0, 400))
                ctxs = [int(wire["cw"][g]).to_bytes(8, "little")[:int(wire["cl"][g])] for g in range(G)]
                b1, g1 = wu.random_vote_batch(rng, k // 2, G, wire["ids"], wire["term"] + np.uint64(1))
                b2, g2 = wu.random_batch(rng, k - k // 2, G, wire["ids"], wire["term"], ctxs)
                order = rng.permutation(k)
                blobs = [(b1 + b2)[i] for i in order]
                groups = np.ascontiguousarray(np.concatenate([g1, g2])[order].astype(np.uint32))
                data, offsets = wu.pack(blobs)
                msgs, status, counts = e.ingest_wire(data, offsets, groups)
                wm, ws, wc = oracle.step_wire_msgs(sh["match"], sh["active"], sh["acks"], wire["ids"], wire["term"], wire["cw"],
                                                   wire["cl"], data, offsets, groups, campaign=wire["camp"], votes=sh["votes"])
                assert np.array_equal(status, ws) and np.array_equal(counts, wc) and np.array_equal(msgs, wm), step
                assert np.array_equal(e.read_votes(), sh["votes"]) and np.array_equal(e.read_match(), sh["match"]), step
                assert np.array_equal(e.read_active(), sh["active"]) and np.array_equal(e.read_acks(), sh["acks"]), step
            elif op == 25:
                sie = bool(rng.integers(0, 2))
                plan, nxt = e.plan_appends(send_if_empty=sie, first=a, count=n)
                rows = None if uniform[0] else (sh["m0"] | sh["m1"] | sh["ml"])  # uniform: every slot holds a row
                if uniform[0] == 0:
                    rows = np.zeros(G, mdt)
                want = po.plan_appends(pl, cap, max_bytes, rows, sh["self"], sh["active"], first_index, sh["tl"], sie)
                assert np.array_equal(plan, want[:, a:a + n]) and np.array_equal(nxt, pl["next"][:, a:a + n]), step
            elif op == 24:
                out, lens = e.emit_heartbeats(first=a, count=n)
                want, wlen = oracle.bcast_heartbeat(sh["match"], sh["commit"], wire["ids"], wire["term"], wire["cw"], wire["cl"],
                                                    sh["self"])
                assert np.array_equal(lens, wlen[:, a:a + n]) and np.array_equal(out, want[:, a:a + n]), step
            elif op == 20:
                k = int(rng.integers(0, 500))
                msgs = np.zeros(k, dtype=raft_amd.MSG_DTYPE)
                msgs["group"] = rng.integers(0, G + 2, size=k)
                msgs["slot"] = rng.integers(0, min(S + 1, 255), size=k)
                msgs["type"] = rng.choice(np.array([6, 18, 4], dtype=np.uint8), size=k)
                msgs["reject"] = rng.integers(0, 2, size=k)
                e.record_votes(msgs)
                oracle.record_votes(sh["votes"], msgs)
                assert np.array_equal(e.read_votes(a, n), sh["votes"][:, a:a + n]), step
            elif op in (18, 19):
                k = int(rng.integers(0, 600))
                msgs, aux = pu.random_events(rng, k, S, G, pl, hot_cells=int(rng.integers(0, 4)))
                act = e.step_progress(msgs, aux)
                want = po.step_progress(pl, cap, max_bytes, sh["active"], sh["acks"], sh["tl"], first_index, msgs, aux)
                assert np.array_equal(act, want), step
                got = e.read_progress()
                for key in PKEYS:
                    assert np.array_equal(got[key], pl[key]), (step, key)
                assert np.array_equal(e.read_match(), sh["match"]), step
            elif op == 10:
                e.commit()
                sh["commit"] = oracle.batch_committed(sh["match"], **cfg_kw())
                assert np.array_equal(e.read_commit(), sh["commit"]), step
            elif op == 11:
                pairs, cnt = e.commit_changed()
                want = oracle.batch_committed(sh["match"], **cfg_kw())
                ch = np.flatnonzero(want != sh["commit"])
                assert cnt == len(ch) and np.array_equal(pairs["group"], ch.astype(np.uint64)), step
                assert np.array_equal(pairs["commit"], want[ch]), step
                sh["commit"] = want
            elif op == 12:
                tf = rng.integers(0, 10, size=n).astype(np.uint64)
                tl = tf + rng.integers(0, 40, size=n).astype(np.uint64) - (rng.random(n) < 0.2)
                e.upload_term_range(tf, np.ascontiguousarray(tl), first=a, count=n)
                sh["tf"][a:a + n], sh["tl"][a:a + n] = tf, tl
                com = rng.integers(0, 20, size=n).astype(np.uint64)
                e.upload_commit(com, first=a, count=n)
                sh["commit"][a:a + n] = com
            elif op == 13:
                pairs, cnt = e.commit_advance()
                mci = oracle.batch_committed(sh["match"], **cfg_kw())
                adv = (mci > sh["commit"]) & (mci >= sh["tf"]) & (mci <= sh["tl"])
                idx = np.flatnonzero(adv)
                assert cnt == len(idx) and np.array_equal(pairs["group"], idx.astype(np.uint64)), step
                sh["commit"] = np.where(adv, mci, sh["commit"])
                assert np.array_equal(e.read_commit(a, n), sh["commit"][a:a + n]), step
                hb = e.heartbeat_commit(first=a, count=n)
                assert np.array_equal(hb, np.minimum(sh["match"][:, a:a + n], sh["commit"][None, a:a + n])), step
            elif op == 14:
                e.tally_votes()
                res, gr, rj = e.read_vote_result(a, n)
                ores, ogr, orj = oracle.batch_tally_votes(sh["votes"], **lrn_kw())
                assert np.array_equal(res, ores[a:a + n]), step
                assert np.array_equal(gr, ogr[a:a + n]) and np.array_equal(rj, orj[a:a + n]), step
                e.ack_quorum()
                assert np.array_equal(e.read_ack_quorum(a, n),
                                      oracle.batch_ack_quorum(sh["acks"], S, **cfg_kw())[a:a + n]), step
            else:
                if rng.random() < 0.5:
                    e.quorum_active()
                    want = oracle.batch_quorum_active(sh["active"], S, **lrn_kw())
                else:
                    e.check_quorum()
                    want = oracle.batch_check_quorum(sh["active"], S, sh["self"], **lrn_kw())
                assert np.array_equal(e.read_quorum_active(a, n), want[a:a + n]), step
                assert np.array_equal(e.read_active(), sh["active"]), step
        assert checked[0] >= (25 if os.environ.get("RAFTHIP_FUZZ_ROUND", "0") == "0" else 10)  # the committed seeds check at least 25 calls each
        assert ticks >= 10
        e2.close()
        assert np.array_equal(e.read_match(), sh["match"])
        assert np.array_equal(e.read_acks(), sh["acks"])
        assert np.array_equal(e.read_term_range()[1], sh["tl"])
        got = e.read_progress()
        for key in PKEYS:
            assert np.array_equal(got[key], pl[key]), key
