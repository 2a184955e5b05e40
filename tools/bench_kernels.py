#!/usr/bin/env python3
"""Per-kernel roofline numbers for every kernel of the engine at the sizes
BASELINE.json's configs name — a development tool;
bench.py stays the headline line.

Each row: kernel, units per launch, algorithmic bytes per unit,
average launch duration from HIP events on the engine's stream over `--iters`
back-to-back launches, achieved GB/s and the fraction of the 8 TB/s HBM peak.
Working sets are chosen larger than the 256 MiB Infinity Cache so the passes
read HBM.  Run under `rocprofv3 --kernel-trace --stats` for the matching
per-kernel averages.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK = 8000.0
ATOMIC_RATE = 26.6e9  # scattered 8-byte atomics per second, measured: tools/scatter_probe.py, DESIGN.md §4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--scale", type=int, default=0, help="log2 shrink of every G (for quick runs)")
    ap.add_argument("--only", default="")
    args = ap.parse_args()

    import torch

    import golden_util as gu
    import raft_amd

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    stream = torch.cuda.Stream(device=dev)
    rows = []

    def timed(name, unit, units, bytes_per_unit, fn, setup=None, iters=None, atomics_per_unit=0):
        if args.only and args.only not in name:
            return
        iters = iters or args.iters
        with torch.cuda.stream(stream):
            for _ in range(3):
                if setup:
                    setup()
                fn()
            torch.cuda.synchronize()
            tot = 0.0
            if setup is None:  # back-to-back launches between two events: no host gap inside
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(iters):
                    fn()
                e1.record(stream)
                e1.synchronize()
                tot = e0.elapsed_time(e1)
            else:
                for _ in range(iters):
                    setup()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    fn()
                    e1.record(stream)
                    e1.synchronize()
                    tot += e0.elapsed_time(e1)
        us = 1e3 * tot / iters
        gbs = units * bytes_per_unit / us / 1e3
        row = {"kernel": name, "unit": unit, "units_per_launch": units, "bytes_per_unit": bytes_per_unit,
               "avg_launch_us": round(us, 2), "achieved_GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / HBM_PEAK, 3),
               "units_per_s": units / us * 1e6}
        if atomics_per_unit:  # bound by the rate of scattered memory-side atomics, not by bytes
            row["bound"] = "atomic-rate"
            row["frac_of_atomic_rate"] = round(atomics_per_unit * units / us * 1e6 / ATOMIC_RATE, 3)
        rows.append(row)
        print(json.dumps(row), flush=True)

    sh = args.scale
    S = raft_amd.capi

    # ---- cfg2: 2^20 x 3 voters is cache-sized; measure the same kernel at 2^24 -------------
    G = (1 << 24) >> sh
    for V in (3, 5, 7):
        m = gu.dist_a(0x5EED0002, V, G)
        with raft_amd.Engine(G, V, stream=stream.cuda_stream) as e:
            e.set_uniform_config(V)
            e.upload_match(m)
            timed("k_commit_uniform<%d> (K1)" % V, "group", G, 8 * V + 8, e.commit)
            if V == 5:
                tf = np.zeros(G, dtype=np.uint64)
                tl = np.full(G, gu.U64MAX, dtype=np.uint64)
                e.upload_term_range(tf, tl)
                zero = np.zeros(G, dtype=np.uint64)
                d_pairs = torch.empty(2 * G, dtype=torch.int64, device=dev)
                # commit_changed / commit_advance = the commit kernel in a reporting mode + k_pack_changed + the
                # count read back; pairs land in device memory.  The commit plane is reset before each launch
                # of the first row so every group reports: Match 8V + commit read 8 + written 8 + pair 16
                timed("commit_changed, all groups change (K1 + k_pack_changed)", "group", G, 8 * V + 32,
                      lambda: e.commit_report_into(d_pairs.data_ptr(), G), setup=lambda: e.upload_commit(zero), iters=5)
                timed("commit_changed, nothing changes (K1 + k_pack_changed)", "group", G, 8 * V + 16,
                      lambda: e.commit_report_into(d_pairs.data_ptr(), G), iters=5)
                timed("commit_advance, nothing advances (K1 kAdvance + k_pack_changed)", "group", G, 8 * V + 32,
                      lambda: e.commit_report_into(d_pairs.data_ptr(), G, advance=True), iters=5)
                del d_pairs
                timed("k_heartbeat_commit (K6)", "group", G, 16 * V + 8,
                      lambda: e.heartbeat_commit(download=False))
        del m

    # ---- cfg3: joint configs, S = 6 ---------------------------------------------------------
    from test_gpu_fullsize import cfg3_masks
    G = (1 << 24) >> sh
    m = gu.dist_a(0x5EED0003, 6, G)
    m0, m1 = cfg3_masks(G, 0x5EED0003)
    with raft_amd.Engine(G, 6, stream=stream.cuda_stream) as e:
        e.upload_config(m0, m1)
        e.upload_match(m)
        timed("k_commit_masked<6> cfg3 mix (K2)", "group", G, 8 * 6 + 2 + 8, e.commit)
        e.upload_config(m0, np.zeros_like(m0))
        timed("k_commit_masked<6> no joint group (K2)", "group", G, 8 * 6 + 2 + 8, e.commit)
    del m

    # ---- K2 at its other slot counts: random masks, half of the groups joint (where the f64 network starts to pay) ----
    for Sk in (3, 4, 5, 8):
        name = "k_commit_masked<%d> random masks, half joint (K2)" % Sk
        if args.only and args.only not in name:
            continue
        G = (1 << 24) >> sh
        rng = np.random.default_rng(100 + Sk)
        m = gu.dist_a(0x5EED0030 + Sk, Sk, G)
        k0 = rng.integers(1, 1 << Sk, size=G).astype(np.uint8)
        k1 = np.where(rng.random(G) < 0.5, rng.integers(1, 1 << Sk, size=G), 0).astype(np.uint8)
        with raft_amd.Engine(G, Sk, stream=stream.cuda_stream) as e:
            e.upload_config(k0, k1)
            e.upload_match(m)
            timed(name, "group", G, 8 * Sk + 2 + 8, e.commit)
        del m, k0, k1

    # ---- K2m: a host in the middle of a conf change — 12 shards of 2^20 groups, cfg3 mix, one launch ----------
    if not args.only or args.only in "k_commit_masked_many<6> (K2m)":
        from raft_amd.capi import EngineSet
        Gs = (1 << 20) >> sh
        shards = []
        for k in range(12):  # 12 x 58 MiB: every pass reads cold HBM
            mk = gu.dist_a(0x5EED0300 + k, 6, Gs)
            a0, a1 = cfg3_masks(Gs, 0x5EED0300 + k)
            e = raft_amd.Engine(Gs, 6, stream=stream.cuda_stream)
            e.upload_config(a0, a1)
            e.upload_match(mk)
            shards.append(e)
        es = EngineSet(shards)
        timed("k_commit_masked_many<6> 12 shards of 2^20, cfg3 mix, one launch (K2m)", "group", 12 * Gs, 58, es.commit)

        def one_by_one():
            for e in shards:
                e.commit()
        timed("k_commit_masked<6> the same 12 shards, one launch each (K2)", "group", 12 * Gs, 58, one_by_one)
        d_pairs = [torch.empty(2 * Gs, dtype=torch.int64, device=dev) for _ in shards]
        timed("commit_report_many, 12 masked shards, nothing changes (K2m + k_pack_changed_many)", "group", 12 * Gs, 58 + 8,
              lambda: es.commit_report(max_pairs=Gs, dsts=[d.data_ptr() for d in d_pairs]), iters=5)
        for e in shards:
            e.close()
        del shards, d_pairs

    # ---- wide voters ------------------------------------------------------------------------
    for Sw in (9, 16, 32, 64):
        G = (1 << 22) >> sh
        rng = np.random.default_rng(Sw)
        # below 2^62 - 2^51, the bound of K4r's one-sort joint form (every real log index is far below it)
        m = rng.integers(0, 0x3FF8 << 48, size=(Sw, G), dtype=np.uint64)
        mdt = gu.mask_dtype(Sw)
        full = mdt((1 << Sw) - 1 if Sw < 64 else gu.U64MAX)
        default = "k_commit_masked 256x1 (K2)" if Sw <= 16 else "k_commit_regs (K4r)"
        for flags, kname in ((0, default), (raft_amd.capi.FLAG_WIDE_BITONIC, "k_commit_wide bitonic (K4)")):
            with raft_amd.Engine(G, Sw, stream=stream.cuda_stream, flags=flags) as e:
                e.upload_config(np.full(G, full, dtype=mdt), np.zeros(G, dtype=mdt))
                e.upload_match(m)
                timed("%s S=%d non-joint" % (kname, Sw), "group", G, 8 * Sw + 2 * mdt().itemsize + 8, e.commit)
                e.upload_config(np.full(G, full, dtype=mdt), np.full(G, full >> mdt(1), dtype=mdt))
                timed("%s S=%d joint" % (kname, Sw), "group", G, 8 * Sw + 2 * mdt().itemsize + 8, e.commit)
                if flags == 0 and Sw > 16:
                    # the reference's own bench distribution, rand.Int63n (bench_test.go:32): beyond that bound, so a
                    # joint wave sorts once in vain and then once per half
                    e.upload_match(gu.dist_a(Sw, Sw, G))
                    timed("%s S=%d joint, keys up to 2^63 (sorts per half)" % (kname, Sw), "group", G,
                          8 * Sw + 2 * mdt().itemsize + 8, e.commit)
        del m

    # ---- cfg4: tallies ----------------------------------------------------------------------
    G = (1 << 26) >> sh
    rng = np.random.default_rng(4)
    lut = np.array([2, 2, 1, 0, 0], dtype=np.uint8)
    votes = lut[rng.integers(0, 5, size=(5, G), dtype=np.uint8)]
    bits = rng.integers(0, 32, size=G, dtype=np.uint8)
    with raft_amd.Engine(G, 5, stream=stream.cuda_stream) as e:
        e.set_uniform_config(5)
        e.upload_votes(votes)
        e.upload_active(bits)
        e.upload_acks(bits)
        e.upload_self(np.ones(G, dtype=np.uint8))
        timed("k_tally_votes_small<5> uniform (K3)", "group", G, 5 + 3, e.tally_votes)
        timed("k_tally_bits_small QuorumActive (K3)", "group", G, 2, e.quorum_active)
        timed("k_tally_bits_small ack quorum (K3)", "group", G, 2, e.ack_quorum)
        timed("k_tally_bits_small CheckQuorum + reset (K3)", "group", G, 4, e.check_quorum)
        # K12, RawNode.Tick for every group (raft.go:823-862): ElectionTick 10, HeartbeatTick 1, CheckQuorum; 40 % of the
        # groups led here, timers in a random phase.  Per group: the two elapsed counters read and written (16 B),
        # the randomized timeout, role and flags read (6 B), the action byte written (1 B) = 23 B; RecentActive, the
        # self mask and the config masks are only read by a wavefront that holds a group whose election timeout is up
        e.tick_config(10, 1, True)
        e.upload_tick_state(rng.integers(0, 10, size=G, dtype=np.uint32), np.zeros(G, dtype=np.uint32),
                            rng.integers(10, 20, size=G, dtype=np.uint32),
                            np.where(rng.random(G) < 0.4, 2, 0).astype(np.uint8), np.ones(G, dtype=np.uint8))
        timed("k_tick RawNode.Tick, 40 % leaders, CheckQuorum (K12)", "group", G, 23, lambda: e.tick(counts=False))
        e.upload_active(bits)
        timed("k_tick + the counts read back (K12)", "group", G, 23, e.tick, iters=5)
        # K12m: twelve shards of 2^20 groups, one launch against one launch each
        if not args.only or args.only in "k_tick_many (K12m)":
            Gs = (1 << 20) >> sh
            shards = []
            for k in range(12):
                es_ = raft_amd.Engine(Gs, 5, stream=stream.cuda_stream)
                es_.set_uniform_config(5)
                es_.tick_config(10, 1, True)
                es_.upload_tick_state(rng.integers(0, 10, size=Gs, dtype=np.uint32), np.zeros(Gs, dtype=np.uint32),
                                      rng.integers(10, 20, size=Gs, dtype=np.uint32),
                                      np.where(rng.random(Gs) < 0.4, 2, 0).astype(np.uint8), np.ones(Gs, dtype=np.uint8))
                es_.upload_active(np.full(Gs, 31, dtype=np.uint8))
                es_.upload_self(np.ones(Gs, dtype=np.uint8))
                shards.append(es_)
            eset = S.EngineSet(shards)
            timed("k_tick_many, 12 shards of 2^20 groups, one launch (K12m)", "group", 12 * Gs, 23, lambda: eset.tick(counts=False))

            def one_by_one():
                for x in shards:
                    x.tick(counts=False)
            timed("k_tick, the same 12 shards, one launch each (K12)", "group", 12 * Gs, 23, one_by_one)
            for x in shards:
                x.close()
        m0 = np.full(G, 0b11111, dtype=np.uint8)
        e.upload_config(m0, m0 >> 1, np.zeros(G, dtype=np.uint8))
        timed("k_tally_votes_small<5> masked joint (K3)", "group", G, 5 + 3 + 3, e.tally_votes)
    del votes, bits

    # ---- tallies on more than 8 slots (7 voters + 2 learners; 16 slots masked) ------------------
    for Sw, uniform in ((9, True), (16, False)):
        G = (1 << 24) >> sh
        rng = np.random.default_rng(40 + Sw)
        mdt = gu.mask_dtype(Sw)
        votes = lut[rng.integers(0, 5, size=(Sw, G), dtype=np.uint8)]
        bits = rng.integers(0, 1 << Sw, size=G).astype(mdt)
        with raft_amd.Engine(G, Sw, stream=stream.cuda_stream) as e:
            if uniform:
                e.set_uniform_config(7)
                tag, vb = "uniform 7 voters", 7 + 3
            else:
                full = mdt((1 << Sw) - 1)
                e.upload_config(np.full(G, full, dtype=mdt), np.full(G, full >> mdt(1), dtype=mdt), np.zeros(G, dtype=mdt))
                tag, vb = "masked joint", Sw + 3 + 3 * mdt().itemsize
            e.upload_votes(votes)
            e.upload_active(bits)
            timed("tally_votes S=%d %s (K3 wide)" % (Sw, tag), "group", G, vb, e.tally_votes)
            timed("quorum_active S=%d %s (K3 wide)" % (Sw, tag), "group", G, 1 + mdt().itemsize * (1 if uniform else 4), e.quorum_active)
        del votes, bits

    # ---- cfg5: ack scatter / message ingest, 7 voters + 2 learners ---------------------------
    G = (1 << 22) >> sh
    Sv = 9
    m = gu.dist_a(0x5EED0005, Sv, G)
    N = 4 * G
    rng = np.random.default_rng(5)
    recs = np.zeros(N, dtype=raft_amd.ACK_DTYPE)
    recs["group"] = rng.integers(0, G, size=N, dtype=np.uint32)
    recs["slot"] = rng.integers(0, Sv, size=N, dtype=np.uint32)
    recs["index"] = rng.integers(0, 1 << 62, size=N, dtype=np.uint64)
    msgs = np.zeros(N, dtype=raft_amd.MSG_DTYPE)
    msgs["group"], msgs["slot"], msgs["index"] = recs["group"], recs["slot"], recs["index"]
    msgs["type"] = rng.choice(np.array([4, 9], dtype=np.uint8), size=N, p=[0.7, 0.3])
    msgs["flags"] = 1
    d_recs = torch.from_numpy(recs.view(np.uint8)).to(dev)
    d_msgs = torch.from_numpy(msgs.view(np.uint8)).to(dev)
    with raft_amd.Engine(G, Sv, stream=stream.cuda_stream, flags=raft_amd.capi.FLAG_SCATTER_ATOMIC) as e:
        e.set_uniform_config(7)
        e.upload_match(m)
        timed("k_scatter_acks, random cells (K5, atomic form pinned)", "ack", N, 24, lambda: e.scatter_acks(d_recs.data_ptr(), N),
              atomics_per_unit=1)
        timed("k_ingest_msgs, random cells (K7)", "msg", N, 24 + 4, lambda: e.ingest_msgs(d_msgs.data_ptr(), N),
              atomics_per_unit=1)
        timed("k_commit_uniform<7> over 9 slots (K1)", "group", G, 64, e.commit)
    with raft_amd.Engine(G, Sv, stream=stream.cuda_stream) as e:  # the default: this batch takes the partitioned form
        e.set_uniform_config(7)
        e.upload_match(m)
        timed("rafthip_scatter_acks, partitioned form, random cells (K5p: count + scan + 2 partition passes + apply)", "ack", N, 24,
              lambda: e.scatter_acks(d_recs.data_ptr(), N), atomics_per_unit=1)
    # the same at BASELINE cfg5's own size: 2^24 groups, 2^26 acks (1.2 GB of planes: past the TLB reach of scattered atomics)
    if not sh and (not args.only or args.only in "K5 cfg5 full size"):
        Gf, Nf = 1 << 24, 1 << 26
        gen = torch.Generator(device=dev)
        gen.manual_seed(55)
        big = torch.empty((Nf, 2), dtype=torch.int64, device=dev)
        big[:, 0] = torch.randint(0, Gf, (Nf,), dtype=torch.int64, device=dev, generator=gen) | (
            torch.randint(0, Sv, (Nf,), dtype=torch.int64, device=dev, generator=gen) << 32)
        big[:, 1] = torch.randint(0, 1 << 62, (Nf,), dtype=torch.int64, device=dev, generator=gen)
        for flags, tag in ((raft_amd.capi.FLAG_SCATTER_ATOMIC, "k_scatter_acks (K5, atomic form pinned)"),
                           (0, "rafthip_scatter_acks, partitioned form (K5p)")):
            with raft_amd.Engine(Gf, Sv, stream=stream.cuda_stream, flags=flags) as e:
                e.set_uniform_config(7)
                timed("%s, cfg5 full size: 2^26 acks over 2^24 groups x 9" % tag, "ack", Nf, 24,
                      lambda: e.scatter_acks(big.data_ptr(), Nf), atomics_per_unit=1, iters=5)
        del big
    # serialized responses (K8): same cells as the K7 stream, as gogo-protobuf bytes
    if not args.only or args.only in "k_ingest_wire, random cells (K8)":
        import wire_util as wu
        ids = np.ascontiguousarray((np.arange(Sv, dtype=np.uint64)[:, None] + np.uint64(1)) * np.uint64(0x10001)
                                   + rng.integers(0, 1 << 16, size=G, dtype=np.uint64)[None, :] * np.uint64(1 << 20))
        term = rng.integers(1, 1 << 20, size=G).astype(np.uint64)
        ctx = rng.integers(1, 1 << 63, size=G, dtype=np.uint64)
        wm = m & np.uint64((1 << 40) - 1)
        data, offsets = wu.encode_responses(rng, ids[recs["slot"], recs["group"]], term[recs["group"]],
                                            wm[recs["slot"], recs["group"]] + np.uint64(17), ctx[recs["group"]])
        d_data = torch.from_numpy(data).to(dev)
        d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
        d_grp = torch.from_numpy(recs["group"].astype(np.uint32).view(np.int32)).to(dev)
        per_msg = len(data) / N + 12 + 8 * Sv + 8 + 8 + 4
        with raft_amd.Engine(G, Sv, stream=stream.cuda_stream) as e:
            e.set_uniform_config(7)
            e.upload_match(wm)
            e.upload_ids(ids)
            e.upload_term(term)
            e.upload_read_ctx(ctx, np.full(G, 8, dtype=np.uint8))
            timed("k_ingest_wire, %.1f B messages, random cells (K8)" % (len(data) / N), "msg", N, round(per_msg),
                  lambda: e.ingest_wire(d_data, d_off, d_grp, n=N, want_msgs=False, want_status=False, want_counts=False),
                  atomics_per_unit=1)
        del data, offsets, d_data, d_off, d_grp
    # per-peer replication state (K9): sends + responses in stream order per Progress
    if not args.only or args.only in "k_progress_link + k_progress_apply (K9) (K9p)":
        import progress_util as pu
        cap = 4
        pl = pu.random_planes(rng, Sv, G, cap, 0)
        ev, aux = pu.random_events(rng, N, Sv, G, pl)
        d_ev = torch.from_numpy(ev.view(np.uint8)).to(dev)
        d_aux = torch.from_numpy(aux.view(np.int64)).to(dev)
        d_act = torch.zeros(N, dtype=torch.uint8, device=dev)
        L = raft_amd.load_library()
        for flag, label in ((raft_amd.capi.FLAG_STEP_LINKED, "k_progress_link + k_progress_apply, random cells (K9)"),
                            (raft_amd.capi.FLAG_STEP_PARTITION, "count + scan + 2 partition passes + k_prog_part_apply, the same events (K9p)")):
            with raft_amd.Engine(G, Sv, stream=stream.cuda_stream, flags=flag) as e:
                e.set_uniform_config(7)
                e.progress_config(cap, 0)
                e.upload_match(pl["match"])
                e.upload_progress(pl["pstate"], pl["next"], pl["pending"])
                e.upload_inflights(pl["infl_meta"], pl["infl_bytes"], pl["ring_index"], pl["ring_bytes"])
                e.upload_term_range(np.zeros(G, dtype=np.uint64), pl["match"].max(axis=0) + np.uint64(16))
                # record 16 + aux 8 + link 4+4 + action 1+1 + the 32-byte Progress record read and written
                # back + Match read (written only when it moves) = 106
                timed(label, "event", N, 106,
                      lambda: L.rafthip_step_progress(e._h, d_ev.data_ptr(), d_aux.data_ptr(), N, d_act.data_ptr()),
                      atomics_per_unit=1)
        del pl, ev, aux, d_ev, d_aux, d_act
    # dense tick through the Progress stepper (K9d): the same 4*G acks as a plane + heartbeat masks
    if not args.only or args.only in "k_progress_dense (K9d)":
        import progress_util as pu
        cap = 4
        pl = pu.random_planes(rng, Sv, G, cap, 0)
        pl["pstate"][:] = 1
        plane = np.zeros((Sv, G), dtype=np.uint64)
        np.maximum.at(plane, (recs["slot"], recs["group"]), pl["match"][recs["slot"], recs["group"]] + np.uint64(3))
        d_plane = torch.from_numpy(plane.view(np.int64)).to(dev)
        hbm = rng.integers(0, 1 << Sv, size=G).astype(np.uint16)
        d_hb = torch.from_numpy(hbm.view(np.int16)).to(dev)
        d_act = torch.zeros((Sv, G), dtype=torch.uint8, device=dev)
        L = raft_amd.load_library()
        with raft_amd.Engine(G, Sv, stream=stream.cuda_stream) as e:
            e.set_uniform_config(7)
            e.progress_config(cap, 0)
            e.upload_match(pl["match"])
            e.upload_progress(pl["pstate"], pl["next"], pl["pending"])
            e.upload_inflights(pl["infl_meta"], pl["infl_bytes"], pl["ring_index"], pl["ring_bytes"])
            e.upload_term_range(np.zeros(G, dtype=np.uint64), pl["match"].max(axis=0) + np.uint64(16))
            nev = int(np.count_nonzero(plane)) + int(sum(bin(int(x)).count("1") for x in hbm[:4096])) * (G // 4096)
            # per cell: ack 8 + Match 8 + record 32 read and written + action 1; per group: masks and last index
            timed("k_progress_dense, %.0f %% acked + %.0f %% heartbeat responses (K9d)" % (
                100.0 * np.count_nonzero(plane) / plane.size, 50.0), "cell", Sv * G, 81,
                lambda: L.rafthip_step_dense(e._h, 0, G, d_plane.data_ptr(), G, d_hb.data_ptr(), None, d_act.data_ptr(), G))
            print(json.dumps({"K9d_events_per_launch": nev}), flush=True)
        del pl, plane, d_plane, d_hb, d_act
    # the same 4*G acks as a dense [slot][group] plane (K5d)
    if not args.only or args.only in "k_merge_acks, dense plane (K5d)":
        plane = np.zeros((Sv, G), dtype=np.uint64)
        np.maximum.at(plane, (recs["slot"], recs["group"]), recs["index"])
        d_plane = torch.from_numpy(plane.view(np.int64)).to(dev)
        with raft_amd.Engine(G, Sv, stream=stream.cuda_stream) as e:
            e.set_uniform_config(7)
            # every launch must really move its cells: Match is re-seeded below the plane each time
            low = m & np.uint64((1 << 40) - 1)
            timed("k_merge_acks, dense plane, %.0f %% of cells acked (K5d)" % (100.0 * np.count_nonzero(plane) / plane.size),
                  "cell", Sv * G, 24, lambda: e.merge_acks(d_plane), setup=lambda: e.upload_match(low), iters=5)
        del plane, d_plane
    # bcastAppend as a plan (K11): cfg5 shape, every group led from slot 0, steady state
    if not args.only or args.only in "k_plan_appends (K11)":
        import progress_util as pu
        cap = 4
        pl = pu.random_planes(rng, Sv, G, cap, 0)
        d_plan = torch.empty((Sv, G), dtype=torch.uint8, device=dev)
        d_next = torch.empty((Sv, G), dtype=torch.int64, device=dev)
        L = raft_amd.load_library()
        with raft_amd.Engine(G, Sv, stream=stream.cuda_stream) as e:
            e.set_uniform_config(7)
            e.progress_config(cap, 0)
            e.upload_match(pl["match"])
            e.upload_progress(pl["pstate"], pl["next"], pl["pending"])
            e.upload_inflights(pl["infl_meta"], pl["infl_bytes"], pl["ring_index"], pl["ring_bytes"])
            e.upload_term_range(np.zeros(G, dtype=np.uint64), pl["match"].max(axis=0) + np.uint64(16))
            e.upload_first_index(np.zeros(G, dtype=np.uint64))
            e.upload_active(np.full(G, 0x1FF, dtype=np.uint16))
            e.upload_self(np.ones(G, dtype=np.uint16))
            # per Progress row: the 32-byte record read + plan byte + Next written; per group 8 + 8 + 3 masks, shared by 9 rows
            timed("k_plan_appends, plan + Next out (K11)", "row", Sv * G, 44,
                  lambda: L.rafthip_plan_appends(e._h, 0, G, 1, d_plan.data_ptr(), d_next.data_ptr(), G))
        del pl, d_plan, d_next
    # bcastHeartbeat as serialized bytes (K10): cfg5 shape, every group led here, ids 1..9, 8-byte ReadIndex context on half
    if not args.only or args.only in "k_emit_heartbeats (K10)":
        ids = np.tile(np.arange(1, Sv + 1, dtype=np.uint64)[:, None], (1, G))
        d_out = torch.empty((Sv, G, 72), dtype=torch.uint8, device=dev)
        d_len = torch.empty((Sv, G), dtype=torch.uint8, device=dev)
        L = raft_amd.load_library()
        with raft_amd.Engine(G, Sv, stream=stream.cuda_stream) as e:
            e.set_uniform_config(7)
            e.upload_match(m)
            e.upload_commit(m[0])
            e.upload_ids(ids)
            e.upload_term(np.full(G, 7, dtype=np.uint64))
            e.upload_read_ctx(rng.integers(0, 1 << 63, size=G).astype(np.uint64),
                              np.where(rng.random(G) < 0.5, 8, 0).astype(np.uint8))
            e.upload_self(np.ones(G, dtype=np.uint16))
            # per Progress row: Match 8 + the 72-byte message slot + its length byte; the group's row, commit and
            # self mask ((3+9 -> 12 words) * 8 + 8 + 2) are shared by its 9 rows
            timed("k_emit_heartbeats, 8 messages per group (K10)", "row", Sv * G, 93,
                  lambda: L.rafthip_emit_heartbeats(e._h, 0, G, d_out.data_ptr(), d_len.data_ptr()))
        del ids, d_out, d_len
    # sorted-by-group stream (a host that buckets acks by range)
    order = np.argsort(recs["group"], kind="stable")
    d_recs = torch.from_numpy(np.ascontiguousarray(recs[order]).view(np.uint8)).to(dev)
    with raft_amd.Engine(G, Sv, stream=stream.cuda_stream, flags=raft_amd.capi.FLAG_SCATTER_ATOMIC) as e:
        e.set_uniform_config(7)
        e.upload_match(m)
        timed("k_scatter_acks, group-sorted stream (K5, atomic form pinned)", "ack", N, 24, lambda: e.scatter_acks(d_recs.data_ptr(), N),
              atomics_per_unit=1)

    print("\n| kernel | units/launch | B/unit | avg µs | GB/s | of 8 TB/s | units/s | of the scattered-atomic rate (26.6e9/s) |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %d | %d | %.1f | %.0f | %.1f %% | %.3g | %s |" % (
            r["kernel"], r["units_per_launch"], r["bytes_per_unit"], r["avg_launch_us"], r["achieved_GBs"],
            100 * r["frac_of_8TBs"], r["units_per_s"],
            ("%.0f %%" % (100 * r["frac_of_atomic_rate"])) if "frac_of_atomic_rate" in r else ""))


if __name__ == "__main__":
    main()
