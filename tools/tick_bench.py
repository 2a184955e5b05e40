#!/usr/bin/env python3
"""One whole leader-side tick at BASELINE cfg5 scale, inputs resident on the device:
2^22 groups x (7 voters + 2 learners), 2^24 serialized responses in ->
  rafthip_step_wire        decode + term gate + Progress lookup + Progress / Inflights stepping
  rafthip_commit_advance   quorum index + raftLog.maybeCommit's term guard, advanced pairs out
  rafthip_plan_appends     bcastAppend as a plan: maybeSendAppend's decision per peer
  rafthip_emit_heartbeats  bcastHeartbeat as serialized MsgHeartbeat bytes (Commit = min(Match, committed) per peer)
  rafthip_check_quorum     QuorumActive + RecentActive reset
Per-stage and whole-tick time from HIP events on the engine's stream; a development tool,
bench.py stays the headline."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import golden_util as gu
    import progress_util as pu
    import raft_amd
    import wire_util as wu

    sh = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    S = int(os.environ.get("TICK_SLOTS", "9"))  # 9 = 7 voters + 2 learners (cfg5); 5 = five voters, 3 = three
    G, V, cap = (1 << 22) >> sh, {9: 7}.get(S, S), 4
    N = 4 * G
    mdt = np.uint8 if S <= 8 else np.uint16  # slot masks: one byte up to 8 slots
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    rng = np.random.default_rng(0x5EED0009)
    pl = pu.random_planes(rng, S, G, cap, 0)
    pl["pstate"][:] = 1  # steady state: every peer replicating, nothing paused
    ids = np.ascontiguousarray((np.arange(S, dtype=np.uint64)[:, None] + np.uint64(1)) * np.uint64(0x10001)
                               + rng.integers(0, 1 << 16, size=G, dtype=np.uint64)[None, :] * np.uint64(1 << 20))
    term = rng.integers(1, 1 << 20, size=G).astype(np.uint64)
    ctx = rng.integers(1, 1 << 63, size=G, dtype=np.uint64)
    groups = rng.integers(0, G, size=N).astype(np.uint32)
    slots = rng.integers(0, S, size=N)
    data, offsets = wu.encode_responses(rng, ids[slots, groups], term[groups],
                                        pl["match"][slots, groups] + rng.integers(1, 64, size=N).astype(np.uint64),
                                        ctx[groups])
    d_data = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_grp = torch.from_numpy(groups.view(np.int32)).to(dev)
    d_status = torch.zeros(N, dtype=torch.uint8, device=dev)
    d_act = torch.zeros(N, dtype=torch.uint8, device=dev)
    last = pl["match"].max(axis=0) + np.uint64(128)
    d_plan = torch.empty((S, G), dtype=torch.uint8, device=dev)
    d_next = torch.empty((S, G), dtype=torch.int64, device=dev)
    d_hbytes = torch.empty((S, G, 72), dtype=torch.uint8, device=dev)
    d_hlen = torch.empty((S, G), dtype=torch.uint8, device=dev)
    L = raft_amd.load_library()
    with raft_amd.Engine(G, S, stream=stream.cuda_stream) as e:
        e.set_uniform_config(V)
        e.progress_config(cap, 0)
        e.upload_ids(ids)
        e.upload_term(term)
        e.upload_read_ctx(ctx, np.full(G, 8, dtype=np.uint8))
        e.upload_self(np.ones(G, dtype=mdt))

        def reset():
            e.upload_match(pl["match"])
            e.upload_progress(pl["pstate"], pl["next"], pl["pending"])
            if not os.environ.get("TICK_DRAINED_RINGS"):  # (rounds 1-2 measured every round after the first on drained Inflights)
                e.upload_inflights(pl["infl_meta"], pl["infl_bytes"], pl["ring_index"], pl["ring_bytes"])
            e.upload_commit(np.zeros(G, dtype=np.uint64))
            e.upload_term_range(np.zeros(G, dtype=np.uint64), last)
            e.upload_active(np.ones(G, dtype=mdt))
            e.upload_acks(np.zeros(G, dtype=mdt))

        stages = [
            ("step_wire", lambda: L.rafthip_step_wire(e._h, d_data.data_ptr(), d_off.data_ptr(), d_grp.data_ptr(), N, None,
                                                     d_status.data_ptr(), None, d_act.data_ptr())),
            ("commit_advance", lambda: e.commit_advance(max_pairs=0)),
            ("plan_appends", lambda: L.rafthip_plan_appends(e._h, 0, G, 1, d_plan.data_ptr(), d_next.data_ptr(), G)),
            ("emit_heartbeats", lambda: L.rafthip_emit_heartbeats(e._h, 0, G, d_hbytes.data_ptr(), d_hlen.data_ptr())),
            ("check_quorum", lambda: e.check_quorum()),
        ]
        times = {k: [] for k, _ in stages}
        advanced = 0
        with torch.cuda.stream(stream):
            for it in range(4):
                reset()
                torch.cuda.synchronize()
                for name, fn in stages:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    r = fn()
                    e1.record(stream)
                    e1.synchronize()
                    if it:
                        times[name].append(e0.elapsed_time(e1))
                    if name == "commit_advance":
                        advanced = r[1]
        # the same tick when the transport hands the responses over densely: an ack plane + heartbeat masks
        plane = np.zeros((S, G), dtype=np.uint64)
        np.maximum.at(plane, (slots, groups), pl["match"][slots, groups] + rng.integers(2, 64, size=N).astype(np.uint64))
        d_plane = torch.from_numpy(plane.view(np.int64)).to(dev)
        hbm = rng.integers(0, 1 << S, size=G).astype(mdt)
        d_hb = torch.from_numpy(hbm.view(np.int8 if S <= 8 else np.int16)).to(dev)
        d_dact = torch.zeros((S, G), dtype=torch.uint8, device=dev)
        dense_events = int(np.count_nonzero(plane)) + int(np.unpackbits(hbm.view(np.uint8)).sum())
        stages_d = [("step_dense", lambda: L.rafthip_step_dense(e._h, 0, G, d_plane.data_ptr(), G, d_hb.data_ptr(), None,
                                                                d_dact.data_ptr(), G))] + stages[1:]
        times_d = {k: [] for k, _ in stages_d}
        with torch.cuda.stream(stream):
            for it in range(4):
                reset()
                torch.cuda.synchronize()
                for name, fn in stages_d:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    fn()
                    e1.record(stream)
                    e1.synchronize()
                    if it:
                        times_d[name].append(e0.elapsed_time(e1))
        out_d = {k: round(float(np.median(v)), 3) for k, v in times_d.items()}
        out = {k: round(float(np.median(v)), 3) for k, v in times.items()}
        tick = sum(out.values())
        # what an in-workgroup sort by Progress cell could coalesce ("Why K8/K9 got no LDS pre-sort"):
        # events that share their cell with an earlier event of the same 256-event window
        cells = (slots.astype(np.int64) * G + groups.astype(np.int64))[: (N // 256) * 256].reshape(-1, 256)
        srt = np.sort(cells, axis=1)
        repeats = int((srt[:, 1:] == srt[:, :-1]).sum())
        lines = (cells >> 1)  # two 32-byte Progress records per 64-byte line: events that share a LINE within a window
        srt = np.sort(lines, axis=1)
        line_repeats = int((srt[:, 1:] == srt[:, :-1]).sum())
        print(json.dumps({"groups": G, "slots": S, "messages": N,
                          "events_sharing_a_cell_within_256": repeats, "events_sharing_a_record_line_within_256": line_repeats, "bytes_per_message": round(len(data) / N, 1),
                          "stage_ms": out, "tick_ms": round(tick, 3), "messages_per_s": N / tick * 1e3,
                          "groups_advanced": int(advanced), "applied": int((d_status == 0).sum().item()),
                          "updated": int(((d_act & 2) != 0).sum().item()),
                          "appends_planned": int((d_plan != 0).sum().item()),
                          "heartbeats_out": int((d_hlen != 0).sum().item()),
                          "heartbeat_bytes_out": int(d_hlen.sum(dtype=torch.int64).item()),
                          "dense": {"events": dense_events, "stage_ms": out_d, "tick_ms": round(sum(out_d.values()), 3),
                                    "events_per_s": dense_events / sum(out_d.values()) * 1e3}}))


if __name__ == "__main__":
    main()
