#!/usr/bin/env python3
"""Latency of a SMALL leader tick through the C ABI, host buffers in and out (what a host with a light load sees):
n serialized responses -> rafthip_step_wire (status + action bytes back) -> rafthip_commit_advance (advanced pairs
back); the same acks as records -> rafthip_scatter_acks, and as decoded messages -> rafthip_step_progress.
2^20 groups x 5 slots; wall-clock microseconds per call, median of 200.  RAFTHIP_NO_SMALL_ARENA=1 in the environment:
every pageable buffer through its own DMA copy, the form before round 3's arena."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import progress_util as pu
    import raft_amd
    import wire_util as wu
    from raft_amd import capi

    G, S, cap = 1 << 20, 5, 4
    rng = np.random.default_rng(0x5EED000B)
    pl = pu.random_planes(rng, S, G, cap, 0)
    pl["pstate"][:] = 1
    ids = np.ascontiguousarray((np.arange(S, dtype=np.uint64)[:, None] + np.uint64(1)) * np.uint64(0x10001)
                               + rng.integers(0, 1 << 16, size=G, dtype=np.uint64)[None, :] * np.uint64(1 << 20))
    term = rng.integers(1, 1 << 20, size=G).astype(np.uint64)
    ctx = rng.integers(1, 1 << 63, size=G, dtype=np.uint64)
    out = {}
    with raft_amd.Engine(G, S) as e:
        e.set_uniform_config(S)
        e.progress_config(cap, 0)
        e.upload_match(pl["match"])
        e.upload_progress(pl["pstate"], pl["next"], pl["pending"])
        e.upload_ids(ids)
        e.upload_term(term)
        e.upload_read_ctx(ctx, np.full(G, 8, dtype=np.uint8))
        e.upload_term_range(np.zeros(G, dtype=np.uint64), pl["match"].max(axis=0) + np.uint64(1 << 20))
        e.upload_active(np.ones(G, dtype=np.uint8))
        bump = 0
        for n in (100, 1000, 10000, 30000, 100000, 200000):
            ts_step, ts_commit, ts_scatter, ts_prog = [], [], [], []
            for it in range(220):
                groups = rng.integers(0, G, size=n).astype(np.uint32)
                slots = rng.integers(0, S, size=n)
                bump += 64
                data, offsets = wu.encode_responses(rng, ids[slots, groups], term[groups],
                                                    pl["match"][slots, groups] + np.uint64(bump), ctx[groups])
                t0 = time.perf_counter()
                e.step_wire(data, offsets, groups)
                t1 = time.perf_counter()
                e.commit_advance()
                t2 = time.perf_counter()
                acks = np.zeros(n, dtype=capi.ACK_DTYPE)
                acks["group"], acks["slot"] = groups, slots
                acks["index"] = pl["match"][slots, groups] + np.uint64(bump)
                msgs = np.zeros(n, dtype=capi.MSG_DTYPE)
                msgs["group"], msgs["slot"], msgs["type"] = groups, slots, capi.MSG_APP_RESP
                msgs["index"] = acks["index"] + np.uint64(1)
                t3 = time.perf_counter()
                e.scatter_acks(acks)
                t4 = time.perf_counter()
                e.step_progress(msgs)
                t5 = time.perf_counter()
                if it >= 20:
                    ts_step.append(t1 - t0)
                    ts_commit.append(t2 - t1)
                    ts_scatter.append(t4 - t3)
                    ts_prog.append(t5 - t4)
            out[str(n)] = {"step_wire_us": round(1e6 * float(np.median(ts_step)), 1),
                           "commit_advance_us": round(1e6 * float(np.median(ts_commit)), 1),
                           "scatter_acks_us": round(1e6 * float(np.median(ts_scatter)), 1),
                           "step_progress_us": round(1e6 * float(np.median(ts_prog)), 1)}
            print(n, out[str(n)], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
