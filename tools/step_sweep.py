#!/usr/bin/env python3
"""rafthip_step_progress, K9 (lists through the records) against K9p (partition by tile, sort by cell), over batch
sizes: where the library's size threshold (2^18 events) sits.  2^22 groups x 9 slots, events uniformly random over the
cells, device-resident records; microseconds per call by HIP events, median of 5.  argv[1]: log2 of the groups."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import progress_util as pu
    import raft_amd

    G, S, cap = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 22), 9, 4
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    rng = np.random.default_rng(0x5EED000A)
    pl = pu.random_planes(rng, S, G, cap, 0)
    L = raft_amd.load_library()
    out = {}
    engines = {}
    for name, flag in (("K9", raft_amd.capi.FLAG_STEP_LINKED), ("K9p", raft_amd.capi.FLAG_STEP_PARTITION)):
        e = raft_amd.Engine(G, S, stream=stream.cuda_stream, flags=flag)
        e.set_uniform_config(7)
        e.progress_config(cap, 0)
        e.upload_match(pl["match"])
        e.upload_progress(pl["pstate"], pl["next"], pl["pending"])
        e.upload_inflights(pl["infl_meta"], pl["infl_bytes"], pl["ring_index"], pl["ring_bytes"])
        e.upload_term_range(np.zeros(G, dtype=np.uint64), pl["match"].max(axis=0) + np.uint64(16))
        engines[name] = e
    for lg in (14, 16, 18, 19, 20, 21, 22, 23, 24):
        n = 1 << lg
        ev, aux = pu.random_events(rng, n, S, G, pl)
        d_ev = torch.from_numpy(ev.view(np.uint8)).to(dev)
        d_aux = torch.from_numpy(aux.view(np.int64)).to(dev)
        d_act = torch.zeros(n, dtype=torch.uint8, device=dev)
        row = {}
        for name, e in engines.items():
            ts = []
            with torch.cuda.stream(stream):
                for it in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    assert L.rafthip_step_progress(e._h, d_ev.data_ptr(), d_aux.data_ptr(), n, d_act.data_ptr()) == 0
                    e1.record(stream)
                    e1.synchronize()
                    if it >= 2:
                        ts.append(1e3 * e0.elapsed_time(e1))
            row[name] = round(float(np.median(ts)), 1)
        out["2^%d" % lg] = row
        print("2^%d events: K9 %.1f us, K9p %.1f us" % (lg, row["K9"], row["K9p"]), flush=True)
    for e in engines.values():
        e.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
