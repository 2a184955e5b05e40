#!/usr/bin/env python3
"""K5 rate against the size of the Match planes (where the cells live: L2, Infinity
Cache, HBM) at a fixed batch of 2^24 uniformly random acks — development probe."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, raft_amd
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
N, S = 1 << 24, 9
for lg in (14, 16, 18, 20, 21, 22, 24):
    G = 1 << lg
    rng = np.random.default_rng(lg)
    recs = np.zeros(N, dtype=raft_amd.ACK_DTYPE)
    recs["group"] = rng.integers(0, G, size=N, dtype=np.uint32)
    recs["slot"] = rng.integers(0, S, size=N, dtype=np.uint32)
    recs["index"] = rng.integers(0, 1 << 62, size=N, dtype=np.uint64)
    d = torch.from_numpy(recs.view(np.uint8)).to(dev)
    with raft_amd.Engine(G, S, stream=stream.cuda_stream) as e, torch.cuda.stream(stream):
        for _ in range(2):
            e.scatter_acks(d.data_ptr(), N)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            e.scatter_acks(d.data_ptr(), N)
        e1.record(stream)
        e1.synchronize()
        us = e0.elapsed_time(e1) * 100
    print("G=2^%d planes %7.1f MB: %8.1f us per 2^24 acks = %5.1f G acks/s" % (lg, S * G * 8 / 1e6, us, N / us / 1e3), flush=True)
