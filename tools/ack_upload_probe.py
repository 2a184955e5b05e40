#!/usr/bin/env python3
"""End-to-end cost of one tick's acks crossing PCIe: rafthip_scatter_acks on 2^24 records that
live in pageable host memory (what a Go slice is), in page-locked memory (rafthip_host_alloc)
and on the device already."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raft_amd  # noqa: E402

G, S, N = 1 << 22, 9, 1 << 24
rng = np.random.default_rng(1)
recs = np.zeros(N, dtype=raft_amd.ACK_DTYPE)
recs["group"] = rng.integers(0, G, size=N)
recs["slot"] = rng.integers(0, S, size=N)
recs["index"] = rng.integers(0, 1 << 40, size=N)


def t(f, n=5):
    f()
    a = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - a) / n * 1e3


with raft_amd.Engine(G, S) as e:
    def run(src):
        e.scatter_acks(src, N)
        e.synchronize()
    ms = t(lambda: run(recs))
    print("pageable host records: %.2f ms  (%.1f GB/s, %.2e acks/s)" % (ms, N * 16 / ms / 1e6, N / ms * 1e3))
    pin = raft_amd.PinnedArray(N, raft_amd.ACK_DTYPE)
    pin.array[:] = recs
    ms = t(lambda: run(pin.array))
    print("page-locked host records: %.2f ms  (%.1f GB/s, %.2e acks/s)" % (ms, N * 16 / ms / 1e6, N / ms * 1e3))
    import torch
    d = torch.from_numpy(recs.view(np.uint8)).cuda()
    ms = t(lambda: run(d.data_ptr()))
    print("device-resident records: %.2f ms  (%.2e acks/s)" % (ms, N / ms * 1e3))
    pin.free()
