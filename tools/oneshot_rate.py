"""Rate of the stateless entry point from pageable numpy buffers, by batch size (PCIe note).
`fresh`: the result array is allocated per call (what raft_amd.batch_committed_index and a Go `make` do: first-touch
page faults are inside the timed call); `reused`: the caller keeps its result buffer.  `cpu`: host CPU time of the
whole process per call (getrusage, user + system, every thread: the caller, three copiers, the retirer) — run once
as is and once with RAFTHIP_SHOT_SPIN=1 (round 2's yield() loops) to see what waiting costs."""
import os
import resource
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import raft_amd  # noqa: E402
from raft_amd.capi import _check, _ptr, load_library  # noqa: E402

L = load_library()
for logG in (10, 12, 14, 16, 18, 20, 22):
    G, S = 1 << logG, 5
    rng = np.random.default_rng(logG)
    match = rng.integers(0, 1 << 62, size=(S, G), dtype=np.uint64)
    full = np.full(G, 31, dtype=np.uint8)
    raft_amd.batch_committed_index(match, full)
    reps = max(3, min(200, (1 << 24) // G))
    a = time.perf_counter()
    for _ in range(reps):
        raft_amd.batch_committed_index(match, full)
    dt = (time.perf_counter() - a) / reps
    out = np.zeros(G, dtype=np.uint64)
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    a = time.perf_counter()
    for _ in range(reps):
        _check(L.rafthip_batch_committed_index(0, G, S, _ptr(match), _ptr(full), None, _ptr(out)))
    dt2 = (time.perf_counter() - a) / reps
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu = (r1.ru_utime + r1.ru_stime - r0.ru_utime - r0.ru_stime) / reps
    print("%sG=2^%-2d  fresh result: %8.1f us per call  %.3e groups/s   reused result: %8.1f us  %.3e groups/s  (%.1f GB/s over the link)"
          "   cpu: %8.1f us per call = %.2f cores busy"
          % ("[spin] " if os.environ.get("RAFTHIP_SHOT_SPIN") else "", logG, dt * 1e6, G / dt, dt2 * 1e6, G / dt2, G * 49 / dt2 / 1e9,
             cpu * 1e6, cpu / dt2), flush=True)
