"""Four calls of the stateless entry point at 2^20 x 5 from pageable numpy planes; with RAFTHIP_SHOT_TRACE=1 the
library prints where each call's time went (the stateless call)."""
import sys, numpy as np
sys.path.insert(0, ".")
import raft_amd
from raft_amd.capi import _check, _ptr, load_library
L = load_library()
G, S = 1 << 20, 5
rng = np.random.default_rng(1)
match = rng.integers(0, 1 << 62, size=(S, G), dtype=np.uint64)
full = np.full(G, 31, dtype=np.uint8)
out = np.zeros(G, dtype=np.uint64)
for _ in range(4):
    _check(L.rafthip_batch_committed_index(0, G, S, _ptr(match), _ptr(full), None, _ptr(out)))
