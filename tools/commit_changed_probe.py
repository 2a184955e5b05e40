import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, golden_util as gu, raft_amd
G, V = 1 << 20, 5
m = gu.dist_a(1, V, G)
e = raft_amd.Engine(G, V); e.set_uniform_config(V); e.upload_match(m)
def t(f, n=200):
    for _ in range(20): f()
    a = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - a) / n * 1e6
def c(): e.commit(); e.synchronize()
print("commit+sync us", t(c))
print("commit_changed (nothing changes) us", t(lambda: e.commit_changed(max_pairs=1024)))
print("commit_advance (nothing advances) us", t(lambda: e.commit_advance(max_pairs=1024)))
recs = np.zeros(1000, dtype=raft_amd.ACK_DTYPE)
rng = np.random.default_rng(1)
k = [0]
def tick():
    recs["group"] = rng.integers(0, G, size=1000); recs["slot"] = rng.integers(0, V, size=1000)
    k[0] += 1; recs["index"] = (1 << 62) + k[0]
    e.scatter_acks(recs); return e.commit_changed(max_pairs=4096)
print("scatter 1000 acks + commit_changed us", t(tick), "changed", tick()[1])
