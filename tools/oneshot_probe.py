import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, golden_util as gu, raft_amd
G, V = 1 << 20, 5
m = gu.dist_a(1, V, G); full = np.full(G, 31, np.uint8)
raft_amd.batch_committed_index(m, full)
def t(f, n=5):
    f(); a = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - a) / n * 1e3
print("oneshot ms", t(lambda: raft_amd.batch_committed_index(m, full)))
def mk():
    e = raft_amd.Engine(G, V); e.close()
print("create+destroy ms", t(mk))
e = raft_amd.Engine(G, V)
print("upload_config ms", t(lambda: e.upload_config(full, np.zeros_like(full))))
print("upload_match ms", t(lambda: e.upload_match(m)))
def cm(): e.commit(); e.synchronize()
print("commit+sync ms", t(cm))
print("read_commit ms", t(lambda: e.read_commit()))
