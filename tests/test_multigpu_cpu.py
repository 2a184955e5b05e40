"""World-size-2 gloo tests of the sharding plumbing and the changed-pair
all-gather (runs on CPU; the per-rank compute is covered by the GPU tests)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from raft_amd import multigpu
from raft_amd.capi import ACK_DTYPE, PAIR_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_everything():
    for total in (1, 7, 8, 1 << 20, (1 << 24) + 3):
        for world in (1, 2, 3, 8):
            nxt = 0
            for r in range(world):
                first, count = multigpu.shard_range(total, r, world)
                assert first == nxt
                nxt += count
                if count:
                    assert multigpu.owner_of(first, total, world) == r
                    assert multigpu.owner_of(first + count - 1, total, world) == r
            assert nxt == total


def test_route_acks():
    rng = np.random.default_rng(0)
    total, world = 1003, 4
    recs = np.zeros(5000, dtype=ACK_DTYPE)
    recs["group"] = rng.integers(0, total, size=len(recs))
    recs["slot"] = rng.integers(0, 5, size=len(recs))
    recs["index"] = rng.integers(0, 1 << 40, size=len(recs))
    parts = multigpu.route_acks(recs, total, world)
    assert sum(len(p) for p in parts) == len(recs)
    for r, p in enumerate(parts):
        first, count = multigpu.shard_range(total, r, world)
        assert (p["group"] < count).all()
        back = p.copy()
        back["group"] += np.uint32(first)
        sel = recs[(recs["group"] >= first) & (recs["group"] < first + count)]
        assert np.array_equal(np.sort(back, order=["group", "slot", "index"]),
                              np.sort(sel, order=["group", "slot", "index"]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeEngine:
    """Stands in for raft_amd.Engine on a GPU-less host: hands out prepared pairs the
    way rafthip_commit_changed does (up to max_pairs copied to dst, total returned)."""

    def __init__(self, G, pairs):
        self.G, self.pairs = G, np.ascontiguousarray(pairs)

    def commit_report_into(self, dst, max_pairs, advance=False):
        import ctypes
        n = min(len(self.pairs), max_pairs)
        ctypes.memmove(dst, self.pairs.ctypes.data, n * PAIR_DTYPE.itemsize)
        return len(self.pairs)


def _worker(rank, world, port, total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = multigpu.shard_range(total, rank, world)
    rng = np.random.default_rng(100 + rank)
    # rank 1 reports nothing in round 0 (ragged / empty payloads)
    for rnd in range(2):
        n = 0 if (rank == 1 and rnd == 0) else int(rng.integers(1, count))
        local_groups = np.sort(rng.choice(count, size=n, replace=False))
        pairs = np.zeros(n, dtype=PAIR_DTYPE)
        pairs["group"] = local_groups
        pairs["commit"] = (local_groups + first) * 3 + rnd  # a function of the GLOBAL id
        if rnd == 0:
            allp = multigpu.allgather_changed(pairs, first)
        else:
            # the engine-fed form bench.py uses: pairs land in the collective's own buffer
            allp, n_local = multigpu.commit_changed_allgather(_FakeEngine(count, pairs), first)
            assert n_local == n
        q.put((rank, rnd, n, allp.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_allgather_changed_gloo_world2():
    world, total = 2, 1001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in range(world * 2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rnd in range(2):
        views = {r: np.frombuffer(b, dtype=PAIR_DTYPE) for r, rr, n, b in got if rr == rnd}
        ns = {r: n for r, rr, n, b in got if rr == rnd}
        assert np.array_equal(views[0], views[1])  # every rank sees the same set
        allp = views[0]
        assert len(allp) == sum(ns.values())
        assert (np.diff(allp["group"].astype(np.int64)) > 0).all()  # global ids, sorted
        assert np.array_equal(allp["commit"], allp["group"] * 3 + rnd)


def _dense_worker(rank, world, port, total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = multigpu.shard_range(total, rank, world)
    stride = (count + 4095) // 4096 * 4096 + (4096 if rank == 1 else 0)  # ranks with different plane strides
    width = multigpu.agree_width(stride)
    plane = (np.arange(count, dtype=np.uint64) + np.uint64(first)) * np.uint64(7)  # a function of the GLOBAL id
    planes = multigpu.allgather_commit_planes(plane, width)
    q.put((rank, width, planes.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_dense_plane_allgather_gloo_world2():
    """The dense form of the exchange at world size 2: the agreed width is the larger stride, every rank receives
    [world][width], and rank r's first `count_r` entries are its groups' commit indexes."""
    world, total = 2, 9001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dense_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    widths = {w for _, w, _ in got}
    assert len(widths) == 1
    width = widths.pop()
    views = [np.frombuffer(b, dtype=np.uint64).reshape(world, width) for _, _, b in got]
    assert np.array_equal(views[0], views[1])
    for r in range(world):
        first, count = multigpu.shard_range(total, r, world)
        assert width >= count
        assert np.array_equal(views[0][r, :count], (np.arange(count, dtype=np.uint64) + np.uint64(first)) * np.uint64(7))


def test_bench_refuses_a_rank_environment_that_disagrees_with_gpus():
    """`bench.py --gpus N` under a WORLD_SIZE that is not N is an error (before anything touches a GPU), not a
    silent single-rank run; without a rank environment and N > 1 it becomes the launcher instead."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3 but --gpus 2" in p.stderr and p.stdout == ""


_HUNG_EXCHANGE = r"""
import json, os, sys, time
sys.path.insert(0, %r)
import bench
from raft_amd import multigpu

multigpu.make_comm = lambda *a, **k: time.sleep(3600)   # a communicator whose rendezvous never completes
extra = {}

def exchange():   # the shape of bench.py's exchange extra: everything behind make_comm never happens
    comm = multigpu.make_comm(None, group_base=0)
    extra["rccl_ranks_seen"] = comm.nranks

hung = bench.guarded_exchange(exchange, extra, 1)
sys.stdout.write(json.dumps({"metric": "m", "value": 1.0, "extra": extra}) + "\n")
sys.stdout.flush()
code = bench.exit_code(extra, int(sys.argv[1]), hung)
os._exit(code) if hung else sys.exit(code)
"""


def test_bench_reports_a_hung_exchange_with_the_line_and_a_nonzero_exit():
    """A stuck collective is the finding a scaling run exists to surface: the JSON line is still printed (the headline
    was measured before the exchange extra), but the process must not end with rc 0."""
    import json
    p = subprocess.run([sys.executable, "-c", _HUNG_EXCHANGE % ROOT, "8"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3, (p.returncode, p.stderr[-500:])
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert "no return within" in line["extra"]["commit_changed_allgather_error"] and "rccl_ranks_seen" not in line["extra"]


def test_bench_exit_code_rules():
    import bench
    assert bench.exit_code({}, 1, False) == 0 and bench.exit_code({}, 8, False) == 0
    assert bench.exit_code({"commit_changed_allgather_error": "x"}, 1, False) == 0   # one rank: an extra, reported in the line
    assert bench.exit_code({"commit_changed_allgather_error": "x"}, 2, False) == 3   # N > 1: the exchange is what is being exercised
    assert bench.exit_code({}, 1, True) == 3
