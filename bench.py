#!/usr/bin/env python3
"""bench.py — headline benchmark of the batched quorum engine.

Metric (BASELINE.json): quorum evals/sec (groups/s) at 1M x 5 voters.

A "step" is one ProgressTracker.Committed pass (K1, k_commit_uniform<5>) over
one batch of 2^20 Raft groups x 5 voters, non-joint, dist A
(quorum/bench_test.go:30-33), with Match planes already resident in HBM.  One
batch is 40 MiB in + 8 MiB out — smaller than the 256 MiB Infinity Cache — so
the steps rotate over --batches distinct resident batches (default 12, 576 MiB)
to make every pass read cold HBM instead of re-reading a cached batch.

N > 1: one process per GPU (torch.distributed / RCCL for the barrier and the
max-over-ranks); groups shard by group-id range, every rank runs the same
steps on its own shard (weak scaling, no data-path collective).  The one
exchange step of the design — the all-gather of changed (group, commit) pairs
— is timed separately after the main loop and reported under "extra".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

G_BATCH = 1 << 20
VOTERS = 5
BYTES_PER_GROUP = VOTERS * 8 + 8  # 5 Match loads + 1 commit store
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec
CPU_EXTRA = {}


def splitmix_batch(seed, S, G):
    import golden_util as gu
    return gu.dist_a(seed, S, G)


def cpu_baseline(match, seconds=10.0):
    """The oracle's reference-shaped loop (per-group id set + id-keyed map,
    CommittedIndex per group) on the host cores.  A baseline, not the product."""
    import oracle
    threads = min(16, os.cpu_count() or 1)
    cb = oracle.CpuBaseline(match, VOTERS)
    used = cb.run(threads)  # warm-up pass
    n, t0 = 0, time.perf_counter()
    while True:
        cb.run(threads)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    ref = cb.out.copy()
    cb.close()
    # two more CPU figures for context: the reference's own micro-benchmark shape
    # (BASELINE cfg1, one thread, ns/op) and the best the host cores do on the engine's SoA layout
    leaf_ns = oracle.bench_leaf_ns(VOTERS)
    k, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < 2.0:
        soa_out, soa_threads = oracle.bench_soa5(match, threads)
        k += 1
    soa_rate = k * match.shape[1] / (time.perf_counter() - t1)
    assert np.array_equal(soa_out, ref)
    CPU_EXTRA.update({"cpu_leaf_ns_per_op_voters5_1thread": round(leaf_ns, 2),
                      "cpu_best_soa_groups_per_s": soa_rate, "cpu_best_soa_threads": soa_threads})
    return {
        "value": n * match.shape[1] / dt, "unit": "groups/s", "cores": int(used), "kind": "port",
        "sample": "%d passes over one 2^20-group x 5-voter dist-A batch (%.1f s wall, %d OpenMP threads), "
                  "oracle/quorum_oracle.c map-shaped CommittedIndex loop" % (n, dt, used),
    }, ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batches", type=int, default=12, help="distinct resident batches to rotate over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import raft_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    # Rehearsal on a one-GPU box (RAFTHIP_BENCH_REHEARSE=1): every rank drives cuda:0 and the
    # collectives run over gloo on host tensors — the multi-rank control flow without RCCL.
    rehearse = os.environ.get("RAFTHIP_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearse else dev  # where collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    stream = torch.cuda.Stream(device=dev)
    engines = []
    mats = []
    for b in range(args.batches):
        # group-id range of this rank's shard of batch b: seeds differ per (rank, batch)
        m = splitmix_batch(0x5EED0000 + 1000 * rank + b, VOTERS, G_BATCH)
        e = raft_amd.Engine(G_BATCH, VOTERS, device=local_rank, stream=stream.cuda_stream)
        e.set_uniform_config(VOTERS)
        e.upload_match(m)
        engines.append(e)
        mats.append(m)
    first_match = mats[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n):
        nb = len(engines)
        for i in range(n):
            engines[i % nb].commit()

    with torch.cuda.stream(stream):
        run(args.warmup)
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        run(args.steps)
        ev1.record(stream)
        barrier()
        t1 = time.perf_counter()
    wall = t1 - t0
    gpu_ms = ev0.elapsed_time(ev1)

    if world > 1:
        t = torch.tensor([wall, gpu_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, gpu_ms = float(t[0]), float(t[1])

    # correctness spot check of what was timed (oracle = checker only)
    import oracle
    got = engines[0].read_commit()
    idx = np.arange(0, G_BATCH, 4099)
    want = oracle.batch_committed(np.ascontiguousarray(first_match[:, idx]), uniform_voters=VOTERS)
    if not np.array_equal(got[idx], want):
        raise SystemExit("bench: GPU result differs from the oracle")

    extra = {}
    # single-launch latency (launch + kernel + sync), the floor a host sees at G = 2^20
    with torch.cuda.stream(stream):
        lat = []
        for i in range(50):
            torch.cuda.synchronize()
            a = time.perf_counter()
            engines[i % len(engines)].commit()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - a)
    extra["single_launch_sync_us"] = round(1e6 * float(np.median(lat)), 2)

    # the same pass when one engine's planes stay cache-resident between ticks (48 MiB < the
    # 256 MiB Infinity Cache): what a host with ONE 2^20-group engine per GPU sees; not `value`
    with torch.cuda.stream(stream):
        ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(50):
            engines[0].commit()
        ev2.record(stream)
        for _ in range(500):
            engines[0].commit()
        ev3.record(stream)
        ev3.synchronize()
    extra["same_batch_avg_launch_us"] = round(1e3 * ev2.elapsed_time(ev3) / 500, 3)

    # HIP-graph-amortised launch (SURVEY.md §8d caveat): one rotation over all batches captured
    # on the engine stream and replayed — the per-launch host cost is gone, what remains is the
    # kernel plus the kernel-to-kernel boundary on the device
    try:
        if world > 1:  # single-GPU measurement extra: no stream capture next to a live RCCL communicator
            raise RuntimeError("skipped at n_gpus > 1")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for e in engines:
                e.commit()
        with torch.cuda.stream(stream):
            for _ in range(5):
                graph.replay()
            ev6, ev7 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(1, 1200 // len(engines))
            ev6.record(stream)
            for _ in range(reps):
                graph.replay()
            ev7.record(stream)
            ev7.synchronize()
        extra["hip_graph_avg_launch_us"] = round(1e3 * ev6.elapsed_time(ev7) / (reps * len(engines)), 3)
        del graph
    except Exception as ex:  # capture is a measurement extra, never the headline
        extra["hip_graph_avg_launch_us"] = None
        extra["hip_graph_error"] = str(ex)[:200]

    # the same rotation with every engine on its OWN stream (what rafthip_create gives an engine by
    # default): passes over different engines are independent, so one pass's drain overlaps the
    # next one's ramp-up.  Aggregate throughput of a host with several shards per GPU; not `value`
    # (the headline keeps one stream so that its per-launch duration is the kernel's own).
    if rank == 0 and world == 1:
        own = []
        for m in mats:
            e = raft_amd.Engine(G_BATCH, VOTERS, device=local_rank)
            e.set_uniform_config(VOTERS)
            e.upload_match(m)
            own.append(e)
        for i in range(200):
            own[i % len(own)].commit()
        torch.cuda.synchronize()
        a = time.perf_counter()
        for i in range(args.steps):
            own[i % len(own)].commit()
        torch.cuda.synchronize()
        dt = time.perf_counter() - a
        extra["own_stream_engines_groups_per_s"] = args.steps * G_BATCH / dt
        extra["own_stream_engines_GBs"] = round(args.steps * G_BATCH * BYTES_PER_GROUP / dt / 1e9, 1)
        if not np.array_equal(own[0].read_commit()[idx], want):
            raise SystemExit("bench: own-stream result differs from the oracle")
        for e in own:
            e.close()
    del mats

    # on-box streaming ceiling: a device-to-device copy of 1 GiB, read + write bytes
    if rank == 0:
        a_t = torch.empty(1 << 28, dtype=torch.int32, device=dev)
        b_t = torch.empty_like(a_t)
        with torch.cuda.stream(stream):
            for _ in range(3):
                b_t.copy_(a_t)
            ev4, ev5 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev4.record(stream)
            for _ in range(10):
                b_t.copy_(a_t)
            ev5.record(stream)
            ev5.synchronize()
        extra["dtod_copy_GBs"] = round(10 * 2 * a_t.numel() * 4 / (ev4.elapsed_time(ev5) * 1e-3) / 1e9, 1)
        del a_t, b_t

    # the stateless entry point (what quorum.BatchCommittedIndex binds to): host planes in, host
    # result out, engine created and destroyed per call — the PCIe-inclusive rate, never `value`
    if rank == 0:
        full = np.full(G_BATCH, (1 << VOTERS) - 1, dtype=np.uint8)
        raft_amd.batch_committed_index(first_match, full, device=local_rank)
        a = time.perf_counter()
        for _ in range(3):
            one = raft_amd.batch_committed_index(first_match, full, device=local_rank)
        extra["oneshot_host_buffers_groups_per_s"] = 3 * G_BATCH / (time.perf_counter() - a)
        if not np.array_equal(one[idx], want):
            raise SystemExit("bench: one-shot result differs from the oracle")

    # the one exchange step: changed (group, commit) pairs -> every rank.  An extra, measured after the
    # timed region: a failure here is reported in the line, it does not take the headline with it.
    from raft_amd import multigpu
    try:
        with torch.cuda.stream(stream):
            e0 = engines[0]
            recs = np.zeros(G_BATCH // 8, dtype=raft_amd.ACK_DTYPE)  # acks touching 1/8 of the groups
            rng = np.random.default_rng(rank)
            recs["group"] = rng.integers(0, G_BATCH, size=len(recs))
            recs["slot"] = rng.integers(0, VOTERS, size=len(recs))
            recs["index"] = np.uint64(1) << np.uint64(62)
            e0.scatter_acks(recs)
            multigpu.commit_changed_allgather(e0, rank * G_BATCH, cdev)  # untimed: first-use costs of torch / RCCL
            recs["index"] = np.uint64(1) << np.uint64(63)
            e0.scatter_acks(recs)
            torch.cuda.synchronize()
            a = time.perf_counter()
            allp, n_changed = multigpu.commit_changed_allgather(e0, rank * G_BATCH, cdev)
            torch.cuda.synchronize()
            c = time.perf_counter()
        extra["changed_pairs_rank0"] = int(n_changed)
        extra["commit_changed_allgather_ms"] = round(1e3 * (c - a), 3)
        extra["allgather_pairs_total"] = int(len(allp))
        if not (np.diff(allp["group"].astype(np.int64)) > 0).all():
            extra["commit_changed_allgather_error"] = "gathered pairs are not sorted by global group id"
    except Exception as ex:
        extra["commit_changed_allgather_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:300])

    if rank == 0:
        total_groups = world * args.steps * G_BATCH
        kernel_s = gpu_ms * 1e-3 / args.steps  # average launch duration over the timed region
        achieved = BYTES_PER_GROUP * G_BATCH / kernel_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_commit_uniform5.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "quorum evals/sec (groups/s) at 1M x 5 voters",
            "value": total_groups / wall,
            "unit": "groups/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "ProgressTracker.Committed, 2^20 groups x 5 voters per GPU per step, non-joint, "
                            "dist A (bench_test.go:30-33), rotating over %d HBM-resident batches" % args.batches,
                "groups_per_step_per_gpu": G_BATCH, "voters": VOTERS, "batches": args.batches,
                "sharding": "group-id range, no data-path collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "frac_of_dtod_copy": (achieved / extra["dtod_copy_GBs"]) if extra.get("dtod_copy_GBs") else None,
                "kernel": "k_commit_uniform<5, kStore>", "bytes_per_group": BYTES_PER_GROUP,
                "avg_launch_us": kernel_s * 1e6,
            },
            "extra": extra,
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"], _ = cpu_baseline(first_match)
            out["extra"].update(CPU_EXTRA)
        print(json.dumps(out), flush=True)

    for e in engines:
        e.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
