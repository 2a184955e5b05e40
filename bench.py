#!/usr/bin/env python3
"""bench.py — headline benchmark of the batched quorum engine.

Metric (BASELINE.json): quorum evals/sec (groups/s) at 1M x 5 voters.

A "step" is one ProgressTracker.Committed pass (K1, sorting network over 5 Match
planes) over one batch of 2^20 Raft groups x 5 voters, non-joint, dist A
(quorum/bench_test.go:30-33), with Match planes already resident in HBM.  One
batch is 40 MiB in + 8 MiB out — smaller than the 256 MiB Infinity Cache — so the
steps rotate over --batches distinct resident batches (default 12, 576 MiB) to
make every pass read cold HBM.  The batches are the shards a multi-raft host keeps
on one GPU; rafthip_commit_many runs L of them per kernel launch (L steps per
launch, every launch of a run the same L, so the per-kernel average rocprofv3
reports is the figure `roofline` is computed from).

N > 1: one process per GPU.  `python bench.py --gpus N` starts the N ranks itself
(fresh child processes, before anything touches a GPU); under
torch.distributed.run (WORLD_SIZE set) the process is one of the ranks.  Groups
shard by group-id range, every rank runs the same steps on its own shard (weak
scaling, no data-path collective); torch.distributed / RCCL carries the barrier and
the max-over-ranks.  The one exchange step of the design — the all-gather of
changed (group, commit) pairs, rafthip_commit_allgather over RCCL — is timed
separately after the main loop and reported under "extra".

--config cfg3 | cfg5 run the two BASELINE configs that are defined across GPUs
(16M joint groups sharded; MsgAppResp scatter -> commit with acks routed by group
range) as the workload; at N = 1 the default run also reports both under "extra".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import socket
import subprocess
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
ATOMIC_RATE_PEAK = 26.6e9         # scattered 8-byte atomics/s, measured (tools/scatter_probe.py)
CFG3_GROUPS = 1 << 24             # BASELINE configs[2]: 16M groups, joint, across the GPUs
CFG5_GROUPS = 1 << 24             # BASELINE configs[4]: 7 voters + 2 learners, 4 acks per group
EXCHANGE_TIMEOUT_S = int(os.environ.get("RAFTHIP_BENCH_EXCHANGE_TIMEOUT", "120"))  # the exchange extra may not hold the line back longer
LAUNCH_TIMEOUT_S = int(os.environ.get("RAFTHIP_BENCH_LAUNCH_TIMEOUT", "1500"))      # launch_ranks: longest wait for rank 0
EXIT_EXCHANGE_FAILED = 3          # the JSON line was written, but the one collective hung or failed: the record must not read rc 0
CPU_EXTRA = {}


def splitmix_batch(seed, S, G):
    import golden_util as gu
    return gu.dist_a(seed, S, G)


# ---------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a rank environment
# ---------------------------------------------------------------------------------------------

def launch_ranks(n, argv):
    """Start n fresh rank processes of this script (the parent never touches a GPU, never execs
    after GPU init), pass rank 0's stdout through, return the worst exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RAFTHIP_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    try:
        out, _ = procs[0].communicate(timeout=LAUNCH_TIMEOUT_S)
        rcs = [procs[0].returncode]
    except subprocess.TimeoutExpired:
        procs[0].kill()  # exactly the children this function started, nothing by pattern
        out, _ = procs[0].communicate()
        rcs = [124]
    deadline = time.time() + (120 if rcs[0] != 124 else 1)
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()  # exactly the child we started
            rcs.append(-9)
    sys.stdout.write(out.decode("utf-8", "replace"))
    sys.stdout.flush()
    return max((abs(rc) for rc in rcs), default=0)


# ---------------------------------------------------------------------------------------------
# the exchange extra must never hide a stuck or failed collective behind exit code 0
# ---------------------------------------------------------------------------------------------

def guarded_exchange(fn, extra, timeout_s):
    """Run fn() on a helper thread.  True when it has not returned after timeout_s (a collective that
    never completes): the caller still prints its line, then leaves with exit_code()."""
    import threading
    th = threading.Thread(target=fn, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        extra["commit_changed_allgather_error"] = "no return within %d s" % timeout_s
        return True
    return False


def exit_code(extra, world, hung):
    """What the process ends with AFTER the JSON line: a hung exchange, or — with more than one rank, where
    the exchange is the thing a scaling run exists to exercise — a failed one, is not success."""
    if hung or (world > 1 and extra.get("commit_changed_allgather_error")):
        return EXIT_EXCHANGE_FAILED
    return 0


# ---------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------

def usable_cpus():
    """CPUs this process may use: the affinity mask, cut by a cgroup CPU quota when one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = max(1, min(n, int(math.ceil(int(quota) / int(period)))))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(match, seconds=10.0):
    """The oracle's reference-shaped loop (per-group id set + id-keyed map, CommittedIndex per
    group) on the host cores.  A baseline, not the product.  Thread counts from the box's
    OMP_NUM_THREADS share up to every host core are tried briefly; the fastest one runs the sample."""
    import oracle
    host_cores = os.cpu_count() or 1
    usable = usable_cpus()
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    tried = {}
    cb = oracle.CpuBaseline(match, VOTERS)
    for t in sorted({c for c in (omp, 16, usable, host_cores) if 0 < c <= host_cores}):
        cb.run(t)  # warm-up pass
        k, a = 0, time.perf_counter()
        while time.perf_counter() - a < 1.0:
            cb.run(t)
            k += 1
        tried[t] = k * match.shape[1] / (time.perf_counter() - a)
    threads = max(tried, key=tried.get)
    used = cb.run(threads)
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
        "host_cores": host_cores, "usable_cores": usable, "threads": int(used),
        "threads_tried_groups_per_s": {str(t): round(v) for t, v in sorted(tried.items())},
        "sample": "%d passes over one 2^20-group x 5-voter dist-A batch (%.1f s wall, %d OpenMP threads — the fastest of "
                  "the thread counts tried, %d host cores), oracle/quorum_oracle.c map-shaped CommittedIndex loop"
                  % (n, dt, used, host_cores),
    }, ref


# ---------------------------------------------------------------------------------------------
# the BASELINE configs that are defined across GPUs
# ---------------------------------------------------------------------------------------------

def _timed(stream, fn, reps):
    import torch
    with torch.cuda.stream(stream):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for _ in range(reps):
            fn()
        ev1.record(stream)
        ev1.synchronize()
    return ev0.elapsed_time(ev1) * 1e-3 / reps


def run_cfg3(raft_amd, torch, dev, local_rank, stream, rank, world, reps):
    """BASELINE configs[2] / SURVEY.md §8(d) cfg3: 2^24 groups over the ranks, S = 6 slots:
    80 % joint (incoming 5, outgoing 5, overlap 4: slot 4 replaced by slot 5), 10 % non-joint
    V = 5, 10 % asymmetric (outgoing 3 -> incoming 5).  Returns (figures, engine, seconds per pass)."""
    import oracle
    from raft_amd import multigpu
    first, G = multigpu.shard_range(CFG3_GROUPS, rank, world)
    S = 6
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0003 + rank)
    match = torch.randint(0, (1 << 62), (S, G), dtype=torch.int64, device=dev, generator=gen)
    kind = torch.randint(0, 10, (G,), device=dev, generator=gen)
    in_mask = torch.full((G,), 0b011111, dtype=torch.uint8, device=dev)
    out_mask = torch.where(kind < 8, 0b101111, 0)
    out_mask = torch.where(kind == 9, 0b000111, out_mask).to(torch.uint8).contiguous()
    e = raft_amd.Engine(G, S, device=local_rank, stream=stream.cuda_stream)
    try:
        e.upload_config(in_mask, out_mask)
        e.upload_match(match)
        with torch.cuda.stream(stream):
            e.commit()
        torch.cuda.synchronize()
        n = min(G, 8192)  # oracle = checker, on a slice
        want = oracle.batch_committed(match[:, :n].cpu().numpy().astype(np.uint64), in_mask[:n].cpu().numpy(),
                                      out_mask[:n].cpu().numpy())
        if not np.array_equal(e.read_commit(0, n), want):
            raise SystemExit("bench cfg3: GPU result differs from the oracle")
        del match, kind
        sec = _timed(stream, e.commit, reps)
        bytes_per_group = 8 * S + 2 + 8  # 58 B at S = 6
        return {"groups_total": CFG3_GROUPS, "groups_this_rank": G, "slots": S, "ms_per_pass": round(sec * 1e3, 4),
                "groups_per_s_this_rank": G / sec, "GBs": round(G * bytes_per_group / sec / 1e9, 1),
                "frac_of_hbm_peak": round(G * bytes_per_group / sec / 1e9 / HBM_PEAK_GBS, 4),
                "kernel": "k_commit_masked<6>", "bytes_per_group": bytes_per_group}, e, sec
    except BaseException:
        e.close()
        raise


def run_cfg2(raft_amd, torch, dev, local_rank, stream, reps):
    """BASELINE configs[1]'s sibling / SURVEY.md §8(d) cfg2: 2^20 groups x 3 voters, non-joint, dist A; like the
    headline it rotates over 12 resident shards in one rafthip_commit_many launch so that every pass reads HBM."""
    import oracle
    from raft_amd.capi import EngineSet
    G, V, nb = 1 << 20, 3, 12
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0002)
    engines = []
    try:
        for k in range(nb):
            m = torch.randint(0, (1 << 63) - 1, (V, G), dtype=torch.int64, device=dev, generator=gen)
            e = raft_amd.Engine(G, V, device=local_rank, stream=stream.cuda_stream)
            e.set_uniform_config(V)
            e.upload_match(m)
            engines.append(e)
            if k == 0:
                first = m[:, :8192].cpu().numpy().astype(np.uint64)
        es = EngineSet(engines)
        with torch.cuda.stream(stream):
            es.commit()
        torch.cuda.synchronize()
        if not np.array_equal(engines[0].read_commit(0, 8192), oracle.batch_committed(first, uniform_voters=V)):
            raise SystemExit("bench cfg2: GPU result differs from the oracle")
        sec = _timed(stream, es.commit, reps) / nb  # per 2^20-group pass
        b = 8 * V + 8  # 32 B per group
        return {"groups_per_pass": G, "voters": V, "shards_per_launch": nb, "us_per_pass": round(sec * 1e6, 3),
                "groups_per_s": G / sec, "GBs": round(G * b / sec / 1e9, 1), "frac_of_hbm_peak": round(G * b / sec / 1e9 / HBM_PEAK_GBS, 4),
                "kernel": "k_commit_uniform_many<3>", "bytes_per_group": b}
    finally:
        for e in engines:
            e.close()


def run_cfg4(raft_amd, torch, dev, local_rank, stream, reps):
    """BASELINE configs[3] / SURVEY.md §8(d) cfg4: 2^26 groups x 5 voters, the three tallies: TallyVotes over
    vote cells (40 % granted / 20 % rejected / 40 % missing), QuorumActive over RecentActive bits, the ReadIndex
    ack quorum.  10 B per group for the three (5 + 1 vote bytes, 1 + 1, 1 + 1)."""
    import oracle
    G, V = 1 << 26, 5
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0004)
    lut = torch.tensor([2, 2, 1, 0, 0], dtype=torch.uint8, device=dev)
    votes = lut[torch.randint(0, 5, (V, G), device=dev, generator=gen)]
    bits = torch.randint(0, 32, (G,), device=dev, generator=gen).to(torch.uint8)
    e = raft_amd.Engine(G, V, device=local_rank, stream=stream.cuda_stream)
    try:
        e.set_uniform_config(V)
        e.upload_votes(votes)
        e.upload_active(bits)
        e.upload_acks(bits)
        with torch.cuda.stream(stream):
            e.tally_votes()
            e.quorum_active()
            e.ack_quorum()
        torch.cuda.synchronize()
        n = 8192
        res, _, _ = e.read_vote_result(0, n)
        ores, _, _ = oracle.batch_tally_votes(votes[:, :n].cpu().numpy(), uniform_voters=V)
        hb = bits[:n].cpu().numpy()
        if not (np.array_equal(res, ores) and np.array_equal(e.read_quorum_active(0, n), oracle.batch_quorum_active(hb, V, uniform_voters=V))
                and np.array_equal(e.read_ack_quorum(0, n), oracle.batch_ack_quorum(hb, V, uniform_voters=V))):
            raise SystemExit("bench cfg4: GPU result differs from the oracle")
        del votes, bits
        t = [_timed(stream, f, reps) for f in (e.tally_votes, e.quorum_active, e.ack_quorum)]
        sec = sum(t)
        return {"groups": G, "voters": V, "tally_votes_ms": round(t[0] * 1e3, 4), "quorum_active_ms": round(t[1] * 1e3, 4),
                "ack_quorum_ms": round(t[2] * 1e3, 4), "groups_per_s_three_tallies": G / sec,
                "GBs": round(G * 10 / sec / 1e9, 1), "frac_of_hbm_peak": round(G * 10 / sec / 1e9 / HBM_PEAK_GBS, 4),
                "bytes_per_group": 10}
    finally:
        e.close()


def run_cfg5(raft_amd, torch, dev, local_rank, stream, rank, world, reps):
    """BASELINE configs[4] / SURVEY.md §8(d) cfg5: 2^24 groups over the ranks, 7 voters + 2 learners
    (S = 9 rows, quorum over 7); per pass 4 acks per group, uniformly random cell, index = the cell's
    old Match + U[-3, 61) (about 5 % stale), routed to the owning rank by group range; then the commit pass."""
    from raft_amd import multigpu
    first, G = multigpu.shard_range(CFG5_GROUPS, rank, world)
    S, V = 9, 7
    n_acks = 4 * G
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED0005 + rank)
    base = 1 << 30
    e = raft_amd.Engine(G, S, device=local_rank, stream=stream.cuda_stream)
    e.set_uniform_config(V)
    e.upload_match(torch.full((S, G), base, dtype=torch.int64, device=dev))
    # rafthip_ack records {u32 group, u32 slot, u64 index} as int64 pairs, built on the device:
    # what a host has after routing the node's ack stream by group range (multigpu.route_acks)
    grp = torch.randint(0, G, (n_acks,), dtype=torch.int64, device=dev, generator=gen)
    slot = torch.randint(0, S, (n_acks,), dtype=torch.int64, device=dev, generator=gen)
    recs = torch.empty((n_acks, 2), dtype=torch.int64, device=dev)
    recs[:, 0] = grp | (slot << 32)
    del grp, slot
    bump = [0]

    def one_pass():
        # every pass acknowledges a little further than the last one, so the atomics really update
        recs[:, 1] = base + bump[0] * 64 + torch.randint(-3, 61, (n_acks,), dtype=torch.int64, device=dev, generator=gen)
        bump[0] += 1

    try:
        import oracle
        with torch.cuda.stream(stream):
            one_pass()
            e.scatter_acks(recs.data_ptr(), n_acks)
            e.commit()
        torch.cuda.synchronize()
        n = min(G, 8192)  # oracle = checker, on a slice: the commit pass over the scattered planes
        if not np.array_equal(e.read_commit(0, n), oracle.batch_committed(e.read_match(0, n), uniform_voters=V)):
            raise SystemExit("bench cfg5: GPU result differs from the oracle")
        t_sc = t_co = 0.0
        for _ in range(reps):
            with torch.cuda.stream(stream):
                one_pass()
            torch.cuda.synchronize()
            t_sc += _timed(stream, lambda: e.scatter_acks(recs.data_ptr(), n_acks), 1)
            t_co += _timed(stream, e.commit, 1)
        t_sc /= reps
        t_co /= reps
        sec = t_sc + t_co
        # which form rafthip_scatter_acks took (its own rule, rafthip.hip part_wanted): the partitioned one (K5p) for
        # device-resident batches of >= 2^21 records that touch at least a sixth of the cells, while the engine has
        # at most 32768 tiles of 1024 groups; and the other form on a twin engine pinned to atomics, for the A/B
        stride = (G + 4095) // 4096 * 4096
        partitioned = n_acks >= (1 << 21) and 6 * n_acks >= S * stride and stride // 1024 <= 32768
        t_at = None
        try:
            from raft_amd.capi import FLAG_SCATTER_ATOMIC
            with raft_amd.Engine(G, S, device=local_rank, stream=stream.cuda_stream, flags=FLAG_SCATTER_ATOMIC) as ea:
                ea.set_uniform_config(V)
                ea.upload_match(torch.full((S, G), base, dtype=torch.int64, device=dev))
                t_at = 0.0
                for _ in range(min(reps, 2) + 1):
                    with torch.cuda.stream(stream):
                        one_pass()
                    torch.cuda.synchronize()
                    t_at = _timed(stream, lambda: ea.scatter_acks(recs.data_ptr(), n_acks), 1)  # the last one counts (first: warm-up)
        except Exception:
            t_at = None
        return {"groups_total": CFG5_GROUPS, "groups_this_rank": G, "acks_this_rank": n_acks, "slots": S, "voters": V,
                "scatter_form": "partitioned (K5p: count, scan, 2 partition passes, tile apply in LDS)" if partitioned else "atomic (K5)",
                "scatter_atomic_form_ms": None if t_at is None else round(t_at * 1e3, 4),
                "scatter_ms": round(t_sc * 1e3, 4), "commit_ms": round(t_co * 1e3, 4),
                "groups_per_s_this_rank": G / sec, "acks_per_s_this_rank": n_acks / t_sc,
                "scatter_frac_of_hbm_peak": round(n_acks * 24 / t_sc / 1e9 / HBM_PEAK_GBS, 4),
                "scatter_frac_of_atomic_rate": round(n_acks / t_sc / ATOMIC_RATE_PEAK, 4),
                "commit_frac_of_hbm_peak": round(G * 64 / t_co / 1e9 / HBM_PEAK_GBS, 4)}, e, sec
    except BaseException:
        e.close()
        raise


# ---------------------------------------------------------------------------------------------

def run_tick(raft_amd, torch, dev, local_rank, stream, reps):
    """One leader-side tick at BASELINE cfg5's per-GPU scale (tools/tick_bench.py): 2^22 groups x
    (7 voters + 2 learners), 2^24 serialized responses resident on the device -> rafthip_step_wire (decode, term gate,
    Progress lookup, Progress / Inflights stepping; the partitioned stepper K9p at this size), rafthip_commit_advance,
    rafthip_plan_appends, rafthip_emit_heartbeats, rafthip_check_quorum.  The first round is checked against the
    oracle on the sub-stream of the first 4096 groups: status, action bytes, Progress state, Match."""
    import progress_util as pu
    import wire_util as wu
    from oracle import pyoracle as po
    G, S, V, cap = 1 << 22, 9, 7, 4
    N = 4 * G
    rng = np.random.default_rng(0x5EED0009)
    pl = pu.random_planes(rng, S, G, cap, 0)
    pl["pstate"][:] = 1  # steady state: every peer replicating, nothing paused
    ids = np.ascontiguousarray((np.arange(S, dtype=np.uint64)[:, None] + np.uint64(1)) * np.uint64(0x10001)
                               + rng.integers(0, 1 << 16, size=G, dtype=np.uint64)[None, :] * np.uint64(1 << 20))
    term = rng.integers(1, 1 << 20, size=G).astype(np.uint64)
    ctx = rng.integers(1, 1 << 63, size=G, dtype=np.uint64)
    groups = rng.integers(0, G, size=N).astype(np.uint32)
    slots = rng.integers(0, S, size=N)
    data, offsets = wu.encode_responses(rng, ids[slots, groups], term[groups],
                                        pl["match"][slots, groups] + rng.integers(1, 64, size=N).astype(np.uint64), ctx[groups])
    last = pl["match"].max(axis=0) + np.uint64(128)
    d_data = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_grp = torch.from_numpy(groups.view(np.int32)).to(dev)
    d_status = torch.zeros(N, dtype=torch.uint8, device=dev)
    d_act = torch.zeros(N, dtype=torch.uint8, device=dev)
    d_plan = torch.empty((S, G), dtype=torch.uint8, device=dev)
    d_next = torch.empty((S, G), dtype=torch.int64, device=dev)
    d_hbytes = torch.empty((S, G, 72), dtype=torch.uint8, device=dev)
    d_hlen = torch.empty((S, G), dtype=torch.uint8, device=dev)
    L = raft_amd.load_library()
    ones = np.ones(G, dtype=np.uint16)
    with raft_amd.Engine(G, S, device=local_rank, stream=stream.cuda_stream) as e:
        e.set_uniform_config(V)
        e.progress_config(cap, 0)
        e.upload_ids(ids)
        e.upload_term(term)
        e.upload_read_ctx(ctx, np.full(G, 8, dtype=np.uint8))
        e.upload_self(ones)

        def reset():
            e.upload_match(pl["match"])
            e.upload_progress(pl["pstate"], pl["next"], pl["pending"])
            e.upload_inflights(pl["infl_meta"], pl["infl_bytes"], pl["ring_index"], pl["ring_bytes"])
            e.upload_commit(np.zeros(G, dtype=np.uint64))
            e.upload_term_range(np.zeros(G, dtype=np.uint64), last)
            e.upload_active(ones)
            e.upload_acks(np.zeros(G, dtype=np.uint16))

        def ok(rc):
            if rc != 0:
                raise RuntimeError(L.rafthip_last_error().decode())

        stages = [
            ("step_wire", lambda: ok(L.rafthip_step_wire(e._h, d_data.data_ptr(), d_off.data_ptr(), d_grp.data_ptr(), N, None,
                                                         d_status.data_ptr(), None, d_act.data_ptr()))),
            ("commit_advance", lambda: e.commit_advance(max_pairs=0)),
            ("plan_appends", lambda: ok(L.rafthip_plan_appends(e._h, 0, G, 1, d_plan.data_ptr(), d_next.data_ptr(), G))),
            ("emit_heartbeats", lambda: ok(L.rafthip_emit_heartbeats(e._h, 0, G, d_hbytes.data_ptr(), d_hlen.data_ptr()))),
            ("check_quorum", lambda: e.check_quorum()),
        ]
        times = {k: [] for k, _ in stages}
        checked = 0
        with torch.cuda.stream(stream):
            for it in range(reps + 1):
                reset()
                torch.cuda.synchronize()
                for name, fn in stages:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    fn()
                    e1.record(stream)
                    e1.synchronize()
                    if it:
                        times[name].append(e0.elapsed_time(e1))
                    if it == 0 and name == "step_wire":
                        # oracle = checker: the responses of the first n groups, in stream order
                        n = 4096
                        sel = np.flatnonzero(groups < n)
                        o64 = offsets.astype(np.int64)
                        sub = np.concatenate([data[o64[i]:o64[i + 1]] for i in sel])
                        sub_off = np.zeros(len(sel) + 1, dtype=np.uint64)
                        np.cumsum((o64[sel + 1] - o64[sel]).astype(np.uint64), out=sub_off[1:])
                        spl = {k: np.ascontiguousarray(v[..., :n]) for k, v in pl.items()}
                        act_m, ack_m = ones[:n].copy(), np.zeros(n, dtype=np.uint16)
                        wm, wa, wst, _ = po.decode_wire_msgs(np.ascontiguousarray(ids[:, :n]), term[:n].copy(), ctx[:n].copy(),
                                                             np.full(n, 8, dtype=np.uint8), sub, sub_off,
                                                             np.ascontiguousarray(groups[sel]))
                        want = po.step_progress(spl, cap, 0, act_m, ack_m, last[:n].copy(), np.zeros(n, dtype=np.uint64), wm, wa)
                        sel_t = torch.from_numpy(sel).to(dev)
                        got_act = d_act[sel_t].cpu().numpy()
                        got_st = d_status[sel_t].cpu().numpy()
                        got = e.read_progress(0, n)
                        if not (np.array_equal(got_st, wst) and np.array_equal(got_act, want)
                                and all(np.array_equal(got[k], spl[k]) for k in ("pstate", "next", "pending", "infl_meta", "infl_bytes"))
                                and np.array_equal(e.read_match(0, n), spl["match"])):
                            raise SystemExit("bench tick: GPU result differs from the oracle")
                        checked = int(len(sel))
        out = {k: round(float(np.median(v)), 4) for k, v in times.items()}
        tick = sum(out.values())
        return {"groups": G, "slots": S, "messages": N, "bytes_per_message": round(len(data) / N, 1), "stage_ms": out,
                "tick_ms": round(tick, 4), "messages_per_s": N / tick * 1e3,
                "applied": int((d_status == 0).sum().item()), "updated": int(((d_act & 2) != 0).sum().item()),
                "heartbeats_out": int((d_hlen != 0).sum().item()),
                "stepper": "K9p (partition by tile, sort by cell, apply); rafthip_step_wire picks it from 2^22 events over state beyond the Infinity Cache",
                "oracle_checked_messages": checked}


def run_raw_tick(raft_amd, torch, dev, local_rank, stream, reps):
    """RawNode.Tick for every group of a GPU's share of BASELINE cfg5 (2^24 groups x (7 voters + 2 learners)):
    rafthip_tick (K12: tickElection / tickHeartbeat with the CheckQuorum step fused in, raft.go:823-862), ElectionTick
    10, HeartbeatTick 1, CheckQuorum on, 40 % of the groups led here, timers in a random phase.  The first tick is
    checked against the oracle on the first 2^18 groups (timers, role, flags, RecentActive, action bytes); then the
    rare events (campaigns, step-downs, transfer timeouts) are fetched as an ordered list."""
    from oracle import pyoracle as po
    G, S, V, et, ht, n = 1 << 24, 9, 7, 10, 1, 1 << 18
    rng = np.random.default_rng(0x5EED000C)
    el = rng.integers(0, et, size=G, dtype=np.uint32)
    hb = np.zeros(G, dtype=np.uint32)
    rt = rng.integers(et, 2 * et, size=G, dtype=np.uint32)
    role = np.where(rng.random(G) < 0.4, 2, 0).astype(np.uint8)
    flags = np.ones(G, dtype=np.uint8)
    active = rng.integers(0, 1 << S, size=G).astype(np.uint16)
    self_mask = np.ones(G, dtype=np.uint16)
    with raft_amd.Engine(G, S, device=local_rank, stream=stream.cuda_stream) as e:
        e.set_uniform_config(V)
        e.tick_config(et, ht, True)
        e.upload_tick_state(el, hb, rt, role, flags)
        e.upload_active(active)
        e.upload_self(self_mask)
        with torch.cuda.stream(stream):
            e.tick(counts=False)
            want = [a[:n].copy() for a in (el, hb, rt, role, flags, active)]
            c0 = time.perf_counter()
            want_act = po.batch_tick(want[0], want[1], want[2], want[3], want[4], et, ht, True, S, want[5],
                                     self_mask[:n].copy(), uniform_voters=V)
            t_cpu = time.perf_counter() - c0
            got = e.read_tick_state(0, n) + (e.read_active(0, n),)
            if not (np.array_equal(e.read_tick_actions(0, n), want_act) and all(np.array_equal(a, b) for a, b in zip(got, want))):
                raise SystemExit("bench raw tick: GPU result differs from the oracle")
            # steady state for the timed ticks: every peer has answered since the last CheckQuorum
            e.upload_active(np.full(G, (1 << S) - 1, dtype=np.uint16))
            t_tick = _timed(stream, lambda: e.tick(counts=False), reps)
            t0 = time.perf_counter()
            counts = e.tick()
            rare = raft_amd.capi.TICK_HUP | raft_amd.capi.TICK_STEP_DOWN | raft_amd.capi.TICK_ABORT_TRANSFER
            ev, n_ev = e.read_tick_events(rare, max_events=1 << 22)
            t_ev = time.perf_counter() - t0
        return {"groups": G, "slots": S, "election_tick": et, "heartbeat_tick": ht, "check_quorum": True,
                "tick_ms": round(t_tick * 1e3, 4), "groups_per_s": G / t_tick, "bytes_per_group": 23,
                "frac_of_hbm_peak": round(G * 23 / t_tick / 1e9 / HBM_PEAK_GBS, 4), "kernel": "k_tick<uint16_t>",
                "counts_hup_checkquorum_stepdown_aborttransfer_beat": [int(c) for c in counts],
                "tick_with_counts_and_event_list_wall_ms": round(t_ev * 1e3, 3), "events_listed": int(n_ev),
                "oracle_checked_groups": n,
                "cpu_port_1_thread_groups_per_s": n / t_cpu}  # oracle_batch_tick on the checked slice: the reference's loop shape


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batches", type=int, default=12, help="distinct resident batches (shards) to rotate over")
    ap.add_argument("--shards-per-launch", type=int, default=0,
                    help="steps per rafthip_commit_many launch (0: the largest divisor of gcd(steps, warmup) <= --batches)")
    ap.add_argument("--config", choices=["headline", "cfg3", "cfg5"], default="headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (profiling runs)")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no rank environment: be the launcher.  Nothing above has touched a GPU.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d; launch %d ranks (or drop the rank environment and "
                         "let `python bench.py --gpus %d` start them)" % (world, args.gpus, args.gpus, args.gpus))

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner on stdout when a
    # communicator comes up (gloo a connection notice): from here on file descriptor 1 is stderr, and the
    # line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist

    import raft_amd
    from raft_amd import multigpu
    from raft_amd.capi import EngineSet

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    # Rehearsal on a one-GPU box (RAFTHIP_BENCH_REHEARSE=1): every rank drives cuda:0 and the
    # collectives run over gloo on host tensors — the multi-rank control flow without RCCL.
    rehearse = os.environ.get("RAFTHIP_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearse else dev  # where torch's collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    stream = torch.cuda.Stream(device=dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(*vals):
        if world == 1:
            return [float(v) for v in vals]
        t = torch.tensor(list(vals), dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    import oracle

    extra = {}
    K, W = args.steps, args.warmup

    if args.config != "headline":
        # One of the cross-GPU BASELINE configs as the workload: strong scaling, total groups fixed.
        # cfg3: K commit passes between the barriers, wall clock.  cfg5: a step = scatter + commit, timed
        # per step with HIP events inside the runner (the synthetic acks are refreshed between steps,
        # untimed); the step time is the mean over min(K, 20) steps.
        if args.config == "cfg3":
            info, e, _ = run_cfg3(raft_amd, torch, dev, local_rank, stream, rank, world, max(1, min(W, 3)))
            with torch.cuda.stream(stream):
                barrier()
                t0 = time.perf_counter()
                for _ in range(K):
                    e.commit()
                barrier()
                wall = time.perf_counter() - t0
            total = CFG3_GROUPS
        else:
            info, e, sec = run_cfg5(raft_amd, torch, dev, local_rank, stream, rank, world, max(1, min(K, 20)))
            barrier()
            wall = sec * K
            total = CFG5_GROUPS
        (wall,) = max_over_ranks(wall)
        if rank == 0:
            out = {"metric": "quorum evals/sec (groups/s), BASELINE %s" % args.config, "value": total * K / wall,
                   "unit": "groups/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * wall / K,
                   "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                   "config": {"workload": ("cfg3: 2^24 groups x 6 slots, joint mix, ProgressTracker.Committed, sharded by group range"
                                           if args.config == "cfg3" else
                                           "cfg5: 2^24 groups x (7 voters + 2 learners), 2^26 MsgAppResp acks per step routed by "
                                           "group range -> rafthip_scatter_acks (partitioned form at this size) -> commit pass; step time by HIP events"),
                              "groups_total": total, "sharding": "group-id range, no data-path collective"},
                   "extra": {"rank0": info}}
            if args.config == "cfg3":
                ach = info["groups_this_rank"] * info["bytes_per_group"] / (wall / K) / 1e9
                out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": ach / HBM_PEAK_GBS, "traffic": None, "kernel": info["kernel"]}
            else:
                # both ceilings: the HBM line at the algorithmic 24 B per ack (16 B record + 8 B cell), and the rate of
                # scattered 8-byte atomics the one-atomic-per-ack form is bound by
                ach = info["acks_per_s_this_rank"] * 24 / 1e9
                out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                   "kernel": "rafthip_scatter_acks: " + info["scatter_form"],
                                   "atomic_rate_ceiling": {"achieved": info["acks_per_s_this_rank"], "peak": ATOMIC_RATE_PEAK,
                                                           "unit": "acks/s", "frac": info["scatter_frac_of_atomic_rate"]}}
            emit(out)
        e.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- headline -----------------------------------------------------------------------------
    nb = args.batches
    engines, mats = [], []
    for b in range(nb):
        # group-id range of this rank's shard of batch b: seeds differ per (rank, batch)
        m = splitmix_batch(0x5EED0000 + 1000 * rank + b, VOTERS, G_BATCH)
        e = raft_amd.Engine(G_BATCH, VOTERS, device=local_rank, stream=stream.cuda_stream)
        e.set_uniform_config(VOTERS)
        e.upload_match(m)
        engines.append(e)
        mats.append(m)
    first_match = mats[0]

    # Every launch covers the same number L of shards (= steps), warm-up included.
    L = args.shards_per_launch
    if L <= 0:
        g = math.gcd(K, W) if W > 0 else K
        L = max(d for d in range(1, min(nb, g) + 1) if g % d == 0)
    L = max(1, min(L, nb))
    # rotation: launch j covers shards (j*L .. j*L+L-1) mod nb; one EngineSet per distinct start
    sets = {}
    for start in range(0, nb * L, L):
        s0 = start % nb
        if s0 not in sets:
            sets[s0] = EngineSet([engines[(s0 + i) % nb] for i in range(L)])

    def run(n_steps, at=0):
        """n_steps steps as launches of L shards (+ one shorter launch if L does not divide n_steps)."""
        launches, pos = 0, at
        while n_steps > 0:
            take = min(L, n_steps)
            sets[pos % nb].commit(take)
            pos += take
            n_steps -= take
            launches += 1
        return launches, pos

    with torch.cuda.stream(stream):
        # set-up, before the warm-up: one pass over every shard (the first kernel that touches a freshly
        # uploaded plane pays its page-table walks) and one use of the timing events (HIP creates them lazily)
        for e in engines:
            e.commit()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        ev1.record(stream)
        torch.cuda.synchronize()
        _, pos = run(W)
        barrier()
        t0 = time.perf_counter()
        ev0.record(stream)
        n_launches, pos = run(K, pos)
        ev1.record(stream)
        barrier()
        t1 = time.perf_counter()
    wall = t1 - t0
    gpu_ms = ev0.elapsed_time(ev1)
    wall, gpu_ms = max_over_ranks(wall, gpu_ms)

    # correctness spot check of what was timed (oracle = checker only)
    got = engines[0].read_commit()
    idx = np.arange(0, G_BATCH, 4099)
    want = oracle.batch_committed(np.ascontiguousarray(first_match[:, idx]), uniform_voters=VOTERS)
    if not np.array_equal(got[idx], want):
        raise SystemExit("bench: GPU result differs from the oracle")

    extra["shards_per_launch"] = L
    extra["launches_timed"] = n_launches
    if not args.no_extras:
        # the single-engine entry point, one 2^20-group shard per launch on one stream (round 1's headline
        # shape): what a host with ONE shard per GPU per tick gets
        with torch.cuda.stream(stream):
            for i in range(50):
                engines[i % nb].commit()
            ev2, ev3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev2.record(stream)
            for i in range(600):
                engines[i % nb].commit()
            ev3.record(stream)
            ev3.synchronize()
        one_us = 1e3 * ev2.elapsed_time(ev3) / 600
        extra["one_shard_per_launch_avg_us"] = round(one_us, 3)
        extra["one_shard_per_launch_frac_of_hbm_peak"] = round(BYTES_PER_GROUP * G_BATCH / (one_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)

        # single-launch latency (launch + kernel + sync), the floor a host sees at G = 2^20
        with torch.cuda.stream(stream):
            lat = []
            for i in range(50):
                torch.cuda.synchronize()
                a = time.perf_counter()
                engines[i % nb].commit()
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

        # the stateless entry point (what quorum.BatchCommittedIndex binds to): pageable host planes in,
        # host result out — the PCIe-inclusive rate, never `value`
        if rank == 0:
            full = np.full(G_BATCH, (1 << VOTERS) - 1, dtype=np.uint8)
            one = raft_amd.batch_committed_index(first_match, full, device=local_rank)
            rounds = []
            for _ in range(5):  # five rounds of ten calls, the median round reported: the host side of this call
                a = time.perf_counter()  # (staging threads, page-locked copies) shares the box with other tenants
                for _ in range(10):  # a tick loop keeps its result slice
                    raft_amd.batch_committed_index(first_match, full, device=local_rank, out=one)
                rounds.append(time.perf_counter() - a)
            extra["oneshot_host_buffers_groups_per_s"] = 10 * G_BATCH / sorted(rounds)[2]
            extra["oneshot_rounds_groups_per_s"] = [round(10 * G_BATCH / r) for r in rounds]
            a = time.perf_counter()
            for _ in range(5):   # a fresh result array per call: its first-touch page faults are in the figure
                one = raft_amd.batch_committed_index(first_match, full, device=local_rank)
            extra["oneshot_fresh_result_groups_per_s"] = 5 * G_BATCH / (time.perf_counter() - a)
            if not np.array_equal(one[idx], want):
                raise SystemExit("bench: one-shot result differs from the oracle")

    # the one exchange step: changed (group, commit) pairs -> every rank, rafthip_commit_allgather over
    # RCCL.  An extra, measured after the timed region: a failure here is reported in the line, it does
    # not take the headline with it — not even a collective that never returns: it runs on a helper
    # thread, and a rank that has waited EXCHANGE_TIMEOUT_S for it reports that, prints the line and
    # leaves the process without joining the thread.
    import threading
    exchange_hung = [False]

    def exchange():
        try:
            torch.cuda.set_device(local_rank)
            with torch.cuda.stream(stream):
                e0 = engines[0]
                recs = np.zeros(G_BATCH // 8, dtype=raft_amd.ACK_DTYPE)  # acks touching 1/8 of the groups
                rng = np.random.default_rng(rank)
                recs["group"] = rng.integers(0, G_BATCH, size=len(recs))
                recs["slot"] = rng.integers(0, VOTERS, size=len(recs))
                if rehearse:
                    exch = lambda: multigpu.commit_changed_allgather(e0, rank * G_BATCH, cdev)  # noqa: E731
                    extra["gloo_ranks_seen"] = world  # a rehearsal: no RCCL communicator exists
                else:
                    comm = multigpu.make_comm(e0, group_base=rank * G_BATCH)
                    # comm.nranks is RCCL's own ncclCommCount (rafthip_comm_info): did RCCL see all N ranks?
                    extra["rccl_ranks_seen"] = int(comm.nranks)
                    if comm.nranks != world:
                        raise RuntimeError("the communicator has %d ranks, the job %d" % (comm.nranks, world))
                    gathered = torch.empty(2 * comm.nranks * comm.max_pairs_per_rank, dtype=torch.int64, device=dev)

                    def exch():
                        _, counts, total = comm.commit_allgather(max_pairs=gathered.numel() // 2, dst=gathered.data_ptr())
                        return gathered[:2 * total], int(counts[rank])
                times = []
                for k in range(4):  # the first exchange carries first-use costs and sizes its payload with a host wait
                    recs["index"] = (np.uint64(1) << np.uint64(62)) + np.uint64(k)
                    e0.scatter_acks(recs)
                    barrier()
                    a = time.perf_counter()
                    allp, n_changed = exch()
                    torch.cuda.synchronize()
                    times.append(time.perf_counter() - a)
                if not rehearse:
                    allp = allp.cpu().numpy().view(raft_amd.PAIR_DTYPE)
            (t_ex,) = max_over_ranks(min(times[1:]))
            extra["changed_pairs_rank0"] = int(n_changed)
            extra["commit_changed_allgather_ms"] = round(1e3 * t_ex, 3)
            extra["commit_changed_allgather_first_ms"] = round(1e3 * times[0], 3)
            extra["allgather_pairs_total"] = int(len(allp))
            if not (np.diff(allp["group"].astype(np.int64)) > 0).all():
                extra["commit_changed_allgather_error"] = "gathered pairs are not sorted by global group id"
            if not rehearse:
                comm.close()
        except Exception as ex:
            extra["commit_changed_allgather_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:300])

    exchange_hung[0] = guarded_exchange(exchange, extra, EXCHANGE_TIMEOUT_S)

    if not args.no_extras and world == 1 and not exchange_hung[0]:
        # HIP-graph-amortised launch (SURVEY.md §8d caveat): one rotation over all batches, one shard per
        # kernel, captured on the engine stream and replayed — the per-launch host cost is gone, what
        # remains is the kernel plus the kernel-to-kernel boundary on the device
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for e in engines:
                    e.commit()
            with torch.cuda.stream(stream):
                for _ in range(5):
                    graph.replay()
                ev6, ev7 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = max(1, 1200 // nb)
                ev6.record(stream)
                for _ in range(reps):
                    graph.replay()
                ev7.record(stream)
                ev7.synchronize()
            extra["hip_graph_avg_launch_us"] = round(1e3 * ev6.elapsed_time(ev7) / (reps * nb), 3)
            del graph
        except Exception as ex:  # capture is a measurement extra, never the headline
            extra["hip_graph_avg_launch_us"] = None
            extra["hip_graph_error"] = str(ex)[:200]

        # the same rotation with every engine on its OWN stream and the single-engine entry point: passes
        # over different engines are independent, so one pass's drain overlaps the next one's ramp-up
        own = []
        for m in mats:
            e = raft_amd.Engine(G_BATCH, VOTERS, device=local_rank)
            e.set_uniform_config(VOTERS)
            e.upload_match(m)
            own.append(e)
        for i in range(200):
            own[i % nb].commit()
        torch.cuda.synchronize()
        a = time.perf_counter()
        for i in range(2000):
            own[i % nb].commit()
        torch.cuda.synchronize()
        dt = time.perf_counter() - a
        extra["own_stream_engines_groups_per_s"] = 2000 * G_BATCH / dt
        extra["own_stream_engines_GBs"] = round(2000 * G_BATCH * BYTES_PER_GROUP / dt / 1e9, 1)
        if not np.array_equal(own[0].read_commit()[idx], want):
            raise SystemExit("bench: own-stream result differs from the oracle")
        # a multi-shard host's commit step with the changed pairs reported (here: nothing changes, the steady
        # state between acks): shard by shard — a launch pair and a wait each — against rafthip_commit_report_many
        try:
            own_set = EngineSet(own)
            for e in own:
                e.commit_changed(max_pairs=0)
            own_set.commit_report(max_pairs=0)
            torch.cuda.synchronize()
            a = time.perf_counter()
            for _ in range(20):
                for e in own:
                    e.commit_changed(max_pairs=0)
            t_each = (time.perf_counter() - a) / 20
            a = time.perf_counter()
            for _ in range(20):
                own_set.commit_report(max_pairs=0)
            t_many = (time.perf_counter() - a) / 20
            extra["commit_changed_%d_shards_one_by_one_us" % nb] = round(1e6 * t_each, 1)
            extra["commit_changed_%d_shards_report_many_us" % nb] = round(1e6 * t_many, 1)
        except Exception as ex:
            extra["commit_report_many_error"] = str(ex)[:200]
        for e in own:
            e.close()
    del mats
    if not exchange_hung[0]:
        for e in engines:
            e.close()
        engines = []

    if not args.no_extras and world == 1 and not exchange_hung[0]:
        # the two BASELINE configs that are defined across GPUs, whole on this one GPU
        for name, runner in (("cfg3_joint_16M", run_cfg3), ("cfg5_scatter_commit_16M", run_cfg5)):
            try:
                info, e, _ = runner(raft_amd, torch, dev, local_rank, stream, 0, 1, 3)
                e.close()
                extra[name] = info
            except Exception as ex:
                extra[name] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
            torch.cuda.empty_cache()
        # ... and the two single-GPU ones that are not the headline
        for name, runner in (("cfg2_1M_x3", run_cfg2), ("cfg4_64M_tallies", run_cfg4)):
            try:
                extra[name] = runner(raft_amd, torch, dev, local_rank, stream, 5)
            except Exception as ex:
                extra[name] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
            torch.cuda.empty_cache()
        # the rows SURVEY.md §8 marks "next" (f2, f3, f4), as one leader-side tick from serialized bytes in to bytes out
        try:
            extra["tick_4M_groups_16M_responses"] = run_tick(raft_amd, torch, dev, local_rank, stream, 3)
        except Exception as ex:
            extra["tick_4M_groups_16M_responses"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
        torch.cuda.empty_cache()
        # the caller of the CheckQuorum tally and the heartbeat fan-out: RawNode.Tick for every group
        try:
            extra["raw_node_tick_16M_groups"] = run_raw_tick(raft_amd, torch, dev, local_rank, stream, 10)
        except Exception as ex:
            extra["raw_node_tick_16M_groups"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
        torch.cuda.empty_cache()

    if rank == 0:
        total_groups = world * K * G_BATCH
        launch_s = gpu_ms * 1e-3 / n_launches          # average launch duration over the timed region
        bytes_per_launch = BYTES_PER_GROUP * G_BATCH * K / n_launches
        achieved = bytes_per_launch / launch_s / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_commit_uniform5.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                per_group = rec.get("hbm_bytes_per_group")
                traffic = per_group * G_BATCH * K / n_launches if per_group else None
            except Exception:
                traffic = None
        out = {
            "metric": "quorum evals/sec (groups/s) at 1M x 5 voters",
            "value": total_groups / wall,
            "unit": "groups/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * wall / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "ProgressTracker.Committed, 2^20 groups x 5 voters per GPU per step, non-joint, dist A "
                            "(bench_test.go:30-33), rotating over %d HBM-resident batches (shards); %d steps per "
                            "rafthip_commit_many launch" % (nb, L),
                "groups_per_step_per_gpu": G_BATCH, "voters": VOTERS, "batches": nb, "steps_per_launch": L,
                "sharding": "group-id range, no data-path collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/pmc_commit_uniform5.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per group)",
                "frac_of_dtod_copy": (achieved / extra["dtod_copy_GBs"]) if extra.get("dtod_copy_GBs") else None,
                "kernel": "k_commit_uniform_many<5>", "bytes_per_group": BYTES_PER_GROUP,
                "bytes_per_launch": bytes_per_launch, "avg_launch_us": launch_s * 1e6, "launches": n_launches,
            },
            "extra": extra,
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"], _ = cpu_baseline(first_match)
            out["extra"].update(CPU_EXTRA)
        emit(out)

    code = exit_code(extra, world, exchange_hung[0])
    if exchange_hung[0]:
        os._exit(code)  # a collective is stuck on the helper thread: no teardown can complete; the line is out
    if world > 1:
        dist.barrier()  # the job's own process group is healthy (the failure, if any, was inside the exchange)
        dist.destroy_process_group()
    if code:
        sys.exit(code)


if __name__ == "__main__":
    main()
