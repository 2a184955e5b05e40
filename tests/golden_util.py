"""Helpers shared by the oracle tests and the GPU parity tests.

`lookuper` restates the reference harness's makeLookuper
(quorum/datadriven_test.go:122-150): inputs are consumed in the order the ids
appear in (cfg, cfgj) without repetition, and "_" placeholders leave no entry.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
U64MAX = (1 << 64) - 1


def load_cases():
    with open(os.path.join(HERE, "golden", "quorum_datadriven.json")) as fh:
        return json.load(fh)["cases"]


def load_tables():
    with open(os.path.join(HERE, "golden", "tables.json")) as fh:
        return json.load(fh)


def ordered_ids(case):
    seen, out = set(), []
    for i in list(case["cfg"]) + list(case["cfgj"]):
        if i not in seen:
            seen.add(i)
            out.append(i)
    return out


def lookuper(case):
    """id -> value map (idx for committed; True/False for vote), '_' omitted."""
    vals = case["idx"] if case["cmd"] == "committed" else case["votes"]
    ids = ordered_ids(case)
    assert len(ids) == len(vals), case["src"]
    m = {}
    for i, v in zip(ids, vals):
        if v is None:
            continue
        m[i] = v if case["cmd"] == "committed" else (v == "y")
    return m


def expected_value(case):
    if case["cmd"] == "committed":
        return U64MAX if case["expected"] == "inf" else case["expected"]
    return {"VotePending": 1, "VoteLost": 2, "VoteWon": 3}[case["expected"]]


def mask_dtype(nslots):
    return {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[mask_bytes(nslots)]


def mask_bytes(nslots):
    return 1 if nslots <= 8 else 2 if nslots <= 16 else 4 if nslots <= 32 else 8


def cases_to_planes(cases, cmd, nslots=None, swap=False):
    """Pack golden cases of one command into the engine's SoA shape.

    Slot s of a group = the s-th id of sorted(cfg ∪ cfgj); unknown idx -> 0
    (majority.go:149-161), missing vote -> cell 0, n -> 1, y -> 2.
    Returns (plane [S][G], in_mask, out_mask, expected).
    """
    cs = [c for c in cases if c["cmd"] == cmd]
    S = nslots or max(1, max(len(ordered_ids(c)) for c in cs))
    G = len(cs)
    mdt = mask_dtype(S)
    plane = np.zeros((S, G), dtype=np.uint64 if cmd == "committed" else np.uint8)
    m0 = np.zeros(G, dtype=mdt)
    m1 = np.zeros(G, dtype=mdt)
    exp = np.zeros(G, dtype=np.uint64 if cmd == "committed" else np.uint8)
    for g, c in enumerate(cs):
        ids = sorted(ordered_ids(c))
        slot = {i: s for s, i in enumerate(ids)}
        l = lookuper(c)
        for i, v in l.items():
            plane[slot[i], g] = v if cmd == "committed" else (2 if v else 1)
        a = sum(1 << slot[i] for i in c["cfg"])
        b = sum(1 << slot[i] for i in c["cfgj"])
        if swap:
            a, b = b, a
        m0[g], m1[g] = a, b
        exp[g] = expected_value(c)
    return plane, m0, m1, exp


class SplitMix64:
    """splitmix64 — the bench/test PRNG named in SURVEY.md §8(d)."""

    def __init__(self, seed):
        self.s = seed & U64MAX

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & U64MAX
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & U64MAX
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & U64MAX
        return z ^ (z >> 31)


def splitmix64_array(seed, n):
    """Vectorised splitmix64 stream: element i = output i+1 of SplitMix64(seed)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def dist_a(seed, S, G):
    """bench-dist: uniform in [0, 2^63-1) — rand.Int63n(MaxInt64), bench_test.go:32."""
    r = splitmix64_array(seed, S * G).reshape(S, G)
    return np.ascontiguousarray((r >> np.uint64(1)) % np.uint64((1 << 63) - 1))


def dist_b(seed, S, G):
    """raft-dist: leader index L in [1,2^40], followers lag
    geometrically, 5% zero cells, 1% all-equal groups, 0.1% cells >= 2^63."""
    r = splitmix64_array(seed, (S + 3) * G).reshape(S + 3, G)
    L = (r[S] % np.uint64(1 << 40)) + np.uint64(1)
    m = np.empty((S, G), dtype=np.uint64)
    for s in range(S):
        x = r[s]
        # geometric(p=1/2) lag = number of trailing one bits of a random word, capped at 20
        low0 = ~x & (x + np.uint64(1))  # lowest zero bit (0 when x is all ones)
        lag = np.minimum(np.bitwise_count(low0 - np.uint64(1)).astype(np.uint64), np.uint64(20))
        lag = np.minimum(lag, L)
        v = L - lag if s else L
        zero = ((x >> np.uint64(32)) % np.uint64(100)) < np.uint64(5)
        big = ((x >> np.uint64(40)) % np.uint64(1000)) == np.uint64(0)
        if s:
            v = np.where(zero, np.uint64(0), v)
        v = np.where(big, v | np.uint64(1 << 63), v)
        m[s] = v
    alleq = (r[S + 1] % np.uint64(100)) == np.uint64(0)
    m[:, alleq] = L[alleq]
    return np.ascontiguousarray(m)


def term_range(log_off, log_terms, log_first, log_snap_term, term):
    """What a host hands to rafthip_upload_term_range: per group the first and
    last log index whose entry carries `term` (log terms never decrease, so they
    form one run; the compacted boundary entry first-1 keeps its term,
    log.go:381-407); an empty range (1, 0) when there is none or term == 0."""
    G = len(log_first)
    tf = np.ones(G, dtype=np.uint64)
    tl = np.zeros(G, dtype=np.uint64)
    for g in range(G):
        if int(term[g]) == 0:
            continue
        t = np.concatenate(([log_snap_term[g]], log_terms[int(log_off[g]):int(log_off[g + 1])]))
        hit = np.flatnonzero(t == term[g])
        if len(hit):
            tf[g] = int(log_first[g]) - 1 + int(hit[0])
            tl[g] = int(log_first[g]) - 1 + int(hit[-1])
    return tf, tl


def random_logs(rng, G, max_len=12):
    """Per-group logs with non-decreasing terms, CSR-packed: (off, terms, first, snap_term)."""
    lens = rng.integers(0, max_len + 1, size=G)
    off = np.zeros(G + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    first = rng.integers(1, 6, size=G).astype(np.uint64)
    snap = rng.integers(0, 3, size=G).astype(np.uint64)
    terms = np.zeros(int(off[-1]), dtype=np.uint64)
    for g in range(G):
        n = int(lens[g])
        if n:
            steps = (rng.random(n) < 0.3).astype(np.uint64)
            terms[int(off[g]):int(off[g + 1])] = snap[g] + np.uint64(rng.integers(0, 2)) + np.cumsum(steps).astype(np.uint64)
    return off, terms, first, snap
