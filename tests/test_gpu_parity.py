"""GPU parity: the HIP engine, called through the C ABI, against the CPU oracle
and the reference's golden vectors.  Bit-exact: the path is uint64/uint8 only."""
import numpy as np
import pytest

import golden_util as gu
import raft_amd

pytestmark = pytest.mark.gpu

CASES = gu.load_cases()
TABLES = gu.load_tables()


def rand_masks(rng, S, G, joint_frac=0.5, empty_frac=0.02):
    mdt = gu.mask_dtype(S)
    full = (1 << S) - 1

    def rnd():
        if S < 64:
            return rng.integers(0, full + 1, size=G, dtype=np.uint64)
        return rng.integers(0, 1 << 63, size=G, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=G, dtype=np.uint64)

    m0 = rnd()
    m1 = np.where(rng.random(G) < joint_frac, rnd(), np.uint64(0))
    m0 = np.where(rng.random(G) < empty_frac, np.uint64(0), m0)
    return m0.astype(mdt), m1.astype(mdt)


def rand_match(rng, S, G):
    """Mixture: full-range u64 (incl. >= 2^63), small values with many ties, zeros."""
    a = rng.integers(0, 1 << 63, size=(S, G), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(S, G), dtype=np.uint64)
    b = rng.integers(0, 8, size=(S, G), dtype=np.uint64)
    sel = rng.random((S, G))
    m = np.where(sel < 0.4, a, b)
    m = np.where(sel > 0.9, np.uint64(0), m)
    return np.ascontiguousarray(m.astype(np.uint64))


# ---------------------------------------------------------------------------
# golden vectors through the engine
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("nslots", [None, 8, 12, 16, 24, 32, 40, 64])
@pytest.mark.parametrize("swap", [False, True])
def test_golden_committed(oracle, nslots, swap):
    plane, m0, m1, exp = gu.cases_to_planes(CASES, "committed", nslots=nslots, swap=swap)
    S, G = plane.shape
    with raft_amd.Engine(G, S) as e:
        e.upload_config(m0, m1)
        e.upload_match(plane)
        e.commit()
        got = e.read_commit()
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("nslots", [None, 8, 12, 32, 64])
@pytest.mark.parametrize("swap", [False, True])
def test_golden_votes(oracle, nslots, swap):
    votes, m0, m1, exp = gu.cases_to_planes(CASES, "vote", nslots=nslots, swap=swap)
    S, G = votes.shape
    with raft_amd.Engine(G, S) as e:
        e.upload_config(m0, m1)
        e.upload_votes(votes)
        e.tally_votes()
        res, gr, rj = e.read_vote_result()
    assert np.array_equal(res, exp)
    ores, ogr, orj = oracle.batch_tally_votes(votes, m0, m1)
    assert np.array_equal(gr, ogr) and np.array_equal(rj, orj)


def test_golden_oneshot_entry_points():
    plane, m0, m1, exp = gu.cases_to_planes(CASES, "committed")
    assert np.array_equal(raft_amd.batch_committed_index(plane, m0, m1), exp)
    votes, v0, v1, vexp = gu.cases_to_planes(CASES, "vote")
    assert np.array_equal(raft_amd.batch_vote_result(votes, v0, v1), vexp)
    # non-joint form: out_mask omitted == zero MajorityConfig (datadriven_test.go:179)
    nj = [c for c in CASES if not c["joint"]]
    plane, m0, m1, exp = gu.cases_to_planes(nj, "committed")
    assert np.array_equal(raft_amd.batch_committed_index(plane, m0, None), exp)


def test_testcommit_and_paper_tables(oracle):
    for row in TABLES["TestCommit"]["rows"]:
        n = len(row["matches"])
        plane = np.array(row["matches"], dtype=np.uint64).reshape(n, 1)
        with raft_amd.Engine(1, n) as e:
            e.set_uniform_config(n)
            e.upload_match(plane)
            e.commit()
            mci = int(e.read_commit()[0])
        terms = {i: t for i, t in row["logs"]}
        assert oracle.log_maybe_commit(0, mci, terms.get(mci, 0), row["smTerm"]) == row["w"], row
    li = 1
    for row in TABLES["TestLeaderAcknowledgeCommit"]["rows"]:
        n = row["size"]
        with raft_amd.Engine(1, n) as e:
            e.set_uniform_config(n)
            recs = np.zeros(1 + len(row["acceptors"]), dtype=raft_amd.ACK_DTYPE)
            e.upload_match(np.full((n, 1), li, dtype=np.uint64))
            recs["slot"] = [0] + [a - 1 for a in row["acceptors"]]
            recs["index"] = li + 1
            e.scatter_acks(recs)
            e.commit()
            assert (int(e.read_commit()[0]) > li) == row["wack"], row


def test_progress_update_table():
    t = TABLES["TestProgressUpdate"]
    for row in t["rows"]:
        with raft_amd.Engine(1, 1) as e:
            e.upload_match(np.array([[t["prevM"]]], dtype=np.uint64))
            recs = np.zeros(1, dtype=raft_amd.ACK_DTYPE)
            recs["index"] = row["update"]
            e.scatter_acks(recs)
            assert int(e.read_match()[0, 0]) == row["wm"]


# ---------------------------------------------------------------------------
# randomised differential tests against the oracle
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("V", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("G", [1, 511, 4097])
def test_commit_uniform(oracle, V, G):
    rng = np.random.default_rng(100 + V)
    S = min(8, V + 1)  # one learner row when it fits
    match = rand_match(rng, S, G)
    with raft_amd.Engine(G, S) as e:
        e.set_uniform_config(V)
        e.upload_match(match)
        e.commit()
        got = e.read_commit()
    assert np.array_equal(got, oracle.batch_committed(match, uniform_voters=V))


@pytest.mark.parametrize("S", list(range(1, 9)))
def test_commit_masked_small(oracle, S):
    rng = np.random.default_rng(200 + S)
    G = 20011
    match = rand_match(rng, S, G)
    m0, m1 = rand_masks(rng, S, G)
    with raft_amd.Engine(G, S) as e:
        e.upload_config(m0, m1)
        e.upload_match(match)
        e.commit()
        got = e.read_commit()
    assert np.array_equal(got, oracle.batch_committed(match, m0, m1))


@pytest.mark.parametrize("flags", [0, raft_amd.capi.FLAG_WIDE_BITONIC], ids=["regs", "bitonic"])
@pytest.mark.parametrize("S", [9, 11, 12, 13, 16, 17, 24, 25, 32, 33, 48, 49, 63, 64])
def test_commit_wide(oracle, S, flags):
    """Both wide kernels: the register Batcher network (default) and the
    cross-lane wave64 bitonic network (RAFTHIP_FLAG_WIDE_BITONIC)."""
    rng = np.random.default_rng(300 + S)
    G = 3001
    match = rand_match(rng, S, G)
    m0, m1 = rand_masks(rng, S, G)
    with raft_amd.Engine(G, S, flags=flags) as e:
        e.upload_config(m0, m1)
        e.upload_match(match)
        e.commit()
        got = e.read_commit()
        assert np.array_equal(got, oracle.batch_committed(match, m0, m1))
        # uniform config on the wide path
        V = S - 1
        e.set_uniform_config(V)
        e.commit()
        assert np.array_equal(e.read_commit(), oracle.batch_committed(match, uniform_voters=V))


@pytest.mark.parametrize("flags", [0, raft_amd.capi.FLAG_WIDE_BITONIC], ids=["regs", "bitonic"])
def test_commit_wide_non_joint_wave(oracle, flags):
    """All-non-joint wavefronts take the skip-outgoing branch."""
    rng = np.random.default_rng(5)
    S, G = 20, 1000
    match = rand_match(rng, S, G)
    m0, _ = rand_masks(rng, S, G)
    m1 = np.zeros_like(m0)
    with raft_amd.Engine(G, S, flags=flags) as e:
        e.upload_config(m0, m1)
        e.upload_match(match)
        e.commit()
        assert np.array_equal(e.read_commit(), oracle.batch_committed(match, m0, m1))


def test_empty_config_commits_infinity():
    with raft_amd.Engine(100, 3) as e:  # fresh engine: both halves empty
        e.commit()
        assert (e.read_commit() == np.uint64(gu.U64MAX)).all()
    with raft_amd.Engine(100, 20) as e:
        e.commit()
        assert (e.read_commit() == np.uint64(gu.U64MAX)).all()


@pytest.mark.parametrize("S", [3, 5, 8, 9, 10, 12, 16, 17, 33, 40, 64])
def test_tallies_random(oracle, S):
    rng = np.random.default_rng(400 + S)
    G = 50021 if S <= 8 else 5003
    mdt = gu.mask_dtype(S)
    m0, m1 = rand_masks(rng, S, G)
    votes = rng.choice(np.array([0, 1, 2], dtype=np.uint8), size=(S, G), p=[0.4, 0.2, 0.4])
    votes = np.ascontiguousarray(votes)
    learners = (~(m0 | m1)) & mdt((1 << S) - 1 if S < 64 else gu.U64MAX)
    bits_a, _ = rand_masks(rng, S, G, joint_frac=0, empty_frac=0)
    bits_k, _ = rand_masks(rng, S, G, joint_frac=0, empty_frac=0.3)
    with raft_amd.Engine(G, S) as e:
        e.upload_config(m0, m1, learners)
        e.upload_votes(votes)
        e.tally_votes()
        res, gr, rj = e.read_vote_result()
        ores, ogr, orj = oracle.batch_tally_votes(votes, m0, m1, learners)
        assert np.array_equal(res, ores)
        assert np.array_equal(gr, ogr) and np.array_equal(rj, orj)
        e.upload_active(bits_a)
        e.quorum_active()
        assert np.array_equal(e.read_quorum_active(), oracle.batch_quorum_active(bits_a, S, m0, m1, learners))
        e.upload_acks(bits_k)
        e.ack_quorum()
        assert np.array_equal(e.read_ack_quorum(), oracle.batch_ack_quorum(bits_k, S, m0, m1))
        # uniform config
        V = max(1, S - 2)
        e.set_uniform_config(V)
        e.tally_votes()
        res, gr, rj = e.read_vote_result()
        ores, ogr, orj = oracle.batch_tally_votes(votes, uniform_voters=V)
        assert np.array_equal(res, ores) and np.array_equal(gr, ogr) and np.array_equal(rj, orj)
        e.quorum_active()
        assert np.array_equal(e.read_quorum_active(), oracle.batch_quorum_active(bits_a, S, uniform_voters=V))
        e.ack_quorum()
        assert np.array_equal(e.read_ack_quorum(), oracle.batch_ack_quorum(bits_k, S, uniform_voters=V))


def test_check_quorum_three_voters(oracle):
    """TestLeaderStepdownWhenQuorumActive/Lost, raft_test.go:1911-1944."""
    with raft_amd.Engine(4, 3) as e:
        e.set_uniform_config(3)
        e.upload_active(np.array([0b001, 0b011, 0b101, 0b111], dtype=np.uint8))
        e.quorum_active()
        assert e.read_quorum_active().tolist() == [0, 1, 1, 1]


@pytest.mark.parametrize("S,G", [(9, 4001), (3, 70001), (20, 999)])
def test_scatter_then_commit(oracle, S, G):
    """cfg5 shape: ack stream (stale + duplicate cells) -> atomicMax -> commit pass."""
    rng = np.random.default_rng(500 + S)
    match = rng.integers(0, 1 << 40, size=(S, G), dtype=np.uint64)
    N = 4 * G
    recs = np.zeros(N, dtype=raft_amd.ACK_DTYPE)
    recs["group"] = rng.integers(0, G, size=N)
    recs["slot"] = rng.integers(0, S, size=N)
    old = match[recs["slot"], recs["group"]]
    recs["index"] = old + rng.integers(0, 64, size=N).astype(np.uint64)
    stale = rng.random(N) < 0.05
    recs["index"][stale] = old[stale] // np.uint64(2)
    # a few malformed records must be dropped and counted, not faulted on
    recs["group"][:3] = [G, G + 5, 0xFFFFFFFF]
    recs["slot"][3:5] = [S, 255]
    want = match.copy()
    skipped = oracle.scatter_acks(want, recs)
    V = min(S, 7)
    with raft_amd.Engine(G, S) as e:
        e.set_uniform_config(V)
        e.upload_match(match)
        e.scatter_acks(recs)
        e.commit()
        assert np.array_equal(e.read_match(), want)
        assert np.array_equal(e.read_commit(), oracle.batch_committed(want, uniform_voters=V))
        st = e.stats()
        assert st["acks_rejected"] == skipped == 5
        assert st["acks_applied"] == N


@pytest.mark.parametrize("S,G", [(9, 4001), (3, 70002), (20, 999), (1, 1), (64, 130)])
def test_merge_acks_equals_the_ack_stream(oracle, S, G):
    """The tick's acks as a dense [slot][group] plane (0 = none): the same Match as stepping the
    equivalent ack records one by one (Progress.MaybeUpdate, tracker/progress.go:168-177); host and
    device planes, sub-ranges at odd offsets, values >= 2^63."""
    import torch
    rng = np.random.default_rng(550 + S)
    match = rng.integers(0, 1 << 40, size=(S, G), dtype=np.uint64)
    match[rng.random((S, G)) < 0.02] |= np.uint64(1) << np.uint64(63)
    plane = np.where(rng.random((S, G)) < 0.45, match + rng.integers(0, 64, size=(S, G)).astype(np.uint64)
                     - np.minimum(match, 16), 0).astype(np.uint64)
    plane[rng.random((S, G)) < 0.01] = np.uint64((1 << 64) - 1)
    plane = np.ascontiguousarray(plane)
    ss, gg = np.nonzero(plane)
    recs = np.zeros(len(ss), dtype=raft_amd.ACK_DTYPE)
    recs["slot"], recs["group"], recs["index"] = ss, gg, plane[ss, gg]
    want = match.copy()
    assert oracle.scatter_acks(want, recs) == 0
    assert np.array_equal(want, np.maximum(match, plane))
    with raft_amd.Engine(G, S) as e:
        e.upload_match(match)
        e.merge_acks(plane)
        assert np.array_equal(e.read_match(), want)
        e.merge_acks(plane)  # idempotent
        assert np.array_equal(e.read_match(), want)
        # device plane with a pitch, and an odd sub-range from the host
        e.upload_match(match)
        pitch = G + 6
        d = torch.zeros((S, pitch), dtype=torch.int64, device="cuda")
        d[:, :G] = torch.from_numpy(plane.view(np.int64)).cuda()
        e.merge_acks(d, stride=pitch)
        e.synchronize()
        assert np.array_equal(e.read_match(), want)
        if G > 8:
            e.upload_match(match)
            lo, cnt = 3, G - 7
            e.merge_acks(np.ascontiguousarray(plane[:, lo:lo + cnt]), first=lo, count=cnt)
            w2 = match.copy()
            w2[:, lo:lo + cnt] = want[:, lo:lo + cnt]
            assert np.array_equal(e.read_match(), w2)
        with pytest.raises(raft_amd.RafthipError):
            e.merge_acks(plane, first=1, count=G)


@pytest.mark.parametrize("S,uniform,flags", [(5, True, 0), (6, False, 0), (12, False, 0), (12, False, 1), (40, False, 0),
                                              (12, True, 0)])
def test_commit_changed_pairs(oracle, S, uniform, flags):
    rng = np.random.default_rng(600 + S)
    G = 30011
    match = rng.integers(1, 1 << 30, size=(S, G), dtype=np.uint64)
    m0, m1 = rand_masks(rng, S, G, empty_frac=0)
    with raft_amd.Engine(G, S, flags=flags) as e:
        if uniform:
            e.set_uniform_config(S)
            kw = dict(uniform_voters=S)
        else:
            e.upload_config(m0, m1)
            kw = dict(in_mask=m0, out_mask=m1)
        e.upload_match(match)
        if S == 12 and not uniform and not flags:
            # an engine may move between the two tilings: uniform <= 8 voters runs K1 even at S = 12
            e.set_uniform_config(7)
            _, n7 = e.commit_changed(max_pairs=0)
            assert n7 == G and np.array_equal(e.read_commit(), oracle.batch_committed(match, uniform_voters=7))
            e.upload_commit(np.zeros(G, dtype=np.uint64))
            e.upload_config(m0, m1)
        want0 = oracle.batch_committed(match, **kw)
        pairs, n = e.commit_changed()
        nz = np.flatnonzero(want0 != 0)
        assert n == len(nz)
        assert np.array_equal(pairs["group"], nz.astype(np.uint64))  # sorted by group
        assert np.array_equal(pairs["commit"], want0[nz])
        # nothing changes on a second pass
        pairs, n = e.commit_changed()
        assert n == 0
        # advance ~1% of the cells; only affected groups are reported
        match2 = match.copy()
        bump = rng.random((S, G)) < 0.01
        match2[bump] += np.uint64(1000)
        e.upload_match(match2)
        want1 = oracle.batch_committed(match2, **kw)
        pairs, n = e.commit_changed(max_pairs=10)
        ch = np.flatnonzero(want1 != want0)
        assert n == len(ch)
        assert np.array_equal(pairs["group"], ch[:10].astype(np.uint64))
        assert np.array_equal(e.read_commit(), want1)


def test_commit_changed_pass_after_pass_across_tilings(oracle):
    """The engine alternates between two arrays of per-64-tile sums, each reporting pass zeroing the other one for
    its successor (no memset in front of a pass).  Passes in a row, with the engine moving between the 256-group
    tiles of the wide kernel and the 2048-group tiles of K1 (different numbers of sums in use), must each report
    exactly the groups that changed."""
    rng = np.random.default_rng(4242)
    S, G = 20, 600_001
    match = rng.integers(1, 1 << 40, size=(S, G), dtype=np.uint64)
    m0, m1 = rand_masks(rng, S, G, empty_frac=0)
    with raft_amd.Engine(G, S) as e:
        e.upload_match(match)
        have = np.zeros(G, dtype=np.uint64)
        for step, wide in enumerate([True, True, False, True, False, False, True]):
            if wide:
                e.upload_config(m0, m1)
                kw = dict(in_mask=m0, out_mask=m1)
            else:
                e.set_uniform_config(5)
                kw = dict(uniform_voters=5)
            bump = rng.random((S, G)) < (0.2 if step % 2 else 0.002)
            match[bump] += np.uint64(1 << 20)
            e.upload_match(match)
            want = oracle.batch_committed(match, **kw)
            pairs, n = e.commit_changed()
            ch = np.flatnonzero(want != have)
            assert n == len(ch), step
            assert np.array_equal(pairs["group"], ch.astype(np.uint64)), step
            assert np.array_equal(pairs["commit"], want[ch]), step
            have = want


def test_partial_uploads_and_ranges(oracle):
    rng = np.random.default_rng(9)
    S, G = 5, 10000
    match = rand_match(rng, S, G)
    with raft_amd.Engine(G, S) as e:
        e.set_uniform_config(3)
        e.upload_match(match)
        # conf change on groups [100, 200): materialises the uniform config for the rest
        m0 = np.full(100, 0b11111, dtype=np.uint8)
        m1 = np.full(100, 0b00111, dtype=np.uint8)
        e.upload_config(m0, m1, first=100, count=100)
        fm0 = np.full(G, 0b00111, dtype=np.uint8)
        fm1 = np.zeros(G, dtype=np.uint8)
        fm0[100:200], fm1[100:200] = m0, m1
        e.commit()
        assert np.array_equal(e.read_commit(), oracle.batch_committed(match, fm0, fm1))
        # role change on groups [5000, 5100): Match resets to 0 (raft.go:774-784)
        e.reset_match(5000, 100)
        match[:, 5000:5100] = 0
        e.commit()
        assert np.array_equal(e.read_commit(5000, 100), oracle.batch_committed(match, fm0, fm1)[5000:5100])
        with pytest.raises(raft_amd.RafthipError) as ei:
            e.read_commit(G - 1, 2)
        assert ei.value.code == -34


def test_full_size_properties(oracle):
    """BASELINE sizes (2^20 x 5, dist A and B): properties that need no full oracle pass —
    joint-with-self and joint-with-empty equal the majority result, halves swap
    symmetrically (datadriven_test.go:179-183,219), raising Match never lowers
    the commit index, and a strided sample matches the oracle exactly."""
    G, S = 1 << 20, 5
    for name, match in (("A", gu.dist_a(0x5EED0002, S, G)), ("B", gu.dist_b(0x5EED0002, S, G))):
        with raft_amd.Engine(G, S) as e:
            e.set_uniform_config(S)
            e.upload_match(match)
            e.commit()
            base = e.read_commit()
            full = np.full(G, 0b11111, dtype=np.uint8)
            zero = np.zeros(G, dtype=np.uint8)
            for a, b in ((full, zero), (full, full), (zero, full)):
                e.upload_config(a, b)
                e.commit()
                assert np.array_equal(e.read_commit(), base), name
            # exact check on a sample
            idx = np.arange(0, G, 257)
            sample = np.ascontiguousarray(match[:, idx])
            assert np.array_equal(base[idx], oracle.batch_committed(sample, uniform_voters=S))
            # median-of-5 bounds: 3 values >= result and 3 values <= result
            assert ((match >= base).sum(axis=0) >= 3).all() and ((match <= base).sum(axis=0) >= 3).all()
            # monotone: acks only raise Match, so the commit index cannot drop
            rng = np.random.default_rng(3)
            N = 1 << 20
            recs = np.zeros(N, dtype=raft_amd.ACK_DTYPE)
            recs["group"] = rng.integers(0, G, size=N)
            recs["slot"] = rng.integers(0, S, size=N)
            recs["index"] = rng.integers(0, 1 << 62, size=N)
            e.set_uniform_config(S)
            e.scatter_acks(recs)
            e.commit()
            assert (e.read_commit() >= base).all(), name


def test_device_pointers_and_zero_copy_planes(oracle):
    """`src`/`dst` may be device memory, and a host that fills planes on the GPU
    itself can write them in place (rafthip_plane)."""
    import torch
    rng = np.random.default_rng(77)
    S, G = 5, 70001
    match = rand_match(rng, S, G)
    m0, m1 = rand_masks(rng, S, G)
    want = oracle.batch_committed(match, m0, m1)
    # stateless entry point, every buffer on the device
    dm = torch.from_numpy(match.view(np.int64)).cuda()
    d0, d1 = torch.from_numpy(m0).cuda(), torch.from_numpy(m1).cuda()
    dout = torch.empty(G, dtype=torch.int64, device="cuda")
    L = raft_amd.load_library()
    rc = L.rafthip_batch_committed_index(0, G, S, dm.data_ptr(), d0.data_ptr(), d1.data_ptr(), dout.data_ptr())
    assert rc == 0, L.rafthip_last_error()
    assert np.array_equal(dout.cpu().numpy().view(np.uint64), want)
    # zero-copy: write the engine's own Match plane from a torch kernel, read commit in place
    with raft_amd.Engine(G, S) as e:
        e.upload_config(m0, m1)
        ptr, stride = e.plane(raft_amd.capi.PLANE_MATCH)
        assert stride >= G and stride % 4096 == 0
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        for s in range(S):
            row = dm[s].contiguous()
            assert hip.hipMemcpy(ctypes.c_void_p(ptr + 8 * s * stride), ctypes.c_void_p(row.data_ptr()),
                                 ctypes.c_size_t(8 * G), 3) == 0  # hipMemcpyDeviceToDevice
        e.commit()
        e.synchronize()
        cptr, _ = e.plane(raft_amd.capi.PLANE_COMMIT)
        assert hip.hipMemcpy(ctypes.c_void_p(dout.data_ptr()), ctypes.c_void_p(cptr), ctypes.c_size_t(8 * G), 3) == 0
        assert np.array_equal(dout.cpu().numpy().view(np.uint64), want)


def test_records_in_page_locked_host_memory(oracle):
    """rafthip_host_alloc: records built in page-locked memory are read by the kernels straight over
    the link (no staging copy); same results as pageable host memory."""
    rng = np.random.default_rng(78)
    S, G, N = 7, 20011, 100000
    match = rand_match(rng, S, G)
    recs = np.zeros(N, dtype=raft_amd.ACK_DTYPE)
    recs["group"] = rng.integers(0, G + 3, size=N)
    recs["slot"] = rng.integers(0, S, size=N)
    recs["index"] = rng.integers(0, 1 << 62, size=N, dtype=np.uint64)
    want = match.copy()
    skipped = oracle.scatter_acks(want, recs)
    pin = raft_amd.PinnedArray(N, raft_amd.ACK_DTYPE)
    try:
        pin.array[:] = recs
        with raft_amd.Engine(G, S) as e:
            e.upload_match(match)
            e.scatter_acks(pin.array)
            pin.array["index"] = 0  # the call has returned: the buffer is the caller's again
            assert np.array_equal(e.read_match(), want)
            assert e.stats()["acks_rejected"] == skipped > 0
            msgs = np.zeros(N, dtype=raft_amd.MSG_DTYPE)
            msgs["group"], msgs["slot"], msgs["type"] = recs["group"], recs["slot"], 4
            msgs["index"] = recs["index"] + np.uint64(5)
            pm = raft_amd.PinnedArray(N, raft_amd.MSG_DTYPE)
            pm.array[:] = msgs
            active = np.zeros(G, dtype=np.uint8)
            e.upload_active(active)
            e.upload_acks(active)
            e.ingest_msgs(pm.array)
            last = np.zeros(G, dtype=np.uint64)
            oracle.step_leader_msgs(want, active, active.copy(), last, msgs)
            assert np.array_equal(e.read_match(), want) and np.array_equal(e.read_active(), active)
            pm.free()
    finally:
        pin.free()
    with raft_amd.PinnedArray(0, np.uint8) as empty:
        assert empty.size == 0


def test_two_engines_from_two_threads(oracle):
    """Different engines may be driven from different threads (include/rafthip.h
    conventions); each thread's last-error text is its own."""
    import threading
    errs = []

    def worker(seed, S):
        try:
            rng = np.random.default_rng(seed)
            G = 20000 + seed
            with raft_amd.Engine(G, S) as e:
                e.set_uniform_config(S)
                for _ in range(30):
                    match = rand_match(rng, S, G)
                    e.upload_match(match)
                    e.commit()
                    assert np.array_equal(e.read_commit(), oracle.batch_committed(match, uniform_voters=S))
                    with pytest.raises(raft_amd.RafthipError) as ei:
                        e.read_commit(G, 1)
                    assert "[%d, +1)" % G in str(ei.value)
        except BaseException as ex:  # noqa: BLE001 - reported to the main thread
            errs.append(ex)

    ts = [threading.Thread(target=worker, args=(i, s)) for i, s in enumerate((3, 5, 7, 12))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("S", [2, 5, 6, 8, 9, 13, 16, 17, 24, 31, 32, 40, 48, 63, 64])
def test_commit_joint_one_sort_path_and_its_bound(oracle, S):
    """Joint waves take one sort with the half membership in the two low bits of the shifted key when every
    member key of the wavefront is below 2^62 (kernels.h: regs_joint_packed, 17 slots and up), the two-sort or counting form
    otherwise.  Both against the oracle: keys right up to the bound (2^62 - 1), ties, zeros, empty halves; then
    the same planes with one key of 2^62 or more planted in some wavefronts only."""
    rng = np.random.default_rng(900 + S)
    G = 6000 if S > 32 else 20000
    big = rng.integers(0, 1 << 62, size=(S, G), dtype=np.uint64)
    small = rng.integers(0, 6, size=(S, G), dtype=np.uint64)
    sel = rng.random((S, G))
    match = np.where(sel < 0.5, big, small)
    match = np.where(sel > 0.93, np.uint64((1 << 62) - 1), match)
    match = np.ascontiguousarray(np.where((sel > 0.86) & (sel <= 0.93), np.uint64(0), match).astype(np.uint64))
    m0, m1 = rand_masks(rng, S, G, joint_frac=0.8, empty_frac=0.05)
    m0[::97] = 0  # empty incoming half next to a non-empty outgoing one
    with raft_amd.Engine(G, S) as e:
        e.upload_config(m0, m1)
        e.upload_match(match)
        e.commit()
        assert np.array_equal(e.read_commit(), oracle.batch_committed(match, m0, m1))
        # plant keys >= 2^62 in a few groups: their wavefronts leave the one-sort path, the others stay on it
        planted = match.copy()
        gs = rng.choice(G, size=max(1, G // 700), replace=False)
        planted[rng.integers(0, S, size=len(gs)), gs] = rng.integers(1 << 62, 1 << 63, size=len(gs), dtype=np.uint64) * np.uint64(2) - np.uint64(1)
        e.upload_match(planted)
        e.commit()
        assert np.array_equal(e.read_commit(), oracle.batch_committed(planted, m0, m1))


@pytest.mark.parametrize("S", [6, 7, 8, 9, 12, 13, 16, 17, 24, 32, 33, 48, 64])
def test_commit_wide_on_the_f64_sorting_network(oracle, S):
    """6 slots and up sort on the f64 unit (kernels.h: to_f64_order, cex_f64; K1 and K2 up to 8 / 16, K4r above): a u64 key becomes the f64 bit pattern
    of the same rank — keys below 0x7FF0... negative doubles, the rest +0 .. +inf — and the top 2^53 keys, which have
    no pattern, send their wavefront to the counting form.  Keys on both sides of every seam of that map, in every
    mix: 0, the f64 denormals (every real log index is one), the sign change at 0x7FF0..., the limit at 0xFFE0...,
    the packed-joint limit at 0x3FF8..., ties; non-joint (uniform and masked) and joint."""
    rng = np.random.default_rng(2100 + S)
    G = 8192
    seams = np.array([0, 1, 2, (1 << 52) - 1, 1 << 52, (1 << 52) + 1, 0x3FF8 << 48, (0x3FF8 << 48) - 1, (0x3FF8 << 48) + 1,
                      (1 << 62) - 1, 1 << 62, (0x7FF0 << 48) - 1, 0x7FF0 << 48, (0x7FF0 << 48) + 1, (1 << 63) - 1, 1 << 63,
                      (1 << 63) + 1, (0xFFE0 << 48) - 1, 0xFFE0 << 48], dtype=np.uint64)
    beyond = np.array([(0xFFE0 << 48) + 1, (0xFFF0 << 48), (1 << 64) - 2, (1 << 64) - 1], dtype=np.uint64)
    near = seams[rng.integers(0, len(seams), size=(S, G))]
    jitter = rng.integers(0, 3, size=(S, G), dtype=np.uint64)
    near = np.where(near >= 2, near - np.uint64(1), near) + jitter  # never past 0xFFE0...+1: see `beyond`
    near = np.minimum(near, np.uint64(0xFFE0 << 48))
    wide = rng.integers(0, 0xFFE0 << 47, size=(S, G), dtype=np.uint64) * np.uint64(2)
    logs = rng.integers(0, 1 << 40, size=(S, G), dtype=np.uint64)
    sel = rng.random((S, G))
    match = np.ascontiguousarray(np.where(sel < 0.4, near, np.where(sel < 0.7, wide, logs)).astype(np.uint64))
    m0, m1 = rand_masks(rng, S, G, joint_frac=0.5, empty_frac=0.03)
    m1[: G // 2] = 0  # the first half of the wavefronts holds no joint group: regs_half
    cases = [("masked", m0, m1)]
    for name, a, b in cases:
        with raft_amd.Engine(G, S) as e:
            e.upload_config(a, b)
            e.upload_match(match)
            e.commit()
            assert np.array_equal(e.read_commit(), oracle.batch_committed(match, a, b)), name
            # keys without a pattern in a few wavefronts only: those count, the others still sort
            planted = match.copy()
            gs = rng.choice(G, size=24, replace=False)
            planted[rng.integers(0, S, size=len(gs)), gs] = beyond[rng.integers(0, len(beyond), size=len(gs))]
            e.upload_match(planted)
            e.commit()
            assert np.array_equal(e.read_commit(), oracle.batch_committed(planted, a, b)), name
    # uniform configs (K1 up to 8 voters, on the f64 unit from 6: kernels.h uniform_tile), odd and even, with and
    # without keys that have no pattern; changed pairs through the reporting form of the same tile
    for V in (S - 1, S):
        with raft_amd.Engine(G, S) as e:
            e.set_uniform_config(V)
            for planes in (match, planted):
                e.upload_match(planes)
                want = oracle.batch_committed(planes, uniform_voters=V, threads=0)
                before = e.read_commit()
                pairs, n = e.commit_changed()
                assert np.array_equal(e.read_commit(), want), V
                ch = np.flatnonzero(before != want)
                assert n == len(ch) and np.array_equal(pairs["group"], ch) and np.array_equal(pairs["commit"], want[ch]), V


@pytest.mark.parametrize("S", [16, 24, 64])
def test_commit_wide_a_quarter_million_groups_both_distributions(oracle, S):
    """The f64-unit kernels over 2^18 groups (every tile shape, many wavefronts per CU): the reference's bench
    distribution (rand.Int63n: a joint wave is beyond the one-sort bound and sorts per half) and the raft-like one
    (indexes near 2^40 with lagging followers, a few cells >= 2^63), non-joint and joint masks mixed."""
    G = 1 << 18
    rng = np.random.default_rng(2300 + S)
    m0, m1 = rand_masks(rng, S, G, joint_frac=0.5, empty_frac=0.01)
    m1[G // 4: G // 2] = 0  # whole wavefronts without a joint group
    with raft_amd.Engine(G, S) as e:
        e.upload_config(m0, m1)
        for match in (gu.dist_a(77 + S, S, G), gu.dist_b(78 + S, S, G)):
            e.upload_match(match)
            e.commit()
            assert np.array_equal(e.read_commit(), oracle.batch_committed(match, m0, m1))


def test_commit_changed_short_reports_through_the_arena(oracle):
    """A report of at most 4096 pairs into pageable memory comes back through the engine's page-locked arena
    (rafthip.hip commit_report, k_copy_head), a longer one by a second copy: the same pairs either side of that limit,
    with a destination smaller than the report, and again after a long report."""
    G, S = 20000, 5
    rng = np.random.default_rng(2400)
    match = rand_match(rng, S, G)
    with raft_amd.Engine(G, S) as e:
        e.set_uniform_config(S)
        e.upload_match(match)
        e.commit()
        for k in (0, 1, 4095, 4096, 4097, 9000, 3, G, 4096):
            gs = np.sort(rng.choice(G, size=k, replace=False))
            before = oracle.batch_committed(match, uniform_voters=S, threads=0)
            match[:, gs] = match[:, gs] + np.uint64(1) + rng.integers(0, 5, size=(S, k), dtype=np.uint64)
            e.upload_match(match)
            want = oracle.batch_committed(match, uniform_voters=S, threads=0)
            changed = np.flatnonzero(want != before)
            for room in (G, 4096, max(1, len(changed) // 2)):
                pairs, n = e.commit_changed(max_pairs=room)
                if room == G:
                    assert n == len(changed)
                    assert np.array_equal(pairs["group"], changed) and np.array_equal(pairs["commit"], want[changed])
                else:  # nothing changes between two passes over the same planes: an empty report, whatever the room
                    assert n == 0 and len(pairs) == 0
            # a destination smaller than the report: the first `room` pairs, the full count
            e.upload_commit(before)
            room = max(1, len(changed) // 2)
            pairs, n = e.commit_changed(max_pairs=room)
            assert n == len(changed) and np.array_equal(pairs["group"], changed[:min(room, n)])
            assert np.array_equal(pairs["commit"], want[changed][:min(room, n)])
