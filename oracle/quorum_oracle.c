/*
 * quorum_oracle.c — CPU restatement of the reference's quorum / tracker hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP
 * engine.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it; nothing under raft_amd/ (the product) may call into it.
 *
 * The reference is Go (go.etcd.io/raft/v3 fork) and no Go toolchain exists in
 * the build container, so the reference itself cannot be executed.  Parity is
 * pinned instead by the reference's own golden vectors (quorum/testdata/ *.txt,
 * 127 cases, re-encoded under tests/golden/) and test tables (TestCommit,
 * TestLeaderAcknowledgeCommit, TestProgressUpdate), all of which this oracle
 * must reproduce (tests/test_oracle_golden.py).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root).  The data shape follows the reference too: a config is
 * a set of voter ids, an AckedIndexer / vote map is an id-keyed map (here: a
 * small open-addressing hash map, standing in for Go's map).
 */
#include <stdint.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* quorum/quorum.go:48-58 — VoteResult enum; 0 is not a valid value. */
enum { VOTE_PENDING = 1, VOTE_LOST = 2, VOTE_WON = 3 };

/* ------------------------------------------------------------------------ */
/* A tiny id-keyed map: stands in for Go's map[uint64]T on the oracle side.  */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint64_t *keys;
    uint64_t *vals;
    uint8_t *used;
    uint32_t cap; /* power of two */
    uint32_t len;
} idmap;

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

static void idmap_init(idmap *m, uint32_t expect) {
    uint32_t cap = 4;
    while (cap < expect * 2 + 2) cap <<= 1;
    m->cap = cap;
    m->len = 0;
    m->keys = (uint64_t *)calloc(cap, sizeof(uint64_t));
    m->vals = (uint64_t *)calloc(cap, sizeof(uint64_t));
    m->used = (uint8_t *)calloc(cap, 1);
}

static void idmap_free(idmap *m) {
    free(m->keys);
    free(m->vals);
    free(m->used);
}

static void idmap_put(idmap *m, uint64_t k, uint64_t v) {
    uint32_t i = (uint32_t)mix64(k) & (m->cap - 1);
    while (m->used[i]) {
        if (m->keys[i] == k) {
            m->vals[i] = v;
            return;
        }
        i = (i + 1) & (m->cap - 1);
    }
    m->used[i] = 1;
    m->keys[i] = k;
    m->vals[i] = v;
    m->len++;
}

static int idmap_get(const idmap *m, uint64_t k, uint64_t *v) {
    if (m->cap == 0) return 0;
    uint32_t i = (uint32_t)mix64(k) & (m->cap - 1);
    while (m->used[i]) {
        if (m->keys[i] == k) {
            *v = m->vals[i];
            return 1;
        }
        i = (i + 1) & (m->cap - 1);
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* quorum package                                                            */
/* ------------------------------------------------------------------------ */

/* quorum/majority.go:115-122 — insertionSort, ascending. */
static void insertion_sort(uint64_t *sl, int n) {
    for (int i = 1; i < n; i++) {
        for (int j = i; j > 0 && sl[j] < sl[j - 1]; j--) {
            uint64_t t = sl[j];
            sl[j] = sl[j - 1];
            sl[j - 1] = t;
        }
    }
}

/*
 * quorum/majority.go:126-172 — MajorityConfig.CommittedIndex.
 * cfg: the n voter ids of the config.  l: the AckedIndexer
 * (quorum/quorum.go:34-43), an id→index map; an absent id is "not found".
 */
static uint64_t majority_committed_index(const uint64_t *cfg, int n, const idmap *l) {
    if (n == 0) return UINT64_MAX; /* majority.go:128-132 */
    uint64_t stk[7];
    uint64_t *srt = stk;
    if (n > 7) srt = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n); /* :141-147 */
    memset(srt, 0, sizeof(uint64_t) * (size_t)n);
    int i = n - 1; /* :154 fill from the right; unfound voters stay zero */
    for (int k = 0; k < n; k++) {
        uint64_t idx;
        if (idmap_get(l, cfg[k], &idx)) {
            srt[i] = idx;
            i--;
        }
    }
    insertion_sort(srt, n);       /* :165 */
    int pos = n - (n / 2 + 1);    /* :170 */
    uint64_t r = srt[pos];
    if (srt != stk) free(srt);
    return r;
}

/*
 * quorum/quick_test.go:85-122 — alternativeMajorityCommittedIndex, the
 * reference's own independent O(n^2) cross-check.
 */
static uint64_t alt_majority_committed_index(const uint64_t *cfg, int n, const idmap *l) {
    if (n == 0) return UINT64_MAX;
    uint64_t *have = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
    int m = 0;
    for (int k = 0; k < n; k++) {
        uint64_t idx;
        if (idmap_get(l, cfg[k], &idx)) have[m++] = idx; /* idToIdx */
    }
    int q = n / 2 + 1;
    uint64_t best = 0; /* maxQuorumIdx zero value, :114 */
    for (int a = 0; a < m; a++) {      /* each distinct key idy of idxToVotes */
        int votes = 0;
        for (int b = 0; b < m; b++)    /* voters with idx >= idy, :103-110 */
            if (!(have[a] > have[b])) votes++;
        if (votes >= q && have[a] > best) best = have[a];
    }
    free(have);
    return best;
}

/* quorum/joint.go:49-56 — JointConfig.CommittedIndex = min of the halves. */
static uint64_t joint_committed_index(const uint64_t *c0, int n0, const uint64_t *c1, int n1,
                                      const idmap *l) {
    uint64_t idx0 = majority_committed_index(c0, n0, l);
    uint64_t idx1 = majority_committed_index(c1, n1, l);
    return idx0 < idx1 ? idx0 : idx1;
}

/*
 * quorum/majority.go:178-207 — MajorityConfig.VoteResult.
 * votes: id → 0 (false / rejected) or 1 (true / granted); absent = missing.
 */
static int majority_vote_result(const uint64_t *cfg, int n, const idmap *votes) {
    if (n == 0) return VOTE_WON; /* :179-184 */
    int voted = 0, missing = 0;
    for (int k = 0; k < n; k++) {
        uint64_t v;
        if (!idmap_get(votes, cfg[k], &v)) {
            missing++;
            continue;
        }
        if (v) voted++;
    }
    int q = n / 2 + 1;
    if (voted >= q) return VOTE_WON;
    if (voted + missing >= q) return VOTE_PENDING;
    return VOTE_LOST;
}

/* quorum/joint.go:61-75 — JointConfig.VoteResult. */
static int joint_vote_result(const uint64_t *c0, int n0, const uint64_t *c1, int n1,
                             const idmap *votes) {
    int r1 = majority_vote_result(c0, n0, votes);
    int r2 = majority_vote_result(c1, n1, votes);
    if (r1 == r2) return r1;
    if (r1 == VOTE_LOST || r2 == VOTE_LOST) return VOTE_LOST;
    return VOTE_PENDING;
}

/* ------------------------------------------------------------------------ */
/* Exported per-group entry points (id-level; used by the golden tests).     */
/* A map is passed as parallel arrays (ids, vals, count).                    */
/* ------------------------------------------------------------------------ */

static void build_map(idmap *m, const uint64_t *ids, const uint64_t *vals, int n) {
    idmap_init(m, (uint32_t)n);
    for (int i = 0; i < n; i++) idmap_put(m, ids[i], vals[i]);
}

ORACLE_API uint64_t oracle_majority_committed_index(const uint64_t *cfg, int n,
                                                    const uint64_t *l_ids, const uint64_t *l_idx,
                                                    int l_n) {
    idmap l;
    build_map(&l, l_ids, l_idx, l_n);
    uint64_t r = majority_committed_index(cfg, n, &l);
    idmap_free(&l);
    return r;
}

ORACLE_API uint64_t oracle_alt_majority_committed_index(const uint64_t *cfg, int n,
                                                        const uint64_t *l_ids,
                                                        const uint64_t *l_idx, int l_n) {
    idmap l;
    build_map(&l, l_ids, l_idx, l_n);
    uint64_t r = alt_majority_committed_index(cfg, n, &l);
    idmap_free(&l);
    return r;
}

ORACLE_API uint64_t oracle_joint_committed_index(const uint64_t *c0, int n0, const uint64_t *c1,
                                                 int n1, const uint64_t *l_ids,
                                                 const uint64_t *l_idx, int l_n) {
    idmap l;
    build_map(&l, l_ids, l_idx, l_n);
    uint64_t r = joint_committed_index(c0, n0, c1, n1, &l);
    idmap_free(&l);
    return r;
}

ORACLE_API int oracle_majority_vote_result(const uint64_t *cfg, int n, const uint64_t *v_ids,
                                           const uint64_t *v_vals, int v_n) {
    idmap v;
    build_map(&v, v_ids, v_vals, v_n);
    int r = majority_vote_result(cfg, n, &v);
    idmap_free(&v);
    return r;
}

ORACLE_API int oracle_joint_vote_result(const uint64_t *c0, int n0, const uint64_t *c1, int n1,
                                        const uint64_t *v_ids, const uint64_t *v_vals, int v_n) {
    idmap v;
    build_map(&v, v_ids, v_vals, v_n);
    int r = joint_vote_result(c0, n0, c1, n1, &v);
    idmap_free(&v);
    return r;
}

/*
 * tracker/progress.go:168-177 — Progress.MaybeUpdate.  Returns `updated`.
 * (MsgAppFlowPaused is host-side flow control and is not modelled.)
 */
ORACLE_API int oracle_maybe_update(uint64_t *match, uint64_t *next, uint64_t n) {
    int updated = 0;
    if (*match < n) {
        *match = n;
        updated = 1;
    }
    if (*next < n + 1) *next = n + 1;
    return updated;
}

/*
 * log.go:447-456 — raftLog.maybeCommit's guard, the consumer of the commit
 * index (raft.go:755-758).  term_of_max is zeroTermOnOutOfBounds(term(maxIndex)).
 * Returns the new committed value.
 */
ORACLE_API uint64_t oracle_log_maybe_commit(uint64_t committed, uint64_t max_index,
                                            uint64_t term_of_max, uint64_t term) {
    if (max_index > committed && term != 0 && term_of_max == term) return max_index;
    return committed;
}

/*
 * Differential self-check in the spirit of quorum/quick_test.go:28-64, at a
 * volume the Python loop cannot reach: `cases` random (config, indexer) pairs,
 * CommittedIndex against the reference's own alternative algorithm
 * (quick_test.go:85-122).  Generators: up to 64 voters, ids from a small pool
 * (so config and indexer overlap only partly: missing ids), and index values
 * that are small with many ties, zero, or full-range including >= 2^63.
 * Returns the number of mismatches.
 */
ORACLE_API uint64_t oracle_selfcheck_committed(uint64_t cases, uint64_t seed) {
    uint64_t bad = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : bad) schedule(static)
#endif
    for (int64_t c = 0; c < (int64_t)cases; c++) {
        uint64_t x = seed + (uint64_t)c * 0x9E3779B97F4A7C15ULL;
#define ORACLE_NEXT(dst) do { x += 0x9E3779B97F4A7C15ULL; uint64_t z_ = x; \
        z_ = (z_ ^ (z_ >> 30)) * 0xBF58476D1CE4E5B9ULL; z_ = (z_ ^ (z_ >> 27)) * 0x94D049BB133111EBULL; \
        dst = z_ ^ (z_ >> 31); } while (0)
        uint64_t r;
        ORACLE_NEXT(r);
        int wide = (r & 15) == 0;
        int n = (int)((r >> 8) % (wide ? 65 : 10)); /* mostly <= 9 ids like the reference, sometimes up to 64 */
        int pool = n + 1 + (int)((r >> 20) % 8);
        uint64_t cfg[64], ids[80], vals[80];
        uint64_t used = 0, used_hi = 0; /* pool <= 72 */
        int k = 0;
        while (k < n) {
            ORACLE_NEXT(r);
            int id = (int)(r % (uint64_t)pool);
            uint64_t *w = id < 64 ? &used : &used_hi;
            uint64_t bit = 1ULL << (id & 63);
            if (*w & bit) continue;
            *w |= bit;
            cfg[k++] = (uint64_t)id + 1;
        }
        int ln = 0;
        for (int id = 0; id < pool; id++) {
            ORACLE_NEXT(r);
            if ((r & 3) == 0) continue; /* not in the indexer */
            ids[ln] = (uint64_t)id + 1;
            uint64_t kind = (r >> 2) & 7, v;
            ORACLE_NEXT(v);
            vals[ln] = kind < 4 ? v % (uint64_t)(n + 1) : kind == 4 ? 0 : kind == 5 ? (v | (1ULL << 63)) : v;
            ln++;
        }
#undef ORACLE_NEXT
        idmap l;
        build_map(&l, ids, vals, ln);
        if (majority_committed_index(cfg, n, &l) != alt_majority_committed_index(cfg, n, &l)) bad++;
        idmap_free(&l);
    }
    return bad;
}

/* ------------------------------------------------------------------------ */
/* Batched SoA views.  The engine's layout is [slot][group] planes plus two  */
/* per-group slot bitmasks (incoming = Voters[0], outgoing = Voters[1]).     */
/* The oracle turns every group back into the reference's shape (id sets +   */
/* id-keyed maps, slot number = voter id) and runs the functions above.      */
/* ------------------------------------------------------------------------ */

static uint64_t load_mask(const void *masks, int mask_bytes, uint64_t g) {
    switch (mask_bytes) {
    case 1: return ((const uint8_t *)masks)[g];
    case 2: return ((const uint16_t *)masks)[g];
    case 4: return ((const uint32_t *)masks)[g];
    default: return ((const uint64_t *)masks)[g];
    }
}

static int mask_to_ids(uint64_t mask, int nslots, uint64_t *ids) {
    int n = 0;
    for (int s = 0; s < nslots; s++)
        if ((mask >> s) & 1) ids[n++] = (uint64_t)s;
    return n;
}

/*
 * tracker/tracker.go:164-181 — ProgressTracker.Committed over G groups:
 * Voters.CommittedIndex(matchAckIndexer(Progress)).  Every slot has a
 * Progress row (voter or learner), so AckedIndex is always "found" and
 * returns Match (tracker.go:169-175).  in_mask==NULL means "uniform": all
 * groups have Voters[0] = slots [0,uniform_voters), Voters[1] empty.
 */
ORACLE_API void oracle_batch_committed(uint64_t G, int nslots, const uint64_t *match,
                                       uint64_t stride, const void *in_mask,
                                       const void *out_mask, int mask_bytes,
                                       int uniform_voters, uint64_t *out, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
    {
        uint64_t ids0[64], ids1[64], all[64], vals[64];
        for (int s = 0; s < nslots; s++) all[s] = (uint64_t)s;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t g = 0; g < (int64_t)G; g++) {
            uint64_t m0, m1;
            if (in_mask) {
                m0 = load_mask(in_mask, mask_bytes, (uint64_t)g);
                m1 = out_mask ? load_mask(out_mask, mask_bytes, (uint64_t)g) : 0;
            } else {
                m0 = uniform_voters >= 64 ? ~0ULL : ((1ULL << uniform_voters) - 1);
                m1 = 0;
            }
            int n0 = mask_to_ids(m0, nslots, ids0);
            int n1 = mask_to_ids(m1, nslots, ids1);
            for (int s = 0; s < nslots; s++) vals[s] = match[(uint64_t)s * stride + (uint64_t)g];
            idmap l;
            build_map(&l, all, vals, nslots);
            out[g] = joint_committed_index(ids0, n0, ids1, n1, &l);
            idmap_free(&l);
        }
    }
}

/*
 * tracker/tracker.go:269-290 — TallyVotes over G groups.  votes plane cell:
 * 0 = no vote recorded, 1 = rejected, 2 = granted (the encoding the
 * reference's own harness uses, quorum/datadriven_test.go:55-58,94-100).
 * learner_mask marks slots whose Progress.IsLearner is set (excluded from
 * granted/rejected, tracker.go:275-277); all nslots slots have a Progress row.  result uses Voters only (:288).
 * granted/rejected may be NULL.
 */
ORACLE_API void oracle_batch_tally_votes(uint64_t G, int nslots, const uint8_t *votes,
                                         uint64_t stride, const void *in_mask,
                                         const void *out_mask, const void *learner_mask,
                                         int mask_bytes, int uniform_voters, uint8_t *result,
                                         uint8_t *granted, uint8_t *rejected) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t gi = 0; gi < (int64_t)G; gi++) {
        const uint64_t g = (uint64_t)gi;
        uint64_t ids0[64], ids1[64], vid[64], vval[64];
        uint64_t m0, m1, ml;
        if (in_mask) {
            m0 = load_mask(in_mask, mask_bytes, g);
            m1 = out_mask ? load_mask(out_mask, mask_bytes, g) : 0;
            ml = learner_mask ? load_mask(learner_mask, mask_bytes, g) : 0;
        } else {
            m0 = uniform_voters >= 64 ? ~0ULL : ((1ULL << uniform_voters) - 1);
            m1 = 0;
            ml = nslots >= 64 ? ~m0 : (((1ULL << nslots) - 1) & ~m0);
        }
        int n0 = mask_to_ids(m0, nslots, ids0);
        int n1 = mask_to_ids(m1, nslots, ids1);
        int nv = 0, gr = 0, rj = 0;
        for (int s = 0; s < nslots; s++) {
            uint8_t c = votes[(uint64_t)s * stride + g];
            if (c == 0) continue; /* no entry in Votes */
            vid[nv] = (uint64_t)s;
            vval[nv] = (c == 2);
            nv++;
            if (!((ml >> s) & 1)) { /* tracker.go:274-287 */
                if (c == 2) gr++;
                else rj++;
            }
        }
        idmap v;
        build_map(&v, vid, vval, nv);
        result[g] = (uint8_t)joint_vote_result(ids0, n0, ids1, n1, &v);
        idmap_free(&v);
        if (granted) granted[g] = (uint8_t)gr;
        if (rejected) rejected[g] = (uint8_t)rj;
    }
}

/*
 * tracker/tracker.go:217-227 — QuorumActive over G groups.  active bit s =
 * Progress[s].RecentActive; learner slots are skipped (:220-222), every other
 * Progress row gets an entry (true or false) in the votes map.
 */
ORACLE_API void oracle_batch_quorum_active(uint64_t G, int nslots, const void *active,
                                           const void *in_mask, const void *out_mask,
                                           const void *learner_mask, int mask_bytes,
                                           int uniform_voters, uint8_t *out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t gi = 0; gi < (int64_t)G; gi++) {
        const uint64_t g = (uint64_t)gi;
        uint64_t ids0[64], ids1[64], vid[64], vval[64];
        uint64_t m0, m1, ml;
        if (in_mask) {
            m0 = load_mask(in_mask, mask_bytes, g);
            m1 = out_mask ? load_mask(out_mask, mask_bytes, g) : 0;
            ml = learner_mask ? load_mask(learner_mask, mask_bytes, g) : 0;
        } else {
            m0 = uniform_voters >= 64 ? ~0ULL : ((1ULL << uniform_voters) - 1);
            m1 = 0;
            ml = nslots >= 64 ? ~m0 : (((1ULL << nslots) - 1) & ~m0);
        }
        uint64_t act = load_mask(active, mask_bytes, g);
        int n0 = mask_to_ids(m0, nslots, ids0);
        int n1 = mask_to_ids(m1, nslots, ids1);
        int nv = 0;
        for (int s = 0; s < nslots; s++) {
            if ((ml >> s) & 1) continue;
            vid[nv] = (uint64_t)s;
            vval[nv] = (act >> s) & 1;
            nv++;
        }
        idmap v;
        build_map(&v, vid, vval, nv);
        out[g] = joint_vote_result(ids0, n0, ids1, n1, &v) == VOTE_WON;
        idmap_free(&v);
    }
}

/*
 * read_only.go:68-76 + raft.go:1552 — ReadIndex ack quorum over G groups.
 * acks bit s set = acks[id]=true; the map never holds false (read_only.go:32-36),
 * so an unset bit is a missing entry.  Output is the VoteResult enum.
 */
ORACLE_API void oracle_batch_ack_quorum(uint64_t G, int nslots, const void *acks,
                                        const void *in_mask, const void *out_mask,
                                        int mask_bytes, int uniform_voters, uint8_t *out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t gi = 0; gi < (int64_t)G; gi++) {
        const uint64_t g = (uint64_t)gi;
        uint64_t ids0[64], ids1[64], vid[64], vval[64];
        uint64_t m0, m1;
        if (in_mask) {
            m0 = load_mask(in_mask, mask_bytes, g);
            m1 = out_mask ? load_mask(out_mask, mask_bytes, g) : 0;
        } else {
            m0 = uniform_voters >= 64 ? ~0ULL : ((1ULL << uniform_voters) - 1);
            m1 = 0;
        }
        uint64_t a = load_mask(acks, mask_bytes, g);
        int n0 = mask_to_ids(m0, nslots, ids0);
        int n1 = mask_to_ids(m1, nslots, ids1);
        int nv = 0;
        for (int s = 0; s < nslots; s++) {
            if (!((a >> s) & 1)) continue;
            vid[nv] = (uint64_t)s;
            vval[nv] = 1;
            nv++;
        }
        idmap v;
        build_map(&v, vid, vval, nv);
        out[g] = (uint8_t)joint_vote_result(ids0, n0, ids1, n1, &v);
        idmap_free(&v);
    }
}

/*
 * raft.go:1333,1467-1477 → tracker/progress.go:168-177 — apply a stream of
 * non-reject MsgAppResp acks in order: Match = max(Match, index) per
 * (group, slot).  Records with group >= G or slot >= nslots are skipped and
 * counted (the reference drops messages from unknown peers, raft.go:1327-1331).
 * Returns the number skipped.
 */
typedef struct {
    uint32_t group;
    uint32_t slot;
    uint64_t index;
} oracle_ack;

ORACLE_API uint64_t oracle_scatter_acks(uint64_t G, int nslots, uint64_t *match, uint64_t stride,
                                        const oracle_ack *recs, uint64_t n) {
    uint64_t skipped = 0;
    for (uint64_t i = 0; i < n; i++) {
        if (recs[i].group >= G || recs[i].slot >= (uint32_t)nslots) {
            skipped++;
            continue;
        }
        uint64_t *cell = &match[(uint64_t)recs[i].slot * stride + recs[i].group];
        uint64_t next = 0;
        oracle_maybe_update(cell, &next, recs[i].index);
    }
    return skipped;
}

/* ------------------------------------------------------------------------ */
/* The steps either side of the quorum arithmetic.           */
/* ------------------------------------------------------------------------ */

/*
 * A raftLog reduced to what term() needs: entries first_index..first_index+n-1
 * with their terms, and the term of the compacted entry first_index-1
 * (snapshot / dummy entry, storage.go MemoryStorage.ents[0]).
 */
typedef struct {
    uint64_t first_index; /* raftLog.firstIndex() */
    uint64_t snap_term;   /* term of entry first_index-1 */
    const uint64_t *terms;
    uint64_t n;
} oracle_log;

/* log.go:381-407 — raftLog.term, wrapped in zeroTermOnOutOfBounds (log.go:559-568):
 * ErrCompacted and ErrUnavailable become term 0. */
static uint64_t log_term(const oracle_log *l, uint64_t i, int *err) {
    uint64_t last = l->first_index + l->n - 1; /* lastIndex(); first_index-1 when empty */
    *err = 0;
    if (i + 1 < l->first_index) { *err = 1; return 0; } /* ErrCompacted (log.go:383-386) */
    if (i > last) { *err = 2; return 0; }               /* ErrUnavailable (log.go:387-389) */
    if (i + 1 == l->first_index) return l->snap_term;
    return l->terms[i - l->first_index];
}
static uint64_t log_term_or_zero(const oracle_log *l, uint64_t i) {
    int err;
    return log_term(l, i, &err);
}

/* raftLog.term itself (log.go:381-407), for the reference's own table (log_test.go:830-858 TestTerm):
 * *err = 0 ok, 1 ErrCompacted, 2 ErrUnavailable. */
ORACLE_API uint64_t oracle_raftlog_term(uint64_t first_index, uint64_t snap_term, const uint64_t *terms,
                                        uint64_t n, uint64_t i, int *err) {
    oracle_log l = {first_index, snap_term, terms, n};
    return log_term(&l, i, err);
}

/* log.go:447-456 — raftLog.maybeCommit; commitTo (log.go:318-326) never
 * decreases.  Returns 1 and moves *committed when the guard passes. */
static int raftlog_maybe_commit(const oracle_log *l, uint64_t *committed, uint64_t max_index,
                                uint64_t term) {
    if (max_index > *committed && term != 0 && log_term_or_zero(l, max_index) == term) {
        if (*committed < max_index) *committed = max_index;
        return 1;
    }
    return 0;
}

/* raft.go:2001-2005 — committedEntryInCurrentTerm on an explicit log. */
ORACLE_API int oracle_committed_entry_in_current_term(uint64_t first_index, uint64_t snap_term,
                                                      const uint64_t *terms, uint64_t n,
                                                      uint64_t committed, uint64_t term) {
    oracle_log l = {first_index, snap_term, terms, n};
    return log_term_or_zero(&l, committed) == term;
}

ORACLE_API int oracle_raftlog_maybe_commit(uint64_t first_index, uint64_t snap_term,
                                           const uint64_t *terms, uint64_t n, uint64_t *committed,
                                           uint64_t max_index, uint64_t term) {
    oracle_log l = {first_index, snap_term, terms, n};
    return raftlog_maybe_commit(&l, committed, max_index, term);
}

/* log.go:178-190 — raftLog.findConflictByTerm: the largest guess <= index whose term is <= term,
 * or whose term is unknown (compacted / not yet stored: log.go:381-407 returns an error there, which
 * is not the same as a term of 0).  *term_out = term(guess), 0 when unknown.  The leader runs it on
 * the RejectHint / LogTerm of a rejected MsgAppResp (raft.go:1460-1461) before MaybeDecrTo. */
ORACLE_API uint64_t oracle_find_conflict_by_term(uint64_t first_index, uint64_t snap_term,
                                                 const uint64_t *terms, uint64_t n, uint64_t index,
                                                 uint64_t term, uint64_t *term_out) {
    oracle_log l = {first_index, snap_term, terms, n};
    uint64_t last = l.first_index + l.n - 1;
    for (; index > 0; index--) {
        if (index + 1 < l.first_index || index > last) { /* ErrCompacted / ErrUnavailable */
            if (term_out) *term_out = 0;
            return index;
        }
        uint64_t ours = log_term_or_zero(&l, index);
        if (ours <= term) {
            if (term_out) *term_out = ours;
            return index;
        }
    }
    if (term_out) *term_out = 0;
    return 0;
}

/*
 * raft.go:755-758 — raft.maybeCommit over G groups: mci = trk.Committed(), then
 * raftLog.maybeCommit(mci, r.Term).  Group g's log is entries
 * log_terms[log_off[g] .. log_off[g+1]) starting at index log_first[g].
 * committed[] is updated in place; advanced[g] = the bool maybeCommit returned.
 */
ORACLE_API void oracle_batch_maybe_commit(uint64_t G, int nslots, const uint64_t *match,
                                          uint64_t stride, const void *in_mask,
                                          const void *out_mask, int mask_bytes, int uniform_voters,
                                          const uint64_t *log_off, const uint64_t *log_terms,
                                          const uint64_t *log_first, const uint64_t *log_snap_term,
                                          const uint64_t *term, uint64_t *committed,
                                          uint8_t *advanced) {
    uint64_t *mci = (uint64_t *)malloc(sizeof(uint64_t) * (G ? G : 1));
    oracle_batch_committed(G, nslots, match, stride, in_mask, out_mask, mask_bytes, uniform_voters,
                           mci, 0);
    for (uint64_t g = 0; g < G; g++) {
        oracle_log l = {log_first[g], log_snap_term[g], log_terms + log_off[g],
                        log_off[g + 1] - log_off[g]};
        advanced[g] = (uint8_t)raftlog_maybe_commit(&l, &committed[g], mci[g], term[g]);
    }
    free(mci);
}

static void store_mask(void *masks, int mask_bytes, uint64_t g, uint64_t v) {
    switch (mask_bytes) {
    case 1: ((uint8_t *)masks)[g] = (uint8_t)v; break;
    case 2: ((uint16_t *)masks)[g] = (uint16_t)v; break;
    case 4: ((uint32_t *)masks)[g] = (uint32_t)v; break;
    default: ((uint64_t *)masks)[g] = v; break;
    }
}

/*
 * raft.go:1327-1561 — stepLeader's per-message bookkeeping, one message at a
 * time in order:
 *   no Progress for m.From -> return (:1327-1331)
 *   MsgAppResp (4):       pr.RecentActive = true (:1337); !Reject -> MaybeUpdate(Index) (:1477)
 *   MsgHeartbeatResp (9): pr.RecentActive = true (:1528); Context names the
 *                         pending request -> readOnly.recvAck: acks[from]=true
 *                         (:1552, read_only.go:68-76)
 * plus the local 0xF0 record: the leader's lastIndex grew (appendEntry,
 * raft.go:787-819), tracked as last_index = max(last_index, index).
 * counts[0] = dropped (no Progress), counts[1] = other message types.
 */
typedef struct {
    uint32_t group;
    uint8_t slot, type, reject, flags;
    uint64_t index;
} oracle_msg;

ORACLE_API void oracle_step_leader_msgs(uint64_t G, int nslots, uint64_t *match, uint64_t stride,
                                        void *active, void *acks, int mask_bytes,
                                        uint64_t *last_index, const oracle_msg *msgs, uint64_t n,
                                        uint64_t *counts) {
    counts[0] = counts[1] = 0;
    for (uint64_t i = 0; i < n; i++) {
        const oracle_msg *m = &msgs[i];
        if (m->type == 0xF0) {
            if (m->group >= G) { counts[0]++; continue; }
            if (last_index[m->group] < m->index) last_index[m->group] = m->index;
            continue;
        }
        if (m->type != 4 && m->type != 9) { counts[1]++; continue; }
        if (m->group >= G || m->slot >= nslots) { counts[0]++; continue; }
        store_mask(active, mask_bytes, m->group, load_mask(active, mask_bytes, m->group) | (1ULL << m->slot));
        if (m->type == 4) {
            if (!m->reject) {
                uint64_t next = 0;
                oracle_maybe_update(&match[(uint64_t)m->slot * stride + m->group], &next, m->index);
            }
        } else if (m->flags & 1) {
            store_mask(acks, mask_bytes, m->group, load_mask(acks, mask_bytes, m->group) | (1ULL << m->slot));
        }
    }
}

/*
 * tracker/tracker.go:260-265 — RecordVote for a stream of vote responses, one at
 * a time in order: `_, ok := p.Votes[id]; if !ok { p.Votes[id] = v }`.  votes is
 * the [slot][stride] cell plane (0 none, 1 rejected, 2 granted); records of type
 * MsgVoteResp (6) / MsgPreVoteResp (18) count, vote = !reject.  Returns the
 * number of records dropped (other types, no Progress row).
 */
ORACLE_API uint64_t oracle_record_votes(uint64_t G, int nslots, uint8_t *votes, uint64_t stride,
                                        const oracle_msg *msgs, uint64_t n) {
    uint64_t dropped = 0;
    for (uint64_t i = 0; i < n; i++) {
        const oracle_msg *m = &msgs[i];
        if ((m->type != 6 && m->type != 18) || m->group >= G || m->slot >= nslots) {
            dropped++;
            continue;
        }
        uint8_t *cell = &votes[(uint64_t)m->slot * stride + m->group];
        if (*cell == 0) *cell = m->reject ? 1 : 2;
    }
    return dropped;
}

/* ------------------------------------------------------------------------ */
/* Serialized raftpb.Message -> stepLeader bookkeeping.                       */
/* ------------------------------------------------------------------------ */

/* What Message.Unmarshal (raftpb/raft.pb.go:1801-2179) leaves in the struct,
 * as far as this path reads it.  Nested messages (entries :1944-1977,
 * snapshot :1997-2032, responses :2125-2158) are only measured: a non-empty
 * one sets nested, and the caller leaves the message to a full decoder. */
typedef struct {
    int32_t type;
    uint64_t to, from, term, log_term, index, commit, reject_hint, vote;
    int reject;
    int has_ctx;          /* Context != nil */
    uint64_t ctx_off, ctx_len;
    int nested;
} oracle_pbmsg;

enum { PB_OK = 0, PB_EOF = -1, PB_OVERFLOW = -2, PB_BAD_LENGTH = -3, PB_BAD_WIRETYPE = -4,
       PB_BAD_TAG = -5, PB_END_GROUP = -6 };

/* The varint loop of every field (e.g. :1807-1820): shift >= 64 is checked
 * before the read, so an 11th byte is an overflow; bits past 63 fall off. */
static int pb_varint(const uint8_t *d, uint64_t l, uint64_t *i, uint64_t *v) {
    *v = 0;
    for (unsigned shift = 0;; shift += 7) {
        if (shift >= 64) return PB_OVERFLOW;
        if (*i >= l) return PB_EOF;
        uint8_t b = d[(*i)++];
        *v |= (uint64_t)(b & 0x7F) << shift;
        if (b < 0x80) return PB_OK;
    }
}

/* skipRaft, raft.pb.go:3027-3104: length of one field of any wire type at d[0..l). */
static int pb_skip(const uint8_t *d, uint64_t l, int64_t *n) {
    uint64_t i = 0;
    int depth = 0, rc;
    while (i < l) {
        uint64_t wire, tmp;
        if ((rc = pb_varint(d, l, &i, &wire))) return rc;
        switch ((int)(wire & 7)) {
        case 0:
            if ((rc = pb_varint(d, l, &i, &tmp))) return rc;
            break;
        case 1: i += 8; break;
        case 2:
            if ((rc = pb_varint(d, l, &i, &tmp))) return rc;
            if ((int64_t)tmp < 0) return PB_BAD_LENGTH;
            i += tmp;
            break;
        case 3: depth++; break;
        case 4:
            if (depth == 0) return PB_END_GROUP;
            depth--;
            break;
        case 5: i += 4; break;
        default: return PB_BAD_WIRETYPE;
        }
        if ((int64_t)i < 0) return PB_BAD_LENGTH;
        if (depth == 0) { *n = (int64_t)i; return PB_OK; }
    }
    return PB_EOF;
}

ORACLE_API int oracle_message_unmarshal(const uint8_t *d, uint64_t l, oracle_pbmsg *m) {
    memset(m, 0, sizeof(*m));
    uint64_t i = 0;
    int rc;
    while (i < l) {
        uint64_t pre = i, wire, v;
        if ((rc = pb_varint(d, l, &i, &wire))) return rc;
        int32_t field = (int32_t)(wire >> 3);
        int wt = (int)(wire & 7);
        if (wt == 4) return PB_END_GROUP;
        if (field <= 0) return PB_BAD_TAG;
        switch (field) {
        case 1: case 2: case 3: case 4: case 5: case 6: case 8: case 10: case 11: case 13:
            if (wt != 0) return PB_BAD_WIRETYPE;
            if ((rc = pb_varint(d, l, &i, &v))) return rc;
            switch (field) {
            case 1: m->type = (int32_t)v; break; /* MessageType is int32: shifts >= 32 contribute 0 */
            case 2: m->to = v; break;
            case 3: m->from = v; break;
            case 4: m->term = v; break;
            case 5: m->log_term = v; break;
            case 6: m->index = v; break;
            case 8: m->commit = v; break;
            case 10: m->reject = v != 0; break;
            case 11: m->reject_hint = v; break;
            default: m->vote = v; break;
            }
            break;
        case 7: case 9: case 12: case 14: {
            if (wt != 2) return PB_BAD_WIRETYPE;
            if ((rc = pb_varint(d, l, &i, &v))) return rc;
            if ((int64_t)v < 0) return PB_BAD_LENGTH;
            uint64_t post = i + v;
            if ((int64_t)post < 0) return PB_BAD_LENGTH;
            if (post > l) return PB_EOF;
            if (field == 12) { /* :2101-2104: the last occurrence replaces the value; never nil afterwards */
                m->has_ctx = 1;
                m->ctx_off = i;
                m->ctx_len = v;
            } else if (v > 0) {
                m->nested = 1;
            }
            i = post;
            break;
        }
        default: {
            int64_t skippy;
            i = pre;
            if ((rc = pb_skip(d + i, l - i, &skippy))) return rc;
            if (skippy < 0 || (int64_t)(i + (uint64_t)skippy) < 0) return PB_BAD_LENGTH;
            if (i + (uint64_t)skippy > l) return PB_EOF;
            i += (uint64_t)skippy;
        }
        }
    }
    return PB_OK;
}

/*
 * A batch of serialized messages stepped one at a time on a leader, message i
 * addressed to group groups[i]:
 *   Message.Unmarshal; an error -> status 6.  Types other than MsgAppResp (4)
 *   and MsgHeartbeatResp (9) -> status 1 (not this path); a non-empty nested
 *   payload -> status 5 (left to a full decoder).
 *   raft.Step's term gate, raft.go:1051-1139: Term 0 is local; Term > r.Term
 *   -> status 4 (becomeFollower: the host's business); Term < r.Term -> status 3.
 *   stepLeader: no Progress for m.From -> status 2 (raft.go:1327-1331); else
 *   RecentActive = true; MsgAppResp && !Reject: MaybeUpdate(Index) (:1477);
 *   MsgHeartbeatResp with the pending ReadIndex context: recvAck (:1552,
 *   read_only.go:68-76).
 * ids is the ProgressMap key of each slot, [nslots][ids_stride], 0 = no row.
 *
 * With campaign != NULL the groups may also be campaigning: campaign[g] is
 * stepCandidate's myVoteRespType (raft.go:1628-1633; 0 = not a candidate).  A
 * MsgVoteResp (6) / MsgPreVoteResp (18) passes the term gate — a granted
 * MsgPreVoteResp also with a higher term, raft.go:1068-1076 — and, when it is
 * the group's own type and its sender has a row, is polled: RecordVote, first
 * vote sticks (raft.go:1041-1049, tracker/tracker.go:260-265) -> status 9.
 * The other type / a group that is not campaigning -> status 1.
 */
static void wire_msgs(int apply, uint64_t G, int nslots, uint64_t *match, uint64_t stride,
                      void *active, void *acks, int mask_bytes, const uint64_t *ids,
                      uint64_t ids_stride, const uint64_t *term,
                      const uint64_t *read_ctx, const uint8_t *read_ctx_len,
                      const uint8_t *bytes, uint64_t nbytes, const uint64_t *offsets,
                      const uint32_t *groups, uint64_t n, oracle_msg *out_msgs, uint64_t *out_aux,
                      uint8_t *out_status, uint64_t *counts, const uint8_t *campaign, uint8_t *votes,
                      uint64_t votes_stride) {
    for (int k = 0; k < 10; k++) counts[k] = 0;
    for (uint64_t i = 0; i < n; i++) {
        oracle_msg o = {groups[i], 0xFF, 0xFF, 0, 0, 0};
        oracle_pbmsg m;
        int st, decoded = 0;
        uint64_t aux = 0;
        uint64_t g = groups[i];
        if (g >= G) {
            st = 7;
        } else if (!(offsets[i] <= offsets[i + 1] && offsets[i + 1] <= nbytes)) {
            st = 6;
        } else if (oracle_message_unmarshal(bytes + offsets[i], offsets[i + 1] - offsets[i], &m)) {
            st = 6;
        } else {
            decoded = 1;
            o.type = (uint32_t)m.type <= 0xFF ? (uint8_t)m.type : 0xFF;
            o.reject = (uint8_t)m.reject;
            o.index = m.index;
            if (m.ctx_len > 0) o.flags |= 2;
            int vote = campaign != NULL && (m.type == 6 || m.type == 18);
            if (m.type != 4 && m.type != 9 && !vote) {
                st = 1;
            } else if (m.nested) {
                st = 5;
            } else if (m.term != 0 && m.term > term[g] && !(m.type == 18 && !m.reject)) {
                st = 4;
            } else if (m.term != 0 && m.term < term[g]) {
                st = 3;
            } else if (vote) {
                if (campaign[g] == 0 || (uint32_t)m.type != campaign[g]) {
                    st = 1; /* stepLeader / stepFollower / the other candidate state: ignored */
                } else {
                    int slot = -1;
                    for (int s = 0; s < nslots && slot < 0; s++)
                        if (m.from != 0 && ids[(uint64_t)s * ids_stride + g] == m.from) slot = s;
                    if (slot < 0) {
                        st = 2; /* recorded under an id nothing reads: neither result nor counts change */
                    } else {
                        st = 9;
                        o.slot = (uint8_t)slot;
                        uint8_t *cell = &votes[(uint64_t)slot * votes_stride + g];
                        if (*cell == 0) *cell = m.reject ? 1 : 2; /* RecordVote */
                    }
                }
            } else {
                int slot = -1;
                for (int s = 0; s < nslots && slot < 0; s++) /* r.trk.Progress[m.From] */
                    if (m.from != 0 && ids[(uint64_t)s * ids_stride + g] == m.from) slot = s;
                if (slot < 0) {
                    st = 2;
                } else {
                    st = 0;
                    o.slot = (uint8_t)slot;
                    if (apply) store_mask(active, mask_bytes, g, load_mask(active, mask_bytes, g) | (1ULL << slot));
                    if (m.type == 4) {
                        if (!apply) {
                            /* decode for the Progress stepper: a rejection's nextProbeIdx is RejectHint
                             * unless LogTerm > 0 asks for findConflictByTerm on the leader's log
                             * (raft.go:1362-1459) — then the host steps it: status 8, record type 0xFE */
                            if (m.reject) {
                                aux = m.reject_hint;
                                if (m.log_term > 0) {
                                    st = 8;
                                    o.type = 0xFE;
                                }
                            }
                        } else if (!m.reject) {
                            uint64_t next = 0;
                            oracle_maybe_update(&match[(uint64_t)slot * stride + g], &next, m.index);
                        }
                    } else if (read_ctx_len[g] != 0 && m.ctx_len == read_ctx_len[g]) {
                        /* pendingReadIndex[string(m.Context)] — the one request tracked per group */
                        uint64_t want = read_ctx[g];
                        int same = 1;
                        for (uint64_t b = 0; b < m.ctx_len; b++)
                            if (bytes[offsets[i] + m.ctx_off + b] != (uint8_t)(want >> (8 * b))) same = 0;
                        if (same) {
                            o.flags |= 1;
                            if (apply) store_mask(acks, mask_bytes, g, load_mask(acks, mask_bytes, g) | (1ULL << slot));
                        }
                    }
                }
            }
        }
        (void)decoded;
        counts[st]++;
        if (out_msgs) out_msgs[i] = o;
        if (out_aux) out_aux[i] = aux;
        if (out_status) out_status[i] = (uint8_t)st;
    }
}

ORACLE_API void oracle_step_wire_msgs(uint64_t G, int nslots, uint64_t *match, uint64_t stride,
                                      void *active, void *acks, int mask_bytes, const uint64_t *ids,
                                      uint64_t ids_stride, const uint64_t *term,
                                      const uint64_t *read_ctx, const uint8_t *read_ctx_len,
                                      const uint8_t *bytes, uint64_t nbytes, const uint64_t *offsets,
                                      const uint32_t *groups, uint64_t n, oracle_msg *out_msgs,
                                      uint8_t *out_status, uint64_t *counts /* [10] */,
                                      const uint8_t *campaign, uint8_t *votes, uint64_t votes_stride) {
    wire_msgs(1, G, nslots, match, stride, active, acks, mask_bytes, ids, ids_stride, term, read_ctx, read_ctx_len,
              bytes, nbytes, offsets, groups, n, out_msgs, NULL, out_status, counts, campaign, votes, votes_stride);
}

/* The same decode and classification without stepping a Progress: the records (and, for
 * rejections, RejectHint as aux) that go on to the Progress stepper (oracle_step_progress).
 * Vote responses are polled here too (campaign != NULL). */
ORACLE_API void oracle_decode_wire_msgs(uint64_t G, int nslots, const uint64_t *ids, uint64_t ids_stride,
                                        const uint64_t *term, const uint64_t *read_ctx,
                                        const uint8_t *read_ctx_len, const uint8_t *bytes, uint64_t nbytes,
                                        const uint64_t *offsets, const uint32_t *groups, uint64_t n,
                                        oracle_msg *out_msgs, uint64_t *out_aux, uint8_t *out_status,
                                        uint64_t *counts /* [10] */, const uint8_t *campaign, uint8_t *votes,
                                        uint64_t votes_stride) {
    wire_msgs(0, G, nslots, NULL, 0, NULL, NULL, 1, ids, ids_stride, term, read_ctx, read_ctx_len, bytes, nbytes,
              offsets, groups, n, out_msgs, out_aux, out_status, counts, campaign, votes, votes_stride);
}

/* raftpb/raft.pb.go:1416-1418 — sovRaft: bytes of a varint. */
static int pb_sov(uint64_t x) {
    int n = 1;
    while (x >= 0x80) {
        x >>= 7;
        n++;
    }
    return n;
}

/* raftpb/raft.pb.go:1233-1243 — encodeVarintRaft: writes v so that it ends before `offset`. */
static int64_t pb_encode_varint(uint8_t *d, int64_t offset, uint64_t v) {
    offset -= pb_sov(v);
    int64_t base = offset;
    while (v >= 0x80) {
        d[offset++] = (uint8_t)((v & 0x7f) | 0x80);
        v >>= 7;
    }
    d[offset] = (uint8_t)v;
    return base;
}

/*
 * raftpb/raft.pb.go:1288-1325 (Size) and :937-1025 (MarshalToSizedBuffer), for a
 * Message without Entries, Snapshot and Responses: the buffer is filled from
 * its end, field 13 first.  ctx == NULL is a nil Context (field 12 omitted); a
 * non-nil empty one is written with length 0.  Returns the length, or -1 when
 * cap is too small.
 */
ORACLE_API int64_t oracle_message_marshal(int32_t type, uint64_t to, uint64_t from, uint64_t term,
                                          uint64_t log_term, uint64_t index, uint64_t commit, int reject,
                                          uint64_t reject_hint, const uint8_t *ctx, uint64_t ctx_len,
                                          uint64_t vote, uint8_t *out, uint64_t cap) {
    int64_t n = 0;
    n += 1 + pb_sov((uint64_t)(int64_t)type); /* uint64(m.Type): an int32 sign-extends */
    n += 1 + pb_sov(to);
    n += 1 + pb_sov(from);
    n += 1 + pb_sov(term);
    n += 1 + pb_sov(log_term);
    n += 1 + pb_sov(index);
    n += 1 + pb_sov(commit);
    n += 2;
    n += 1 + pb_sov(reject_hint);
    if (ctx != NULL) n += 1 + (int64_t)ctx_len + pb_sov(ctx_len);
    n += 1 + pb_sov(vote);
    if ((uint64_t)n > cap) return -1;
    uint8_t *d = out;
    int64_t i = n;
    i = pb_encode_varint(d, i, vote);
    d[--i] = 0x68;
    if (ctx != NULL) {
        i -= (int64_t)ctx_len;
        memcpy(d + i, ctx, ctx_len);
        i = pb_encode_varint(d, i, ctx_len);
        d[--i] = 0x62;
    }
    i = pb_encode_varint(d, i, reject_hint);
    d[--i] = 0x58;
    d[--i] = reject ? 1 : 0;
    d[--i] = 0x50;
    i = pb_encode_varint(d, i, commit);
    d[--i] = 0x40;
    i = pb_encode_varint(d, i, index);
    d[--i] = 0x30;
    i = pb_encode_varint(d, i, log_term);
    d[--i] = 0x28;
    i = pb_encode_varint(d, i, term);
    d[--i] = 0x20;
    i = pb_encode_varint(d, i, from);
    d[--i] = 0x18;
    i = pb_encode_varint(d, i, to);
    d[--i] = 0x10;
    i = pb_encode_varint(d, i, (uint64_t)(int64_t)type);
    d[--i] = 0x8;
    return i == 0 ? n : -2;
}

/*
 * raft.go:698-716 — bcastHeartbeat over G groups, one leader each: Visit every
 * Progress id but r.id, sendHeartbeat (raft.go:669-685): Commit = min(Match,
 * committed), Context = the last pending ReadIndex context or nil; r.send fills
 * From = r.id and Term = r.Term (raft.go:502-533).  The message of (slot s,
 * group g) is marshalled at out[(s*G + g) * msg_stride] (zero-padded), its
 * length in out_len[s*G + g]; 0 = none (empty slot, own slot, or no self bit:
 * the group is not led here).
 */
ORACLE_API void oracle_bcast_heartbeat(uint64_t G, int nslots, const uint64_t *match, uint64_t stride,
                                       const uint64_t *committed, const uint64_t *ids, uint64_t ids_stride,
                                       const uint64_t *term, const uint64_t *read_ctx,
                                       const uint8_t *read_ctx_len, const void *self_mask, int mask_bytes,
                                       uint8_t *out, uint64_t msg_stride, uint8_t *out_len) {
    memset(out, 0, (size_t)nslots * G * msg_stride);
    memset(out_len, 0, (size_t)nslots * G);
    for (uint64_t g = 0; g < G; g++) {
        uint64_t self = load_mask(self_mask, mask_bytes, g);
        if (self == 0) continue;
        int me = 0;
        while (!((self >> me) & 1)) me++;
        if (me >= nslots) continue;
        uint64_t my_id = ids[(uint64_t)me * ids_stride + g];
        if (my_id == 0) continue;
        uint8_t ctx[8];
        for (int b = 0; b < 8; b++) ctx[b] = (uint8_t)(read_ctx[g] >> (8 * b));
        for (int s = 0; s < nslots; s++) {
            uint64_t id = ids[(uint64_t)s * ids_stride + g];
            if (s == me || id == 0) continue;
            uint64_t m = match[(uint64_t)s * stride + g];
            uint64_t commit = m < committed[g] ? m : committed[g];
            int64_t len = oracle_message_marshal(8 /* MsgHeartbeat */, id, my_id, term[g], 0, 0, commit, 0, 0,
                                                 read_ctx_len[g] ? ctx : NULL, read_ctx_len[g], 0,
                                                 out + ((uint64_t)s * G + g) * msg_stride, msg_stride);
            out_len[(uint64_t)s * G + g] = (uint8_t)(len > 0 ? len : 0);
        }
    }
}

/*
 * raft.go:1231-1243 — MsgCheckQuorum over G groups: out = QuorumActive(), then
 * every Progress except the leader's own (self bit) gets RecentActive = false.
 */
ORACLE_API void oracle_batch_check_quorum(uint64_t G, int nslots, void *active, const void *in_mask,
                                          const void *out_mask, const void *learner_mask,
                                          int mask_bytes, int uniform_voters, const void *self_mask,
                                          uint8_t *out) {
    oracle_batch_quorum_active(G, nslots, active, in_mask, out_mask, learner_mask, mask_bytes,
                               uniform_voters, out);
    for (uint64_t g = 0; g < G; g++) {
        uint64_t a = load_mask(active, mask_bytes, g), self = load_mask(self_mask, mask_bytes, g);
        uint64_t keep = 0;
        for (int s = 0; s < nslots; s++) /* Visit: id != r.id -> RecentActive = false */
            if (((self >> s) & 1) && ((a >> s) & 1)) keep |= 1ULL << s;
        store_mask(active, mask_bytes, g, keep);
    }
}

/*
 * rawnode.go:64-66 — RawNode.Tick for G groups: r.tick() is tickHeartbeat for a leader (raft.go:835-862), tickElection
 * for every other state (raft.go:823-832).  Per group the fields of `raft` the two functions touch:
 * electionElapsed, heartbeatElapsed, randomizedElectionTimeout, state (StateType, raft.go:48-53), promotable()
 * (flag bit 0, raft.go:1883-1886) and leadTransferee != None (flag bit 1); electionTimeout, heartbeatTimeout and
 * checkQuorum are the same for all (raft.Config).  Step(MsgCheckQuorum) is restated from raft.go:1231-1243 with
 * becomeFollower -> reset (raft.go:864-871,760-789) as far as this state reaches: both timers 0, the transfer
 * aborted, every Progress fresh (RecentActive false).  action bits: 1 MsgHup stepped, 2 MsgCheckQuorum stepped,
 * 4 stepped down, 8 transfer aborted, 16 MsgBeat stepped.
 */
ORACLE_API void oracle_batch_tick(uint64_t G, int nslots, uint32_t *election_elapsed, uint32_t *heartbeat_elapsed,
                                  const uint32_t *randomized_timeout, uint8_t *state, uint8_t *flags,
                                  uint32_t election_timeout, uint32_t heartbeat_timeout, int check_quorum,
                                  void *active, const void *in_mask, const void *out_mask, const void *learner_mask,
                                  int mask_bytes, int uniform_voters, const void *self_mask, uint8_t *action) {
    enum { FOLLOWER = 0, LEADER = 2 };
    for (uint64_t g = 0; g < G; g++) {
        uint8_t act = 0;
        if (state[g] == LEADER) { /* tickHeartbeat */
            heartbeat_elapsed[g]++;
            election_elapsed[g]++;
            if (election_elapsed[g] >= election_timeout) {
                election_elapsed[g] = 0;
                if (check_quorum) {
                    act |= 2;
                    uint8_t qa = 0;
                    const char *a = (const char *)active + g * (uint64_t)mask_bytes;
                    oracle_batch_quorum_active(1, nslots, a, in_mask ? (const char *)in_mask + g * (uint64_t)mask_bytes : NULL,
                                               out_mask ? (const char *)out_mask + g * (uint64_t)mask_bytes : NULL,
                                               learner_mask ? (const char *)learner_mask + g * (uint64_t)mask_bytes : NULL,
                                               mask_bytes, uniform_voters, &qa);
                    if (!qa) { /* raft.go:1232-1235 */
                        act |= 4;
                        state[g] = FOLLOWER;
                        election_elapsed[g] = 0;
                        heartbeat_elapsed[g] = 0;
                        flags[g] &= (uint8_t)~2u; /* abortLeaderTransfer */
                        store_mask(active, mask_bytes, g, 0);
                    } else { /* raft.go:1236-1242 */
                        store_mask(active, mask_bytes, g, load_mask(active, mask_bytes, g) & load_mask(self_mask, mask_bytes, g));
                    }
                }
                if (state[g] == LEADER && (flags[g] & 2)) { /* raft.go:846-849 */
                    act |= 8;
                    flags[g] &= (uint8_t)~2u;
                }
            }
            if (state[g] == LEADER && heartbeat_elapsed[g] >= heartbeat_timeout) { /* raft.go:852-861 */
                heartbeat_elapsed[g] = 0;
                act |= 16;
            }
        } else { /* tickElection */
            election_elapsed[g]++;
            if ((flags[g] & 1) && election_elapsed[g] >= randomized_timeout[g]) {
                election_elapsed[g] = 0;
                act |= 1;
            }
        }
        action[g] = act;
    }
}

/* raft.go:669-685 — sendHeartbeat's commit := min(pr.Match, r.raftLog.committed). */
ORACLE_API void oracle_batch_heartbeat_commit(uint64_t G, int nslots, const uint64_t *match,
                                              uint64_t stride, const uint64_t *committed,
                                              uint64_t *out, uint64_t out_stride) {
    for (int s = 0; s < nslots; s++)
        for (uint64_t g = 0; g < G; g++) {
            uint64_t m = match[(uint64_t)s * stride + g];
            out[(uint64_t)s * out_stride + g] = m < committed[g] ? m : committed[g];
        }
}

/* ------------------------------------------------------------------------ */
/* CPU baseline (bench.py cpu_baseline leg, kind "port"): the reference's    */
/* data shape — one id set + one id-keyed map per group, built up front —    */
/* and a goroutine-per-core style parallel loop over groups calling          */
/* CommittedIndex (quorum/bench_test.go:24-40 run across many groups).       */
/* ------------------------------------------------------------------------ */
typedef struct {
    uint64_t G;
    int V;
    uint64_t *cfg; /* [G][V] voter ids */
    idmap *l;      /* [G] */
} oracle_bench;

ORACLE_API void *oracle_bench_create(uint64_t G, int V, const uint64_t *match, uint64_t stride) {
    oracle_bench *b = (oracle_bench *)calloc(1, sizeof(*b));
    b->G = G;
    b->V = V;
    b->cfg = (uint64_t *)malloc(sizeof(uint64_t) * G * (size_t)V);
    b->l = (idmap *)malloc(sizeof(idmap) * G);
    for (uint64_t g = 0; g < G; g++) {
        idmap_init(&b->l[g], (uint32_t)V);
        for (int s = 0; s < V; s++) {
            b->cfg[g * (uint64_t)V + (uint64_t)s] = (uint64_t)s + 1; /* ids 1..V, bench_test.go:30-31 */
            idmap_put(&b->l[g], (uint64_t)s + 1, match[(uint64_t)s * stride + g]);
        }
    }
    return b;
}

ORACLE_API int oracle_bench_run(void *h, uint64_t *out, int threads) {
    oracle_bench *b = (oracle_bench *)h;
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(static)
        for (int64_t g = 0; g < (int64_t)b->G; g++)
            out[g] = majority_committed_index(&b->cfg[(uint64_t)g * (uint64_t)b->V], b->V, &b->l[g]);
    }
#else
    (void)threads;
    for (uint64_t g = 0; g < b->G; g++)
        out[g] = majority_committed_index(&b->cfg[g * (uint64_t)b->V], b->V, &b->l[g]);
#endif
    return used;
}

/*
 * quorum/bench_test.go:24-40 as written: ONE MajorityConfig of V voters (ids
 * 1..V) and one mapAckIndexer with rand.Int63n(MaxInt64)-style indexes,
 * CommittedIndex called `iters` times on one thread.  Returns ns per op
 * (BASELINE cfg1).  `sink` keeps the loop from being optimised away.
 */
ORACLE_API double oracle_bench_leaf_ns(int V, uint64_t iters, uint64_t seed, uint64_t *sink) {
    uint64_t cfg[64];
    idmap l;
    idmap_init(&l, (uint32_t)V);
    uint64_t x = seed;
    for (int s = 0; s < V; s++) {
        cfg[s] = (uint64_t)s + 1;
        x += 0x9E3779B97F4A7C15ULL;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        idmap_put(&l, cfg[s], ((z ^ (z >> 31)) >> 1) % 0x7FFFFFFFFFFFFFFFULL);
    }
    struct timespec t0, t1;
    uint64_t acc = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint64_t i = 0; i < iters; i++) {
        acc += majority_committed_index(cfg, V, &l);
        __asm__ volatile("" : "+r"(acc) : : "memory");
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    idmap_free(&l);
    *sink = acc;
    return ((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / (double)iters;
}

/*
 * "Best CPU" figure of SURVEY.md §8(d)(ii): NOT the reference's shape — the
 * engine's own [slot][group] planes and a branch-free median network on the
 * host cores — reported beside the port so the GPU/CPU ratio is not flattered
 * by the map-shaped loop.  V = 5 only (the headline shape).
 */
ORACLE_API int oracle_bench_soa5(uint64_t G, const uint64_t *match, uint64_t stride, uint64_t *out,
                                 int threads) {
    int used = 1;
#define ORACLE_CE(a, b) do { uint64_t lo_ = a < b ? a : b, hi_ = a < b ? b : a; a = lo_; b = hi_; } while (0)
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for schedule(static)
#else
    (void)threads;
    {
#endif
        for (int64_t g = 0; g < (int64_t)G; g++) {
            uint64_t a = match[g], b = match[stride + g], c = match[2 * stride + g],
                     d = match[3 * stride + g], e = match[4 * stride + g];
            ORACLE_CE(a, d); ORACLE_CE(b, e); ORACLE_CE(a, c); ORACLE_CE(b, d); ORACLE_CE(a, b);
            ORACLE_CE(c, e); ORACLE_CE(b, c); ORACLE_CE(d, e); ORACLE_CE(c, d);
            out[g] = c; /* srt[5-(5/2+1)] = srt[2] */
        }
    }
#undef ORACLE_CE
    return used;
}

ORACLE_API void oracle_bench_destroy(void *h) {
    oracle_bench *b = (oracle_bench *)h;
    for (uint64_t g = 0; g < b->G; g++) idmap_free(&b->l[g]);
    free(b->l);
    free(b->cfg);
    free(b);
}
