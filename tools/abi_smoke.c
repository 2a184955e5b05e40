/* abi_smoke.c — every entry point of include/rafthip.h called once from plain C
 * with exactly-sized heap buffers, over a partial group range where the entry
 * point takes one.  Built with AddressSanitizer by tools/asan_host.sh: a
 * staging copy or download that runs past a caller's buffer aborts here.  It
 * checks return codes and a few invariants, not parity (tests/ does that).
 *
 *   cc -std=c11 -Iinclude tools/abi_smoke.c -Lraft_amd/lib -lrafthip -o abi_smoke
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rafthip.h"

#define G 3001u
#define S 9u
#define FIRST 130u
#define COUNT 2049u
#define CAP 3u

static int failures;
#define OK(call)                                                                         \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != 0) {                                                                  \
            fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_,     \
                    rafthip_last_error());                                               \
            failures++;                                                                  \
        }                                                                                \
    } while (0)
#define FAILS(call)                                                                      \
    do {                                                                                 \
        if ((call) == 0) {                                                               \
            fprintf(stderr, "%s:%d %s succeeded, want an error\n", __FILE__, __LINE__, #call); \
            failures++;                                                                  \
        }                                                                                \
    } while (0)

static uint64_t rng_state = 0x5EED1234ull;
static uint64_t rnd(void) { /* splitmix64 */
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* The stateless call at a size that takes its pipelined path (staging threads, result thread), from two threads at
 * once, with heap buffers sized to the byte: match[s][g] = base + g + s, so 5 voters commit base + g + 2. */
#define SHOT_G 300001u
static void *shot_worker(void *arg) {
    const uint64_t base = (uint64_t)(uintptr_t)arg;
    uint64_t *m = malloc((size_t)5 * SHOT_G * 8), *res = malloc((size_t)SHOT_G * 8);
    uint8_t *in = malloc(SHOT_G), *votes = malloc((size_t)5 * SHOT_G), *vr = malloc(SHOT_G);
    for (uint32_t s = 0; s < 5; s++)
        for (uint32_t g = 0; g < SHOT_G; g++) {
            m[(size_t)s * SHOT_G + g] = base + g + s;
            votes[(size_t)s * SHOT_G + g] = s < 3 ? 2 : 1; /* granted : rejected */
        }
    memset(in, 0x1F, SHOT_G);
    long bad = 0;
    for (int rep = 0; rep < 3; rep++) {
        memset(res, 0, (size_t)SHOT_G * 8);
        if (rafthip_batch_committed_index(0, SHOT_G, 5, m, in, NULL, res) != 0) bad++;
        for (uint32_t g = 0; g < SHOT_G; g++) bad += res[g] != base + g + 2;
        if (rafthip_batch_vote_result(0, SHOT_G, 5, votes, in, NULL, vr) != 0) bad++;
        for (uint32_t g = 0; g < SHOT_G; g++) bad += vr[g] != RAFTHIP_VOTE_WON;
    }
    free(m), free(res), free(in), free(votes), free(vr);
    return (void *)(intptr_t)bad;
}

static void *buf(size_t n) { /* exactly n bytes, filled with noise */
    unsigned char *p = malloc(n ? n : 1);
    for (size_t i = 0; i < n; i++) p[i] = (unsigned char)rnd();
    return p;
}
static size_t put_varint(uint8_t *d, uint64_t v) {
    size_t n = 0;
    while (v >= 0x80) d[n++] = (uint8_t)(v | 0x80), v >>= 7;
    d[n++] = (uint8_t)v;
    return n;
}
static size_t put_msg(uint8_t *d, uint32_t type, uint64_t from, uint64_t term, uint64_t index, int reject) {
    size_t n = 0;
    const uint64_t f[6] = {type, 1, from, term, reject ? 3 : 0, index};
    for (int k = 0; k < 6; k++) d[n++] = (uint8_t)((k + 1) << 3), n += put_varint(d + n, f[k]);
    d[n++] = 0x40, d[n++] = 0;
    d[n++] = 0x50, d[n++] = (uint8_t)(reject != 0);
    d[n++] = 0x58, n += put_varint(d + n, index ? index - 1 : 0);
    d[n++] = 0x68, d[n++] = 0;
    return n;
}

int main(void) {
    int ndev = 0;
    OK(rafthip_device_count(&ndev));
    if (rafthip_abi_version() != RAFTHIP_ABI_VERSION || ndev < 1) {
        fprintf(stderr, "abi %u, %d devices\n", rafthip_abi_version(), ndev);
        return 2;
    }
    rafthip_config cfg = {sizeof cfg, 0, G, S, 0, NULL};
    rafthip_engine *e = NULL;
    OK(rafthip_create(&cfg, &e));
    if (!e) return 2;
    const int mb = rafthip_mask_bytes(e);
    if (mb != 2) failures++;
    FAILS(rafthip_upload_match(e, G - 1, 2, NULL, 2)); /* range past the end */
    FAILS(rafthip_create(NULL, &e));

    /* configuration and the planes, partial range, exactly-sized sources */
    uint16_t *in = buf(COUNT * 2), *out = buf(COUNT * 2), *lrn = buf(COUNT * 2);
    for (unsigned g = 0; g < COUNT; g++) in[g] &= 0x7F, out[g] = g % 3 ? 0 : (out[g] & 0xFE), lrn[g] = 0x180 & ~(in[g] | out[g]);
    OK(rafthip_upload_config(e, FIRST, COUNT, in, out, lrn));
    uint64_t *match = buf((size_t)S * COUNT * 8);
    OK(rafthip_upload_match(e, FIRST, COUNT, match, COUNT));
    OK(rafthip_read_match(e, FIRST, COUNT, match, COUNT));
    OK(rafthip_reset_match(e, 0, FIRST));
    uint8_t *votes = buf((size_t)S * COUNT);
    for (size_t i = 0; i < (size_t)S * COUNT; i++) votes[i] %= 3;
    OK(rafthip_upload_votes(e, FIRST, COUNT, votes, COUNT));
    OK(rafthip_reset_votes(e, FIRST, 7));
    OK(rafthip_read_votes(e, FIRST, COUNT, votes, COUNT));
    uint16_t *bits = buf(COUNT * 2);
    OK(rafthip_upload_active(e, FIRST, COUNT, bits));
    OK(rafthip_upload_acks(e, FIRST, COUNT, bits));
    for (unsigned g = 0; g < COUNT; g++) bits[g] = (uint16_t)(g % 11 ? 1u << (g % S) : 0);
    OK(rafthip_upload_self(e, FIRST, COUNT, bits));
    OK(rafthip_read_active(e, FIRST, COUNT, bits));
    OK(rafthip_read_acks(e, FIRST, COUNT, bits));
    uint64_t *tf = buf(COUNT * 8), *tl = buf(COUNT * 8), *com = buf(COUNT * 8);
    for (unsigned g = 0; g < COUNT; g++) tf[g] = 0, tl[g] = ~0ull, com[g] >>= 1;
    OK(rafthip_upload_term_range(e, FIRST, COUNT, tf, tl));
    OK(rafthip_read_term_range(e, FIRST, COUNT, tf, tl));
    OK(rafthip_upload_commit(e, FIRST, COUNT, com));

    /* record streams */
    const size_t n = 5000;
    rafthip_ack *acks = buf(n * sizeof *acks);
    rafthip_msg *msgs = buf(n * sizeof *msgs);
    uint64_t *aux = buf(n * 8);
    uint8_t *act = buf(n);
    for (size_t i = 0; i < n; i++) {
        acks[i].group = (uint32_t)(rnd() % (G + 3)), acks[i].slot = (uint32_t)(rnd() % (S + 1));
        msgs[i].group = (uint32_t)(rnd() % (G + 3)), msgs[i].slot = (uint8_t)(rnd() % (S + 1));
        const uint8_t types[8] = {4, 9, 10, 11, 0xF0, 0xF1, 0xF2, 6};
        msgs[i].type = types[rnd() % 8], msgs[i].reject &= 1, msgs[i].flags &= 1, msgs[i].index >>= 40, aux[i] >>= 44;
    }
    OK(rafthip_scatter_acks(e, acks, n));
    OK(rafthip_ingest_msgs(e, msgs, n));
    OK(rafthip_record_votes(e, msgs, n));
    uint64_t *plane = buf((size_t)S * COUNT * 8);
    OK(rafthip_merge_acks(e, FIRST, COUNT, plane, COUNT));

    /* per-peer replication state */
    OK(rafthip_progress_config(e, CAP, 100));
    FAILS(rafthip_progress_config(e, CAP + 1, 100)); /* fixed once configured */
    uint8_t *state = buf((size_t)S * COUNT);
    uint64_t *next = buf((size_t)S * COUNT * 8), *pend = buf((size_t)S * COUNT * 8), *ib = buf((size_t)S * COUNT * 8);
    uint32_t *im = buf((size_t)S * COUNT * 4);
    uint64_t *ri = buf((size_t)CAP * S * COUNT * 8), *rb = buf((size_t)CAP * S * COUNT * 8);
    for (size_t i = 0; i < (size_t)S * COUNT; i++) state[i] = (uint8_t)((state[i] % 3) | (state[i] & 4)), im[i] = 0, ib[i] = 0;
    OK(rafthip_upload_progress(e, FIRST, COUNT, state, next, pend, COUNT));
    OK(rafthip_upload_inflights(e, FIRST, COUNT, im, ib, ri, rb, COUNT));
    OK(rafthip_init_progress_row(e, FIRST, S - 1, RAFTHIP_STATE_PROBE, 7, 0));
    FAILS(rafthip_init_progress_row(e, FIRST, S, RAFTHIP_STATE_PROBE, 7, 0)); /* slot past the end */
    OK(rafthip_upload_first_index(e, FIRST, COUNT, com));
    OK(rafthip_step_progress(e, msgs, aux, n, act));
    OK(rafthip_step_progress(e, msgs, NULL, n, NULL));
    uint8_t *dact = buf((size_t)S * COUNT);
    OK(rafthip_step_dense(e, FIRST, COUNT, plane, COUNT, bits, bits, dact, COUNT));
    uint8_t *plan = buf((size_t)S * COUNT);
    OK(rafthip_plan_appends(e, FIRST, COUNT, 1, plan, next, COUNT));
    OK(rafthip_plan_appends(e, FIRST, COUNT, 0, plan, NULL, COUNT));
    for (size_t i = 0; i < (size_t)S * COUNT; i++)
        if (plan[i] > RAFTHIP_SEND_SNAPSHOT) failures++;
    OK(rafthip_read_progress(e, FIRST, COUNT, state, next, pend, im, ib, COUNT));
    OK(rafthip_read_inflights(e, FIRST, COUNT, ri, rb, COUNT));

    /* the wire path */
    uint64_t *ids = buf((size_t)S * COUNT * 8), *term = buf(COUNT * 8), *ctx = buf(COUNT * 8);
    uint8_t *ctx_len = buf(COUNT), *camp = buf(COUNT);
    for (size_t i = 0; i < (size_t)S * COUNT; i++) ids[i] = 1 + i / COUNT;
    for (unsigned g = 0; g < COUNT; g++) term[g] = 5, ctx_len[g] %= 9, camp[g] = (uint8_t)(g % 3 == 0 ? 6 : g % 3 == 1 ? 18 : 0);
    FAILS(rafthip_ingest_wire(e, (const uint8_t *)"", (const uint64_t *)tf, (const uint32_t *)tf, 1, NULL, NULL, NULL)); /* ids first */
    OK(rafthip_upload_ids(e, FIRST, COUNT, ids, COUNT));
    OK(rafthip_upload_term(e, FIRST, COUNT, term));
    OK(rafthip_upload_read_ctx(e, FIRST, COUNT, ctx, ctx_len));
    OK(rafthip_upload_campaign(e, FIRST, COUNT, camp));
    uint8_t *wire = malloc(n * 64);
    uint64_t *off = buf((n + 1) * 8);
    uint32_t *grp = buf(n * 4);
    size_t nb = 0;
    for (size_t i = 0; i < n; i++) {
        const uint32_t types[4] = {4, 9, 6, 18};
        off[i] = nb;
        grp[i] = (uint32_t)(FIRST + rnd() % COUNT);
        nb += put_msg(wire + nb, types[rnd() % 4], 1 + rnd() % (S + 1), 4 + rnd() % 3, rnd() >> 44, (int)(rnd() & 1));
    }
    off[n] = nb;
    uint8_t *exact = malloc(nb ? nb : 1); /* the byte stream, sized to the byte */
    memcpy(exact, wire, nb);
    rafthip_msg *omsgs = buf(n * sizeof *omsgs);
    uint8_t *ostat = buf(n);
    uint64_t counts[RAFTHIP_WIRE_STATUSES], sum = 0;
    OK(rafthip_ingest_wire(e, exact, off, grp, n, omsgs, ostat, counts));
    for (int k = 0; k < RAFTHIP_WIRE_STATUSES; k++) sum += counts[k];
    if (sum != n || counts[RAFTHIP_WIRE_VOTE] == 0 || counts[RAFTHIP_WIRE_APPLIED] == 0) {
        fprintf(stderr, "wire counts: sum %llu votes %llu applied %llu\n", (unsigned long long)sum,
                (unsigned long long)counts[RAFTHIP_WIRE_VOTE], (unsigned long long)counts[RAFTHIP_WIRE_APPLIED]);
        failures++;
    }
    OK(rafthip_step_wire(e, exact, off, grp, n, omsgs, ostat, counts, act));
    OK(rafthip_step_wire(e, exact, off, grp, n, NULL, NULL, NULL, NULL));
    OK(rafthip_upload_campaign(e, 0, G, NULL));
    uint8_t *hb = buf((size_t)S * COUNT * RAFTHIP_HEARTBEAT_STRIDE), *hlen = buf((size_t)S * COUNT);
    OK(rafthip_emit_heartbeats(e, FIRST, COUNT, hb, hlen));
    for (size_t i = 0; i < (size_t)S * COUNT; i++)
        if (hlen[i] > RAFTHIP_HEARTBEAT_STRIDE || (hlen[i] && hb[i * RAFTHIP_HEARTBEAT_STRIDE] != 0x08)) failures++;

    /* quorum arithmetic and its read-backs */
    rafthip_commit_pair *pairs = buf(100 * sizeof *pairs); /* fewer than can change: the copy is clipped */
    uint64_t np = 0;
    OK(rafthip_commit(e));
    OK(rafthip_commit_many(&e, 1));
    OK(rafthip_read_commit(e, FIRST, COUNT, com));
    OK(rafthip_commit_changed(e, pairs, 100, &np));
    OK(rafthip_upload_commit(e, FIRST, COUNT, tf));
    OK(rafthip_commit_advance(e, pairs, 100, &np));
    OK(rafthip_commit_advance(e, NULL, 0, &np));
    {   /* the several-engines form, here with one engine (its config is per-group: served engine by engine) */
        rafthip_commit_pair *dsts[1];
        uint64_t mp[1] = {100}, cnt[1] = {0};
        dsts[0] = pairs;
        OK(rafthip_commit_report_many(&e, 1, 0, dsts, mp, cnt));
        OK(rafthip_commit_report_many(&e, 1, 1, NULL, NULL, cnt));
        FAILS(rafthip_commit_report_many(&e, 1, 0, NULL, mp, cnt)); /* room without a buffer */
    }
    {   /* the allocation-free variant writes its pairs itself: page-locked landing buffer, sized to the byte */
        void *pin = NULL;
        OK(rafthip_host_alloc(&pin, 100 * sizeof *pairs));
        OK(rafthip_commit_changed_static(e, pin, 100, &np));
        OK(rafthip_host_free(pin));
    }
    uint8_t *r8 = buf(COUNT), *g8 = buf(COUNT), *j8 = buf(COUNT);
    OK(rafthip_committed_in_term(e));
    OK(rafthip_read_committed_in_term(e, FIRST, COUNT, r8));
    OK(rafthip_heartbeat_commit(e, FIRST, COUNT, match, COUNT));
    OK(rafthip_tally_votes(e));
    OK(rafthip_read_vote_result(e, FIRST, COUNT, r8, g8, j8));
    OK(rafthip_read_vote_result(e, FIRST, COUNT, r8, NULL, NULL));
    OK(rafthip_quorum_active(e));
    OK(rafthip_read_quorum_active(e, FIRST, COUNT, r8));
    OK(rafthip_check_quorum(e));
    OK(rafthip_ack_quorum(e));
    OK(rafthip_read_ack_quorum(e, FIRST, COUNT, r8));
    /* RawNode.Tick for every group */
    FAILS(rafthip_tick(e, NULL));                 /* rafthip_tick_config first */
    FAILS(rafthip_tick_config(e, 10, 0, 1));      /* Config.validate, raft.go:296-302 */
    FAILS(rafthip_tick_config(e, 2, 2, 1));
    OK(rafthip_tick_config(e, 10, 1, 1));
    {
        uint32_t *el = buf(COUNT * 4), *hb = buf(COUNT * 4), *rt = buf(COUNT * 4);
        uint8_t *role = buf(COUNT), *fl = buf(COUNT), *act = buf(COUNT);
        uint64_t counts[RAFTHIP_TICK_ACTIONS], seen[RAFTHIP_TICK_ACTIONS] = {0};
        for (unsigned g = 0; g < COUNT; g++) el[g] %= 20, hb[g] = 0, rt[g] = 10 + rt[g] % 10, role[g] &= 3, fl[g] &= 3;
        OK(rafthip_upload_tick_state(e, FIRST, COUNT, el, hb, rt, role, fl));
        OK(rafthip_upload_tick_state(e, FIRST, COUNT, NULL, NULL, NULL, role, NULL));
        OK(rafthip_tick(e, counts));
        {
            uint32_t *heard = buf(777 * 4);
            uint64_t bad = 0, want_bad = 0;
            for (int i = 0; i < 777; i++) heard[i] %= G + 50, want_bad += heard[i] >= G;
            OK(rafthip_tick_heard_from_leader(e, heard, 777, &bad));
            OK(rafthip_tick_heard_from_leader(e, heard, 777, NULL));
            OK(rafthip_tick_heard_from_leader(e, NULL, 0, &bad) || bad != 0);
            FAILS(rafthip_tick_heard_from_leader(e, NULL, 5, &bad));
            OK(rafthip_tick_heard_from_leader(e, heard, 777, &bad));
            if (bad != want_bad) { fprintf(stderr, "heard: %llu rejected, want %llu\n", (unsigned long long)bad, (unsigned long long)want_bad); return 1; }
        }
        OK(rafthip_tick(e, NULL));
        {
            rafthip_engine *two[2] = {e, e};
            uint64_t c2[2 * RAFTHIP_TICK_ACTIONS];
            OK(rafthip_tick_many(two, 1, c2));
            OK(rafthip_tick_many(two, 1, NULL));
            FAILS(rafthip_tick_many(two, 0, NULL));
            FAILS(rafthip_tick_many(two, 2, NULL)); /* the same engine twice */
            FAILS(rafthip_tick_many(NULL, 1, NULL));
        }
        OK(rafthip_tick(e, counts));
        OK(rafthip_read_tick_actions(e, 0, G, buf(G)));
        OK(rafthip_read_tick_actions(e, FIRST, COUNT, act));
        for (unsigned g = 0; g < COUNT; g++)
            for (int b = 0; b < RAFTHIP_TICK_ACTIONS; b++) seen[b] += (act[g] >> b) & 1;
        for (int b = 0; b < RAFTHIP_TICK_ACTIONS; b++)
            if (seen[b] != counts[b]) { fprintf(stderr, "tick count %d: %llu != %llu\n", b, (unsigned long long)counts[b], (unsigned long long)seen[b]); return 1; }
        {
            rafthip_tick_event *ev = buf(G * sizeof(rafthip_tick_event));
            uint64_t n = 0, want = 0;
            uint8_t *all = buf(G);
            OK(rafthip_read_tick_actions(e, 0, G, all));
            for (unsigned g = 0; g < G; g++) want += (all[g] & (RAFTHIP_TICK_HUP | RAFTHIP_TICK_STEP_DOWN)) != 0;
            OK(rafthip_read_tick_events(e, RAFTHIP_TICK_HUP | RAFTHIP_TICK_STEP_DOWN, ev, G, &n));
            if (n != want) { fprintf(stderr, "tick events: %llu != %llu\n", (unsigned long long)n, (unsigned long long)want); return 1; }
            for (uint64_t i = 0; i < n; i++)
                if (ev[i].group >= G || ev[i].action != all[ev[i].group] || (i && ev[i].group <= ev[i - 1].group)) { fprintf(stderr, "tick event %llu is wrong\n", (unsigned long long)i); return 1; }
            OK(rafthip_read_tick_events(e, 0xFF, NULL, 0, &n));
            FAILS(rafthip_read_tick_events(e, 0xFF, NULL, 4, &n));
        }
        OK(rafthip_read_tick_state(e, FIRST, COUNT, el, hb, rt, role, fl));
        OK(rafthip_read_tick_state(e, FIRST, COUNT, NULL, NULL, NULL, role, NULL));
        {
            uint8_t *hb2 = buf((size_t)S * COUNT * RAFTHIP_HEARTBEAT_STRIDE), *hlen2 = buf((size_t)S * COUNT);
            OK(rafthip_emit_heartbeats_due(e, FIRST, COUNT, hb2, hlen2));
            for (size_t i = 0; i < (size_t)S * COUNT; i++)
                if (hlen2[i] && !(act[i % COUNT] & RAFTHIP_TICK_BEAT)) { fprintf(stderr, "heartbeat from a group that did not beat\n"); return 1; }
        }
        FAILS(rafthip_read_tick_actions(e, FIRST, G, act));
    }
    OK(rafthip_set_uniform_config(e, 7));
    OK(rafthip_commit(e));
    OK(rafthip_tally_votes(e));
    void *dev = NULL;
    uint64_t stride = 0;
    OK(rafthip_plane(e, 0, &dev, &stride));
    if (!dev || stride < G) failures++;
    rafthip_stats_t st;
    OK(rafthip_stats(e, &st));
    OK(rafthip_synchronize(e));

    /* one-shot forms and page-locked buffers */
    uint64_t *one = buf(COUNT * 8);
    OK(rafthip_batch_committed_index(0, COUNT, S, match, in, out, one));
    OK(rafthip_batch_vote_result(0, COUNT, S, votes, in, out, r8));
    {
        pthread_t th[2];
        void *bad = NULL;
        for (int t = 0; t < 2; t++) pthread_create(&th[t], NULL, shot_worker, (void *)(uintptr_t)(1000u * (t + 1)));
        for (int t = 0; t < 2; t++) {
            pthread_join(th[t], &bad);
            if (bad) {
                fprintf(stderr, "stateless call from thread %d: %ld wrong results or errors\n", t, (long)(intptr_t)bad);
                failures++;
            }
        }
    }
    void *pin = NULL;
    OK(rafthip_host_alloc(&pin, n * sizeof *acks));
    if (pin) {
        memcpy(pin, acks, n * sizeof *acks);
        OK(rafthip_scatter_acks(e, pin, n));
        OK(rafthip_host_free(pin));
    }
    {   /* the exchange step on a one-rank communicator: per-rank form, then the one-thread-many-GPUs form */
        rafthip_comm_id id;
        rafthip_comm *c = NULL, *all[1] = {NULL};
        uint64_t cnt[1] = {0}, tot = 0, cap = 0, base[1] = {1000000}, mp[1] = {100};
        int r = -1, nr = -1;
        OK(rafthip_comm_unique_id(&id));
        OK(rafthip_comm_create(e, &id, 0, 1, 1000000, 0, &c));
        OK(rafthip_comm_info(c, &r, &nr, &cap));
        if (r != 0 || nr != 1 || cap < G) failures++;
        OK(rafthip_commit_allgather(c, 0, pairs, 100, cnt, &tot)); /* host dst, clipped to 100 pairs */
        OK(rafthip_commit_allgather(c, 1, NULL, 0, NULL, &tot));
        {   /* the dense form: the width first (dst NULL), then a heap buffer sized to the element */
            uint64_t width = 0, w2 = 0;
            OK(rafthip_commit_allgather_dense(c, 0, NULL, 0, &width));
            if (width < G) failures++;
            uint64_t *plane = malloc(width * sizeof *plane);
            OK(rafthip_commit_allgather_dense(c, 0, plane, width, &w2));
            if (w2 != width) failures++;
            FAILS(rafthip_commit_allgather_dense(c, 0, plane, width - 1, NULL)); /* too small */
            OK(rafthip_commit_allgather_dense(c, 1, plane, width, NULL));
            free(plane);
        }
        OK(rafthip_comm_destroy(c));
        OK(rafthip_comm_create_all(&e, 1, base, 0, all));
        rafthip_commit_pair *dsts[1] = {pairs};
        uint64_t *cnts[1] = {cnt};
        OK(rafthip_commit_allgather_many(all, 1, 0, dsts, mp, cnts, &tot));
        {
            uint64_t width = 0;
            OK(rafthip_commit_allgather_dense_many(all, 1, 0, NULL, NULL, &width));
            uint64_t *plane = malloc(width * sizeof *plane), *planes[1];
            planes[0] = plane;
            OK(rafthip_commit_allgather_dense_many(all, 1, 0, planes, &width, NULL));
            free(plane);
        }
        OK(rafthip_comm_destroy(all[0]));
        FAILS(rafthip_comm_create(e, &id, 1, 1, 0, 0, &c)); /* rank outside [0, nranks) */
    }
    OK(rafthip_destroy(e));
    if (failures) {
        fprintf(stderr, "abi_smoke: %d failure(s)\n", failures);
        return 1;
    }
    printf("abi_smoke ok: %llu groups evaluated, %llu messages ingested\n", (unsigned long long)st.groups_evaluated,
           (unsigned long long)st.msgs_ingested);
    return 0;
}
