/*
 * progress_oracle.c — CPU restatement of the leader's per-peer replication
 * state: tracker.Progress, tracker.Inflights and the stepLeader branches that
 * drive them.
 *
 * TEST INFRASTRUCTURE ONLY (see quorum_oracle.c's header): the parity checker
 * for rafthip_step_progress.  Nothing under raft_amd/ may call into it.
 *
 * Every function cites the reference file:line it follows.  Pinned by the
 * reference's own tables and scenarios, re-encoded under tests/golden/progress.json
 * (tracker/progress_test.go, tracker/inflights_test.go, raft_snap_test.go,
 * raft_flow_control_test.go, raft_test.go, testdata/ *.txt): tests/test_progress_cpu.py.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* tracker/state.go:18-34 */
enum { STATE_PROBE = 0, STATE_REPLICATE = 1, STATE_SNAPSHOT = 2 };

/* tracker/inflights.go:18-40 */
typedef struct {
    uint64_t index, bytes;
} inflight;
typedef struct {
    int start, count;
    uint64_t bytes;
    int size;
    uint64_t max_bytes;
    inflight *buffer; /* len(buffer) == size here: the reference grows it lazily (:85-95),
                         which moves no entry — positions are start+i mod size either way */
} inflights;

/* tracker/progress.go:30-98 */
typedef struct {
    uint64_t match, next;
    int state;
    uint64_t pending_snapshot;
    int recent_active, msg_app_flow_paused;
    inflights ins;
} progress;

/* tracker/inflights.go:131-133 */
static int inflights_full(const inflights *in) {
    return in->count == in->size || (in->max_bytes != 0 && in->bytes >= in->max_bytes);
}

/* tracker/inflights.go:65-80; returns -1 where the reference panics */
static int inflights_add(inflights *in, uint64_t index, uint64_t bytes) {
    if (inflights_full(in)) return -1;
    int next = in->start + in->count;
    int size = in->size;
    if (next >= size) next -= size;
    in->buffer[next].index = index;
    in->buffer[next].bytes = bytes;
    in->count++;
    in->bytes += bytes;
    return 0;
}

/* tracker/inflights.go:98-128 */
static void inflights_free_le(inflights *in, uint64_t to) {
    if (in->count == 0 || to < in->buffer[in->start].index) return;
    int idx = in->start;
    int i;
    uint64_t bytes = 0;
    for (i = 0; i < in->count; i++) {
        if (to < in->buffer[idx].index) break;
        bytes += in->buffer[idx].bytes;
        int size = in->size;
        if (++idx >= size) idx -= size;
    }
    in->count -= i;
    in->bytes -= bytes;
    in->start = idx;
    if (in->count == 0) in->start = 0;
}

/* tracker/inflights.go:139-143 */
static void inflights_reset(inflights *in) {
    in->start = 0;
    in->count = 0;
    in->bytes = 0;
}

static uint64_t max64(uint64_t a, uint64_t b) { return a > b ? a : b; }
static uint64_t min64(uint64_t a, uint64_t b) { return a < b ? a : b; }

/* tracker/progress.go:102-107 */
static void reset_state(progress *pr, int state) {
    pr->msg_app_flow_paused = 0;
    pr->pending_snapshot = 0;
    pr->state = state;
    inflights_reset(&pr->ins);
}

/* tracker/progress.go:111-123 */
static void become_probe(progress *pr) {
    if (pr->state == STATE_SNAPSHOT) {
        uint64_t pending = pr->pending_snapshot;
        reset_state(pr, STATE_PROBE);
        pr->next = max64(pr->match + 1, pending + 1);
    } else {
        reset_state(pr, STATE_PROBE);
        pr->next = pr->match + 1;
    }
}

/* tracker/progress.go:126-129 */
static void become_replicate(progress *pr) {
    reset_state(pr, STATE_REPLICATE);
    pr->next = pr->match + 1;
}

/* tracker/progress.go:133-136 */
static void become_snapshot(progress *pr, uint64_t snapshoti) {
    reset_state(pr, STATE_SNAPSHOT);
    pr->pending_snapshot = snapshoti;
}

/* tracker/progress.go:141-163; `last` = nextIndex + entries - 1.  Returns -1 for
 * the error return (:160) and for the Add panic (inflights.go:66-68). */
static int update_on_entries_send(progress *pr, int has_entries, uint64_t bytes, uint64_t last) {
    switch (pr->state) {
    case STATE_REPLICATE:
        if (has_entries) {
            if (inflights_full(&pr->ins)) return -1;
            pr->next = last + 1; /* OptimisticUpdate, :181 */
            inflights_add(&pr->ins, last, bytes);
        }
        pr->msg_app_flow_paused = inflights_full(&pr->ins);
        return 0;
    case STATE_PROBE:
        if (has_entries) pr->msg_app_flow_paused = 1;
        return 0;
    default:
        return -1;
    }
}

/* tracker/progress.go:168-177 */
static int maybe_update(progress *pr, uint64_t n) {
    int updated = 0;
    if (pr->match < n) {
        pr->match = n;
        updated = 1;
        pr->msg_app_flow_paused = 0;
    }
    pr->next = max64(pr->next, n + 1);
    return updated;
}

/* tracker/progress.go:194-217 */
static int maybe_decr_to(progress *pr, uint64_t rejected, uint64_t match_hint) {
    if (pr->state == STATE_REPLICATE) {
        if (rejected <= pr->match) return 0;
        pr->next = pr->match + 1;
        return 1;
    }
    if (pr->next - 1 != rejected) return 0;
    pr->next = max64(min64(rejected, match_hint + 1), 1);
    pr->msg_app_flow_paused = 0;
    return 1;
}

/* tracker/progress.go:225-236 */
static int is_paused(const progress *pr) {
    switch (pr->state) {
    case STATE_PROBE: return pr->msg_app_flow_paused;
    case STATE_REPLICATE: return pr->msg_app_flow_paused;
    default: return 1;
    }
}

/* ---- a Progress as an opaque handle: the unit tests of the tables --------- */

ORACLE_API void *oracle_progress_new(int state, uint64_t match, uint64_t next, uint64_t pending, int paused,
                                     int inflight_size, uint64_t inflight_max_bytes) {
    progress *pr = (progress *)calloc(1, sizeof(*pr));
    pr->state = state;
    pr->match = match;
    pr->next = next;
    pr->pending_snapshot = pending;
    pr->msg_app_flow_paused = paused;
    pr->ins.size = inflight_size;
    pr->ins.max_bytes = inflight_max_bytes;
    pr->ins.buffer = (inflight *)calloc((size_t)(inflight_size > 0 ? inflight_size : 1), sizeof(inflight));
    return pr;
}
ORACLE_API void oracle_progress_free(void *h) {
    progress *pr = (progress *)h;
    free(pr->ins.buffer);
    free(pr);
}
/* out[0..6] = state, match, next, pending, paused flag, IsPaused(), recent_active */
ORACLE_API void oracle_progress_get(const void *h, uint64_t *out) {
    const progress *pr = (const progress *)h;
    out[0] = (uint64_t)pr->state;
    out[1] = pr->match;
    out[2] = pr->next;
    out[3] = pr->pending_snapshot;
    out[4] = (uint64_t)pr->msg_app_flow_paused;
    out[5] = (uint64_t)is_paused(pr);
    out[6] = (uint64_t)pr->recent_active;
}
/* out[0..4] = start, count, bytes, size, Full(); idx/bytes = the ring as laid out */
ORACLE_API void oracle_inflights_get(const void *h, uint64_t *out, uint64_t *idx, uint64_t *bytes) {
    const progress *pr = (const progress *)h;
    out[0] = (uint64_t)pr->ins.start;
    out[1] = (uint64_t)pr->ins.count;
    out[2] = pr->ins.bytes;
    out[3] = (uint64_t)pr->ins.size;
    out[4] = (uint64_t)inflights_full(&pr->ins);
    for (int i = 0; i < pr->ins.size; i++) {
        if (idx) idx[i] = pr->ins.buffer[i].index;
        if (bytes) bytes[i] = pr->ins.buffer[i].bytes;
    }
}
ORACLE_API void oracle_inflights_set_start(void *h, int start) { ((progress *)h)->ins.start = start; }
ORACLE_API int oracle_inflights_add(void *h, uint64_t index, uint64_t bytes) {
    return inflights_add(&((progress *)h)->ins, index, bytes);
}
ORACLE_API void oracle_inflights_free_le(void *h, uint64_t to) { inflights_free_le(&((progress *)h)->ins, to); }
ORACLE_API void oracle_inflights_reset(void *h) { inflights_reset(&((progress *)h)->ins); }
ORACLE_API int oracle_progress_maybe_update(void *h, uint64_t n) { return maybe_update((progress *)h, n); }
ORACLE_API int oracle_progress_maybe_decr_to(void *h, uint64_t rejected, uint64_t hint) {
    return maybe_decr_to((progress *)h, rejected, hint);
}
ORACLE_API void oracle_progress_become(void *h, int state, uint64_t snapshoti) {
    progress *pr = (progress *)h;
    if (state == STATE_PROBE) become_probe(pr);
    else if (state == STATE_REPLICATE) become_replicate(pr);
    else become_snapshot(pr, snapshoti);
}

/* ---- stepLeader's Progress branches, one event at a time ------------------ */

/* event types: raftpb/raft.proto:41-69 plus the local send-side records */
enum {
    EV_APP_RESP = 4,
    EV_HEARTBEAT_RESP = 9,
    EV_UNREACHABLE = 10,
    EV_SNAP_STATUS = 11,
    EV_APPENDED = 0xF0,  /* the leader's log grew to `index` (appendEntry, raft.go:791-819) */
    EV_SENT_APP = 0xF1,  /* a MsgApp went out: UpdateOnEntriesSend (raft.go:653) */
    EV_SENT_SNAP = 0xF2, /* a MsgSnap went out: BecomeSnapshot(index) (raft.go:645) */
    EV_NEEDS_HOST = 0xFE, /* a response only the host can step (a rejection whose hint needs the leader's
                             log, raft.go:1458): it and every later event of the same Progress in the
                             batch are left unstepped (ACT_DEFERRED), so stream order per Progress holds */
};
/* what the reference does next with the message, left to the host */
enum {
    ACT_SEND_APPEND = 1,  /* r.sendAppend(m.From): raft.go:1465, 1545 */
    ACT_UPDATED = 2,      /* the branch of raft.go:1477 was taken: maybeCommit (:1500), then
                             `for r.maybeSendAppend(m.From, false)` (:1516-1519) */
    ACT_WAS_PAUSED = 4,   /* oldPaused (:1468): with ACT_UPDATED and no commit advance, sendAppend (:1505-1509) */
    ACT_CAUGHT_UP = 8,    /* with ACT_UPDATED: pr.Match == lastIndex, the leadTransferee test (:1521) */
    ACT_ERROR = 32,       /* the reference errors / panics on this call (progress.go:160, inflights.go:67) */
    ACT_DEFERRED = 64,    /* not stepped: more than `limit` events for this Progress in one batch */
    ACT_DROPPED = 128,    /* no Progress (raft.go:1327-1331) or not an event type */
};
#define MSGF_READ_CTX 1
#define MSGF_ENTRIES 1

typedef struct {
    uint32_t group;
    uint8_t slot, type, reject, flags;
    uint64_t index;
} oracle_msg;

static uint64_t load_mask(const void *masks, int mask_bytes, uint64_t g) {
    switch (mask_bytes) {
    case 1: return ((const uint8_t *)masks)[g];
    case 2: return ((const uint16_t *)masks)[g];
    case 4: return ((const uint32_t *)masks)[g];
    default: return ((const uint64_t *)masks)[g];
    }
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
 * One message on one Progress.  `aux` is nextProbeIdx for a rejection (what
 * raft.go:1362-1459 computes from RejectHint / LogTerm on the leader's log —
 * the host's job) and the payload byte size for EV_SENT_APP.  *read_ack is set
 * when readOnly.recvAck applies (read_only.go:68-76).
 */
static uint8_t step_one(progress *pr, const oracle_msg *m, uint64_t aux, uint64_t last_index,
                        uint64_t first_index, int *read_ack) {
    uint8_t act = 0;
    switch (m->type) {
    case EV_APP_RESP: /* raft.go:1333-1526 */
        pr->recent_active = 1;
        if (m->reject) {
            if (maybe_decr_to(pr, m->index, aux)) {
                if (pr->state == STATE_REPLICATE) become_probe(pr);
                act |= ACT_SEND_APPEND;
            }
        } else {
            int old_paused = is_paused(pr);
            if (maybe_update(pr, m->index) || (pr->match == m->index && pr->state == STATE_PROBE)) {
                if (pr->state == STATE_PROBE) {
                    become_replicate(pr);
                } else if (pr->state == STATE_SNAPSHOT && pr->match + 1 >= first_index) {
                    become_probe(pr);
                    become_replicate(pr);
                } else if (pr->state == STATE_REPLICATE) {
                    inflights_free_le(&pr->ins, m->index);
                }
                act |= ACT_UPDATED;
                if (old_paused) act |= ACT_WAS_PAUSED;
                if (pr->match == last_index) act |= ACT_CAUGHT_UP;
            }
        }
        break;
    case EV_HEARTBEAT_RESP: /* raft.go:1527-1561 */
        pr->recent_active = 1;
        pr->msg_app_flow_paused = 0;
        if (pr->match < last_index || pr->state == STATE_PROBE) act |= ACT_SEND_APPEND;
        if (m->flags & MSGF_READ_CTX) *read_ack = 1;
        break;
    case EV_SNAP_STATUS: /* raft.go:1562-1579 */
        if (pr->state != STATE_SNAPSHOT) break;
        if (m->reject) pr->pending_snapshot = 0;
        become_probe(pr);
        pr->msg_app_flow_paused = 1;
        break;
    case EV_UNREACHABLE: /* raft.go:1580-1586 */
        if (pr->state == STATE_REPLICATE) become_probe(pr);
        break;
    case EV_SENT_APP: /* raft.go:653 */
        if (update_on_entries_send(pr, m->flags & MSGF_ENTRIES, aux, m->index) != 0) act |= ACT_ERROR;
        break;
    case EV_SENT_SNAP: /* raft.go:645 */
        become_snapshot(pr, m->index);
        break;
    }
    return act;
}

/*
 * A batch of events over [slot][group] planes, in stream order per Progress:
 *   pstate      u8   bits 0-1 State, bit 2 MsgAppFlowPaused
 *   infl_meta   u32  start | count << 16;  infl_bytes u64
 *   ring_index / ring_bytes  u64 [cap][nslots][stride]
 * EV_APPENDED records are applied first (last_index = max), as the engine does.
 * A Progress named by more than `limit` events in the batch is not stepped at
 * all: its events get ACT_DEFERRED (the host resubmits them).
 */
ORACLE_API void oracle_step_progress(uint64_t G, int nslots, uint64_t stride, uint64_t *match, uint64_t *next,
                                     uint8_t *pstate, uint64_t *pending, uint32_t *infl_meta,
                                     uint64_t *infl_bytes, uint64_t *ring_index, uint64_t *ring_bytes, int cap,
                                     uint64_t max_bytes, void *active, void *acks, int mask_bytes,
                                     uint64_t *last_index, const uint64_t *first_index, const oracle_msg *msgs,
                                     const uint64_t *aux, uint64_t n, uint32_t limit, uint8_t *actions) {
    uint32_t *cnt = (uint32_t *)calloc((size_t)nslots * stride, sizeof(uint32_t));
    for (uint64_t i = 0; i < n; i++) {
        const oracle_msg *m = &msgs[i];
        if (m->type == EV_APPENDED) {
            if (m->group < G) {
                if (last_index[m->group] < m->index) last_index[m->group] = m->index;
                actions[i] = 0;
            } else {
                actions[i] = ACT_DROPPED;
            }
            continue;
        }
        const int known = m->type == EV_APP_RESP || m->type == EV_HEARTBEAT_RESP || m->type == EV_UNREACHABLE ||
                          m->type == EV_SNAP_STATUS || m->type == EV_SENT_APP || m->type == EV_SENT_SNAP ||
                          m->type == EV_NEEDS_HOST;
        if (!known || m->group >= G || m->slot >= nslots) {
            actions[i] = ACT_DROPPED;
            continue;
        }
        cnt[(uint64_t)m->slot * stride + m->group]++;
        actions[i] = ACT_DEFERRED;
    }
    const int ring = cap > 0 ? cap : 1;
    inflight *buf = (inflight *)calloc((size_t)ring, sizeof(inflight));
    for (uint64_t i = 0; i < n; i++) {
        const oracle_msg *m = &msgs[i];
        if (actions[i] != ACT_DEFERRED) continue;
        const uint64_t cell = (uint64_t)m->slot * stride + m->group;
        if (cnt[cell] > limit) continue;
        if (m->type == EV_NEEDS_HOST) cnt[cell] = 0xFFFFFFFFu; /* blocks the rest of this Progress's events */
        if (cnt[cell] == 0xFFFFFFFFu) continue;
        progress pr;
        pr.match = match[cell];
        pr.next = next[cell];
        pr.state = pstate[cell] & 3;
        pr.msg_app_flow_paused = (pstate[cell] >> 2) & 1;
        pr.pending_snapshot = pending[cell];
        pr.recent_active = (int)((load_mask(active, mask_bytes, m->group) >> m->slot) & 1);
        pr.ins.start = (int)(infl_meta[cell] & 0xFFFF);
        pr.ins.count = (int)(infl_meta[cell] >> 16);
        pr.ins.bytes = infl_bytes[cell];
        pr.ins.size = cap;
        pr.ins.max_bytes = max_bytes;
        pr.ins.buffer = buf;
        for (int k = 0; k < cap; k++) {
            buf[k].index = ring_index[(uint64_t)k * nslots * stride + cell];
            buf[k].bytes = ring_bytes[(uint64_t)k * nslots * stride + cell];
        }
        int read_ack = 0;
        actions[i] = step_one(&pr, m, aux ? aux[i] : 0, last_index[m->group], first_index[m->group], &read_ack);
        match[cell] = pr.match;
        next[cell] = pr.next;
        pstate[cell] = (uint8_t)(pr.state | (pr.msg_app_flow_paused << 2));
        pending[cell] = pr.pending_snapshot;
        if (pr.recent_active)
            store_mask(active, mask_bytes, m->group, load_mask(active, mask_bytes, m->group) | (1ULL << m->slot));
        if (read_ack)
            store_mask(acks, mask_bytes, m->group, load_mask(acks, mask_bytes, m->group) | (1ULL << m->slot));
        infl_meta[cell] = (uint32_t)pr.ins.start | ((uint32_t)pr.ins.count << 16);
        infl_bytes[cell] = pr.ins.bytes;
        for (int k = 0; k < cap; k++) {
            ring_index[(uint64_t)k * nslots * stride + cell] = buf[k].index;
            ring_bytes[(uint64_t)k * nslots * stride + cell] = buf[k].bytes;
        }
    }
    free(buf);
    free(cnt);
}

/* ------------------------------------------------------------------------ */
/* bcastAppend as a plan: what maybeSendAppend decides, message by message.    */
/* ------------------------------------------------------------------------ */

enum { SEND_NONE = 0, SEND_APPEND = 1, SEND_EMPTY = 2, SEND_SNAPSHOT = 3 };

/* log.go:381-397 — does raftLog.term(i) return an error?  (the valid range is
 * [firstIndex-1, lastIndex]) */
static int log_term_fails(uint64_t i, uint64_t first_index, uint64_t last_index) {
    if (i + 1 < first_index) return 1; /* ErrCompacted */
    if (i > last_index) return 1;      /* ErrUnavailable */
    return 0;
}

/* log.go:409-414 + mustCheckOutOfBounds :543-557 — raftLog.entries(i, maxSize):
 * *n_entries > 0 iff at least one entry comes back (a size limit never returns
 * zero entries); returns 1 on ErrCompacted. */
static int log_entries(uint64_t i, uint64_t first_index, uint64_t last_index, int *n_entries) {
    *n_entries = 0;
    if (i > last_index) return 0; /* nil, nil */
    if (i < first_index) return 1;
    *n_entries = 1;
    return 0;
}

/* raft.go:600-666 — maybeSendAppend(to, sendIfEmpty), up to the point where
 * the message would be built. */
static int maybe_send_append(const progress *pr, uint64_t first_index, uint64_t last_log_index, int send_if_empty) {
    if (is_paused(pr)) return SEND_NONE;
    uint64_t last_index = pr->next - 1, next_index = pr->next;
    int errt = log_term_fails(last_index, first_index, last_log_index);
    int n_ents = 0, erre = 0;
    if (pr->state != STATE_REPLICATE || !inflights_full(&pr->ins))
        erre = log_entries(next_index, first_index, last_log_index, &n_ents);
    if (n_ents == 0 && !send_if_empty) return SEND_NONE;
    if (errt || erre) { /* send snapshot if we failed to get term or entries */
        if (!pr->recent_active) return SEND_NONE;
        return SEND_SNAPSHOT;
    }
    return n_ents > 0 ? SEND_APPEND : SEND_EMPTY;
}

/*
 * raft.go:689-696 — bcastAppend for G groups: Visit every Progress but r.id's.
 * rows = the slots that hold a Progress (NULL: all of them); a group without a
 * self bit is not led here and plans nothing.  plan is [nslots][stride].
 */
ORACLE_API void oracle_plan_appends(uint64_t G, int nslots, uint64_t stride, const uint64_t *next,
                                    const uint8_t *pstate, const uint32_t *infl_meta, const uint64_t *infl_bytes,
                                    int cap, uint64_t max_bytes, const void *rows, const void *self_mask,
                                    const void *active, int mask_bytes, const uint64_t *first_index,
                                    const uint64_t *last_index, int send_if_empty, uint8_t *plan) {
    for (uint64_t g = 0; g < G; g++) {
        uint64_t self = load_mask(self_mask, mask_bytes, g);
        uint64_t have = rows ? load_mask(rows, mask_bytes, g) : ~0ULL;
        uint64_t act = load_mask(active, mask_bytes, g);
        for (int s = 0; s < nslots; s++) {
            uint64_t cell = (uint64_t)s * stride + g;
            plan[cell] = SEND_NONE;
            if (self == 0 || ((self >> s) & 1) || !((have >> s) & 1)) continue;
            progress pr;
            memset(&pr, 0, sizeof pr);
            pr.next = next[cell];
            pr.state = pstate[cell] & 3;
            pr.msg_app_flow_paused = (pstate[cell] >> 2) & 1;
            pr.recent_active = (int)((act >> s) & 1);
            pr.ins.count = (int)(infl_meta[cell] >> 16);
            pr.ins.bytes = infl_bytes[cell];
            pr.ins.size = cap;
            pr.ins.max_bytes = max_bytes;
            plan[cell] = (uint8_t)maybe_send_append(&pr, first_index[g], last_index[g], send_if_empty);
        }
    }
}
