/*
 * rafthip.h — C ABI of librafthip.so, the MI355X (gfx950) batched quorum engine.
 *
 * This is the drop-in boundary for ONE path of the reference (Elbehery/raft, a
 * fork of go.etcd.io/raft/v3): evaluating, for many independent Raft groups at
 * once, what quorum/ and tracker/ compute per group.  The reference has no FFI
 * today; these entry points are what a cgo shim behind the new
 * Go functions quorum.BatchCommittedIndex / quorum.BatchVoteResult binds to
 * (INTEGRATION.md shows that shim).  Each declaration cites the reference
 * code it replaces (paths relative to the reference root).
 *
 * Data model.  One engine per GPU holds G groups × S slots.  A slot is one
 * Progress row of a group (tracker/tracker.go:117-126): a voter of the
 * incoming config (Voters[0]), of the outgoing config (Voters[1]), or a
 * learner.  The id→slot map is host-owned and changes only on conf change
 * (confchange/confchange.go:51-145).  Device planes are laid out
 * [slot][group] so that consecutive lanes read consecutive groups:
 *
 *   match    uint64 [S][G]   Progress.Match                (tracker/progress.go:31)
 *   votes    uint8  [S][G]   0 no vote, 1 rejected, 2 granted
 *                            (encoding of quorum/datadriven_test.go:55-58,94-100)
 *   in/out/learner masks, active, acks:  one slot bitmask per group, bit s =
 *                            slot s; width rafthip_mask_bytes() = 1/2/4/8 bytes
 *                            for S <= 8/16/32/64
 *   commit   uint64 [G]      last computed quorum commit index
 *   after rafthip_progress_config: one 32-byte record per Progress row (Next,
 *                            PendingSnapshot, State, MsgAppFlowPaused, Inflights
 *                            start/count/bytes) and Inflights rings [entry][S][G];
 *                            the ABI still moves them as [S][G] planes
 *
 * Conventions.  Every function returns 0 or a negative errno-style code and
 * never throws or aborts (the reference's quorum/tracker functions return no
 * errors; misuse panics — here misuse is an error return).  Pointers named
 * `src`/`dst` may be host OR device memory (copies use hipMemcpyDefault).
 * Compute calls are asynchronous on the engine's stream; read calls and
 * rafthip_synchronize() wait for it.  Calls on one engine must be serialised
 * by the caller (RawNode is "thread-unsafe" too, rawnode.go:31); different
 * engines may be driven from different threads.  All arithmetic is uint64 /
 * uint8 and bit-exact with the reference; there is no CPU fallback: without a
 * usable GPU rafthip_create fails with RAFTHIP_ENODEV.
 */
#ifndef RAFTHIP_H
#define RAFTHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAFTHIP_ABI_VERSION 1u
#define RAFTHIP_MAX_SLOTS 64u

/* status codes */
#define RAFTHIP_OK 0
#define RAFTHIP_EINVAL (-22) /* bad argument / shape */
#define RAFTHIP_ENOMEM (-12) /* device or host allocation failed */
#define RAFTHIP_ENODEV (-19) /* no such GPU / not a usable HIP device */
#define RAFTHIP_EIO (-5)     /* a HIP runtime call failed; see rafthip_last_error */
#define RAFTHIP_ERANGE (-34) /* group range outside [0, num_groups) */
#define RAFTHIP_EBUSY (-16)  /* a shared resource is held by an overlapping call (rafthip_commit_changed_static) */

/* quorum.VoteResult — quorum/quorum.go:48-58 (0 is not a valid result) */
#define RAFTHIP_VOTE_PENDING 1
#define RAFTHIP_VOTE_LOST 2
#define RAFTHIP_VOTE_WON 3

/* vote-plane cells — quorum/datadriven_test.go:55-58 */
#define RAFTHIP_CELL_NONE 0
#define RAFTHIP_CELL_REJECT 1
#define RAFTHIP_CELL_GRANT 2

typedef struct rafthip_engine rafthip_engine;

/* Engines with more than 8 slots: sort each group across the lanes of a
 * wavefront (bitonic network over DPP / ds_swizzle, LDS-staged tile) instead
 * of the default one-group-per-lane register network.  Same results; kept
 * selectable for A/B measurement. */
#define RAFTHIP_FLAG_WIDE_BITONIC 1u
/* rafthip_scatter_acks has two forms with the same result: one 64-bit atomic max per
 * ack, and — for batches of device-resident records that touch a fair share of the
 * engine's cells — a partition of the acks by tile followed by one LDS-staged update
 * of every tile.  By default the call picks by batch size;
 * _SCATTER_ATOMIC pins the first form, _SCATTER_PARTITION the second whenever the
 * engine's shape admits it (for A/B measurement and tests). */
#define RAFTHIP_FLAG_SCATTER_ATOMIC 2u
#define RAFTHIP_FLAG_SCATTER_PARTITION 4u
/* The Progress stepper (rafthip_step_progress, rafthip_step_wire) has two forms with the
 * same result: K9 links the events of every Progress through a list head in its record
 * (one scattered exchange per event) and steps them wherever they lie; K9p first brings
 * the events into tile order with the partition passes of K5p and then links and steps
 * them tile by tile, list heads in LDS.  By default the call picks by batch
 * size; _STEP_LINKED pins the first form, _STEP_PARTITION the second whenever the engine's
 * shape admits it (for A/B measurement and tests). */
#define RAFTHIP_FLAG_STEP_LINKED 8u
#define RAFTHIP_FLAG_STEP_PARTITION 16u

typedef struct rafthip_config {
    uint32_t struct_size; /* sizeof(rafthip_config), for ABI growth */
    int32_t device;       /* HIP device ordinal */
    uint64_t num_groups;  /* G >= 1 */
    uint32_t num_slots;   /* S in [1, 64] */
    uint32_t flags;       /* RAFTHIP_FLAG_* or 0 */
    void *stream;         /* hipStream_t to run on, or NULL: the engine makes its own */
} rafthip_config;

/* One non-reject MsgAppResp, already resolved to (group, slot):
 * raftpb/raft.proto:71-108 Message{From→slot, Index}; handler raft.go:1333,1477. */
typedef struct rafthip_ack {
    uint32_t group;
    uint32_t slot;
    uint64_t index;
} rafthip_ack;

/* One response message a leader received, resolved by the host to (group,
 * slot) — the fields of raftpb.Message (raftpb/raft.proto:71-108) that
 * stepLeader's per-message bookkeeping reads (raft.go:1327-1561). */
#define RAFTHIP_MSG_APP_RESP 4        /* pb.MsgAppResp,       raftpb/raft.proto:46 */
#define RAFTHIP_MSG_HEARTBEAT_RESP 9  /* pb.MsgHeartbeatResp, raftpb/raft.proto:51 */
#define RAFTHIP_MSG_APPENDED 0xF0     /* local, not a wire type: the leader's log grew to `index`
                                         (appendEntry, raft.go:787-819); slot is ignored */
#define RAFTHIP_MSGF_READ_CTX 1       /* heartbeat response whose Context names the group's pending
                                         ReadIndex request (read_only.go:68-76) */
typedef struct rafthip_msg {
    uint32_t group;
    uint8_t slot;    /* Message.From -> slot */
    uint8_t type;    /* RAFTHIP_MSG_* */
    uint8_t reject;  /* Message.Reject */
    uint8_t flags;   /* RAFTHIP_MSGF_* */
    uint64_t index;  /* Message.Index */
} rafthip_msg;

/* A group whose quorum commit index changed — what raft.maybeCommit
 * (raft.go:755-758) hands to raftLog.maybeCommit (log.go:447-456). */
typedef struct rafthip_commit_pair {
    uint64_t group;
    uint64_t commit;
} rafthip_commit_pair;

typedef struct rafthip_stats_t {
    uint64_t commit_launches;   /* commit passes launched */
    uint64_t groups_evaluated;  /* groups covered by all compute launches */
    uint64_t acks_applied;      /* ack records submitted to rafthip_scatter_acks */
    uint64_t acks_rejected;     /* of those, dropped: group/slot out of range */
    uint64_t changed_reported;  /* pairs returned by rafthip_commit_changed */
    uint64_t bytes_uploaded;
    uint64_t bytes_downloaded;
    uint64_t advanced_reported; /* pairs returned by rafthip_commit_advance */
    uint64_t msgs_ingested;     /* records submitted to rafthip_ingest_msgs */
    uint64_t msgs_dropped;      /* of those, dropped: no Progress (group/slot out of range) */
    uint64_t msgs_ignored;      /* of those, of a type the engine does not handle */
} rafthip_stats_t;

/* ---- lifecycle ---------------------------------------------------------- */

uint32_t rafthip_abi_version(void);
/* Thread-local description of the last failure on this thread ("" if none). */
const char *rafthip_last_error(void);
int rafthip_device_count(int *count);

/* Page-locked host memory for the buffers a host fills every tick (ack / message
 * records, serialized bytes, pair landing buffers): uploads from it run at the
 * link rate instead of through the runtime's pageable staging copies.  No
 * counterpart in the reference; any host pointer is accepted everywhere, this
 * is the fast kind.  Free with rafthip_host_free. */
int rafthip_host_alloc(void **ptr, uint64_t bytes);
int rafthip_host_free(void *ptr);

/* Stands where tracker.MakeProgressTracker (tracker/tracker.go:129-145) stands
 * per group: allocates zeroed planes; config is "empty" (both halves empty,
 * which commits MaxUint64 / wins votes, majority.go:128-132,179-184). */
int rafthip_create(const rafthip_config *cfg, rafthip_engine **out);
int rafthip_destroy(rafthip_engine *e);
int rafthip_synchronize(rafthip_engine *e);
/* 1, 2, 4 or 8: bytes per slot bitmask for this engine's S. */
int rafthip_mask_bytes(const rafthip_engine *e);
int rafthip_stats(rafthip_engine *e, rafthip_stats_t *out);

/* ---- configuration: tracker.Config (tracker/tracker.go:27-78) ----------- */

/* Every group: Voters[0] = slots [0, voters), Voters[1] empty, the remaining
 * slots learners.  The non-joint steady state (confchange.go:318-329); lets
 * the kernels skip the mask planes. */
int rafthip_set_uniform_config(rafthip_engine *e, uint32_t voters);

/* Per-group configs for groups [first, first+count): bit s of in_mask[g] /
 * out_mask[g] says slot s is in Voters[0] / Voters[1] (quorum/joint.go:19);
 * learner_mask (may be NULL = none) marks Progress.IsLearner rows
 * (tracker/progress.go:97).  Shapes per confchange/confchange.go:276-332. */
int rafthip_upload_config(rafthip_engine *e, uint64_t first, uint64_t count, const void *in_mask,
                          const void *out_mask, const void *learner_mask);

/* ---- state -------------------------------------------------------------- */

/* Progress.Match for groups [first, first+count), all S planes; plane s of the
 * source starts at src + s*src_stride (elements). */
int rafthip_upload_match(rafthip_engine *e, uint64_t first, uint64_t count, const uint64_t *src,
                         uint64_t src_stride);
/* Match = 0 for the range: what raft.reset does on every role change
 * (raft.go:774-784). */
int rafthip_reset_match(rafthip_engine *e, uint64_t first, uint64_t count);
int rafthip_read_match(rafthip_engine *e, uint64_t first, uint64_t count, uint64_t *dst,
                       uint64_t dst_stride);

/* ProgressTracker.Votes (tracker/tracker.go:122,254-265) as vote-cell planes. */
int rafthip_upload_votes(rafthip_engine *e, uint64_t first, uint64_t count, const uint8_t *src,
                         uint64_t src_stride);
/* ProgressTracker.ResetVotes — tracker/tracker.go:254-256. */
int rafthip_reset_votes(rafthip_engine *e, uint64_t first, uint64_t count);
/* The vote cells as the engine holds them (after rafthip_record_votes). */
int rafthip_read_votes(rafthip_engine *e, uint64_t first, uint64_t count, uint8_t *dst, uint64_t dst_stride);

/* Progress.RecentActive bits (tracker/progress.go:72; set at raft.go:1337,1528). */
int rafthip_upload_active(rafthip_engine *e, uint64_t first, uint64_t count, const void *src);
/* readIndexStatus.acks bits (read_only.go:29-37,68-76). */
int rafthip_upload_acks(rafthip_engine *e, uint64_t first, uint64_t count, const void *src);
/* The leader's own slot as a one-bit mask per group (0 = the leader has no
 * Progress row): the id != r.id test of raft.go:1238-1242. */
int rafthip_upload_self(rafthip_engine *e, uint64_t first, uint64_t count, const void *src);
/* RecentActive / ReadIndex-ack bits as the engine holds them now (after
 * rafthip_ingest_msgs, rafthip_check_quorum). */
int rafthip_read_active(rafthip_engine *e, uint64_t first, uint64_t count, void *dst);
int rafthip_read_acks(rafthip_engine *e, uint64_t first, uint64_t count, void *dst);
/* For rafthip_commit_advance: the log indexes of each group whose entry carries
 * the leader's current term, [term_first, term_last] (terms never decrease
 * along a log, so they are one run; for a leader term_last is lastIndex).
 * An empty range (first > last; the zero-initialised default counts as empty)
 * means "term(i) != r.Term for every i": not a leader, or r.Term == 0. */
int rafthip_upload_term_range(rafthip_engine *e, uint64_t first, uint64_t count,
                              const uint64_t *term_first, const uint64_t *term_last);
/* Either pointer may be NULL.  term_last also grows through RAFTHIP_MSG_APPENDED records. */
int rafthip_read_term_range(rafthip_engine *e, uint64_t first, uint64_t count, uint64_t *term_first,
                            uint64_t *term_last);
/* Seed the commit plane (the value rafthip_commit_changed compares against;
 * raftLog.committed for rafthip_commit_advance). */
int rafthip_upload_commit(rafthip_engine *e, uint64_t first, uint64_t count, const uint64_t *src);

/* ---- compute ------------------------------------------------------------ */

/* Progress.MaybeUpdate's Match part for a batch of acks — tracker/progress.go:168-177
 * via raft.go:1477: Match = max(Match, index).  Max is commutative and
 * idempotent, so the batch result equals applying the acks in any order.
 * Records with group >= G or slot >= S are dropped and counted
 * (stats.acks_rejected; the reference drops acks from unknown peers,
 * raft.go:1327-1331). */
int rafthip_scatter_acks(rafthip_engine *e, const rafthip_ack *recs, uint64_t n);

/* The same MaybeUpdate when the tick's acks are a dense plane rather than a
 * record stream: acks[s*src_stride + g] = the index slot s of group first+g
 * acknowledged, 0 = nothing; Match = max(Match, ack) cell by cell
 * (tracker/progress.go:168-177).  Streams at HBM rate (24 B per cell) instead
 * of one scattered atomic per ack: the better form once more than about one
 * cell in ten carries an ack.  acks may be host or device memory. */
int rafthip_merge_acks(rafthip_engine *e, uint64_t first, uint64_t count, const uint64_t *acks,
                       uint64_t src_stride);

/* stepLeader's bookkeeping for a batch of responses — raft.go:1327-1561:
 *   MsgAppResp:       pr.RecentActive = true (raft.go:1337); unless Reject,
 *                     pr.MaybeUpdate(m.Index) (raft.go:1477 -> tracker/progress.go:168-177)
 *   MsgHeartbeatResp: pr.RecentActive = true (raft.go:1528); with
 *                     RAFTHIP_MSGF_READ_CTX, readOnly.recvAck: acks[from] = true
 *                     (raft.go:1552 -> read_only.go:68-76)
 *   RAFTHIP_MSG_APPENDED: term_last = max(term_last, index)
 * All updates are max / or, so the batch equals stepping the messages one by
 * one.  Records without a Progress row (group >= G, slot >= S) are dropped as
 * raft.go:1327-1331 drops them (stats.msgs_dropped); other types are counted
 * in stats.msgs_ignored.  The Progress FSM side effects (Probe/Replicate,
 * Inflights, MaybeDecrTo on Reject) stay with the host. */
int rafthip_ingest_msgs(rafthip_engine *e, const rafthip_msg *msgs, uint64_t n);

/* ProgressTracker.RecordVote for a batch of vote responses — tracker/tracker.go:260-265
 * via raft.poll (raft.go:1041-1049): records of type MsgVoteResp (6) or
 * MsgPreVoteResp (18) (the host submits the kind its campaign counts,
 * raft.go:1647-1650), vote = !reject; the FIRST vote of a (group, slot) sticks —
 * first in stream order, also between two records of one batch, also against a
 * vote recorded before the batch (rafthip_upload_votes, earlier calls).  Records
 * of other types or without a Progress row are dropped (stats.msgs_dropped).
 * Followed by rafthip_tally_votes. */
#define RAFTHIP_MSG_VOTE_RESP 6
#define RAFTHIP_MSG_PRE_VOTE_RESP 18
int rafthip_record_votes(rafthip_engine *e, const rafthip_msg *msgs, uint64_t n);

/* ---- wire ingest: the same step from serialized raftpb.Message bytes ------ */

/* The raft id behind each Progress row: ids[s*src_stride + g] is the key of
 * tracker.ProgressMap (tracker/progress.go:263) that slot s of group first+g
 * holds, 0 (raft.None) = empty slot.  The lookup `r.trk.Progress[m.From]` of
 * raft.go:1327 runs against this table on the device. */
int rafthip_upload_ids(rafthip_engine *e, uint64_t first, uint64_t count, const uint64_t *ids,
                       uint64_t src_stride);
/* r.Term of each group, for the term gate of raft.Step (raft.go:1051-1139). */
int rafthip_upload_term(rafthip_engine *e, uint64_t first, uint64_t count, const uint64_t *term);
/* The pending ReadIndex request context of each group (readOnly.pendingReadIndex
 * key, read_only.go:29-37,68-76): ctx_len[g] bytes (0 = none pending, at most 8)
 * held little-endian in ctx[g] (byte i of the context = bits 8i..8i+7).  A
 * length above 8 is RAFTHIP_EINVAL for host input; in device-resident input it
 * is stored as "none pending" (such a host keeps that group's reads in Go). */
int rafthip_upload_read_ctx(rafthip_engine *e, uint64_t first, uint64_t count, const uint64_t *ctx,
                            const uint8_t *ctx_len);
/* The vote response type each group's campaign counts — stepCandidate's
 * myVoteRespType, raft.go:1628-1633: RAFTHIP_MSG_PRE_VOTE_RESP for a
 * StatePreCandidate, RAFTHIP_MSG_VOTE_RESP for a StateCandidate, 0 for every
 * other state (leaders and followers ignore vote responses).  While any group
 * has one, the wire entry points also step vote responses: after the term gate
 * of raft.Step — a granted MsgPreVoteResp passes with a higher term, it carries
 * the pre-candidate's future term (raft.go:1068-1076) — a response of the
 * group's own type from a peer with a Progress row is recorded as
 * rafthip_record_votes records it (raft.poll, raft.go:1041-1049) and comes back
 * RAFTHIP_WIRE_VOTE; the other type, or a group that is not campaigning:
 * RAFTHIP_WIRE_OTHER_TYPE.  Follow with rafthip_tally_votes: the result after a
 * batch is the result the reference reaches one response at a time (Won and
 * Lost are final: a later vote moves one peer from `missing` to yes or no and
 * can undo neither); rafthip_reset_votes when a campaign starts
 * (ProgressTracker.ResetVotes, raft.go:784).  resp_type NULL with the whole
 * range clears every group and turns the vote passes off again. */
int rafthip_upload_campaign(rafthip_engine *e, uint64_t first, uint64_t count, const uint8_t *resp_type);

/* What happened to one serialized message (rafthip_ingest_wire). */
#define RAFTHIP_WIRE_APPLIED 0     /* stepped as rafthip_ingest_msgs steps its record */
#define RAFTHIP_WIRE_OTHER_TYPE 1  /* well-formed, not a response this path steps: the host steps it */
#define RAFTHIP_WIRE_NO_PROGRESS 2 /* no Progress for m.From: dropped, raft.go:1327-1331 */
#define RAFTHIP_WIRE_STALE_TERM 3  /* m.Term < r.Term: ignored, raft.go:1087-1138 */
#define RAFTHIP_WIRE_HIGHER_TERM 4 /* m.Term > r.Term: the host must step down (raft.go:1056-1085); not applied */
#define RAFTHIP_WIRE_NESTED 5      /* carries a non-empty entries / snapshot / responses payload: the host
                                      decodes it; not applied */
#define RAFTHIP_WIRE_MALFORMED 6   /* Message.Unmarshal returns an error (raftpb/raft.pb.go:1801-2179) */
#define RAFTHIP_WIRE_BAD_GROUP 7   /* envelope group id >= num_groups */
#define RAFTHIP_WIRE_NEEDS_LOG 8   /* rafthip_step_wire only: a rejection with LogTerm > 0, whose next probe index needs
                                      findConflictByTerm on the leader's log (raft.go:1458): the host steps it */
#define RAFTHIP_WIRE_VOTE 9        /* a vote response of the group's campaign (rafthip_upload_campaign): recorded */
#define RAFTHIP_WIRE_STATUSES 10
#define RAFTHIP_MSG_NEEDS_HOST 0xFE /* record type of such a message: it and every later event of the same Progress
                                       in the batch come back RAFTHIP_ACT_DEFERRED, so stream order per Progress holds */
#define RAFTHIP_MSGF_HAS_CTX 2     /* out record flag: Message.Context is non-empty */
#define RAFTHIP_WIRE_NO_SLOT 0xFF  /* out record slot / type when unresolved / outside uint8 */

/* Decode n gogo-protobuf serialized raftpb.Message values (raftpb/raft.proto:71-108,
 * Unmarshal raftpb/raft.pb.go:1801-2179, skipRaft :3027-3104) and step the
 * responses among them on the device: message i is bytes[offsets[i], offsets[i+1])
 * and is addressed to group groups[i] (the multi-raft envelope; raftpb.Message
 * carries no group id).  For MsgAppResp / MsgHeartbeatResp: the term gate of
 * raft.Step (Term 0 = local, passes), the Progress lookup by m.From against the
 * id table, then exactly rafthip_ingest_msgs' updates; a heartbeat response
 * records a ReadIndex ack iff its Context equals the group's pending one.
 * Everything else is classified, never applied.  out_msgs (decoded records,
 * slot resolved), out_status (one RAFTHIP_WIRE_* per message) and
 * status_counts[RAFTHIP_WIRE_STATUSES] may each be NULL; all pointers may be
 * host or device memory except status_counts (host).  Requires
 * rafthip_upload_ids and rafthip_upload_term. */
int rafthip_ingest_wire(rafthip_engine *e, const uint8_t *bytes, const uint64_t *offsets,
                        const uint32_t *groups, uint64_t n, rafthip_msg *out_msgs,
                        uint8_t *out_status, uint64_t *status_counts);

/* ---- per-peer replication state: tracker.Progress + tracker.Inflights ------ */

/* tracker.StateType, tracker/state.go:18-34. */
#define RAFTHIP_STATE_PROBE 0
#define RAFTHIP_STATE_REPLICATE 1
#define RAFTHIP_STATE_SNAPSHOT 2
#define RAFTHIP_PROGRESS_PAUSED 4 /* bit 2 of a state byte: Progress.MsgAppFlowPaused */

/* Event types of rafthip_step_progress beyond MsgAppResp / MsgHeartbeatResp /
 * RAFTHIP_MSG_APPENDED (raftpb/raft.proto:41-69 and two local send-side records). */
#define RAFTHIP_MSG_UNREACHABLE 10 /* MsgUnreachable, raft.go:1580-1586 */
#define RAFTHIP_MSG_SNAP_STATUS 11 /* MsgSnapStatus, raft.go:1562-1579; reject = the snapshot failed */
#define RAFTHIP_MSG_SENT_APP 0xF1  /* a MsgApp went out: Progress.UpdateOnEntriesSend (raft.go:653,
                                      tracker/progress.go:141-163); flags bit 0 = it carried entries,
                                      index = index of its last entry, aux = payload bytes */
#define RAFTHIP_MSG_SENT_SNAP 0xF2 /* a MsgSnap went out: Progress.BecomeSnapshot(index), raft.go:645 */

/* What the reference does next with a stepped message — left to the host. */
#define RAFTHIP_ACT_SEND_APPEND 1 /* r.sendAppend(m.From): raft.go:1465, 1545 */
#define RAFTHIP_ACT_UPDATED 2     /* the branch of raft.go:1477 ran: maybeCommit (the commit pass), then
                                     `for r.maybeSendAppend(m.From, false)`, raft.go:1516-1519 */
#define RAFTHIP_ACT_WAS_PAUSED 4  /* oldPaused, raft.go:1468: with UPDATED and no commit advance, sendAppend (:1505-1509) */
#define RAFTHIP_ACT_CAUGHT_UP 8   /* with UPDATED: pr.Match == lastIndex, the leadTransferee test of raft.go:1521 */
#define RAFTHIP_ACT_ERROR 32      /* the reference errors or panics on this call (progress.go:160, inflights.go:66-68): not applied */
#define RAFTHIP_ACT_DEFERRED 64   /* not stepped: its Progress is named by more than RAFTHIP_STEP_LIMIT events of this batch */
#define RAFTHIP_ACT_DROPPED 128   /* no Progress row (raft.go:1327-1331) or not an event type */
#define RAFTHIP_STEP_LIMIT 64

/* Allocate the per-peer planes: Next, State, MsgAppFlowPaused, PendingSnapshot
 * and an Inflights ring of max_inflight entries per Progress (Config.MaxInflightMsgs,
 * raft.go:205-211; max_inflight_bytes = Config.MaxInflightBytes, 0 = no limit,
 * raft.go:212-220; tracker.NewInflights, tracker/inflights.go:46-51), at most
 * 16383 entries.  Device memory: S * stride * (16 * max_inflight + 32) bytes.
 * Once per engine. */
int rafthip_progress_config(rafthip_engine *e, uint32_t max_inflight, uint64_t max_inflight_bytes);

/* (Re)initialise Progress rows — what raft.reset (raft.go:774-784), becomeLeader
 * (raft.go:915-920) or a conf change (confchange/confchange.go:231-272) leave
 * behind: state[s*src_stride + g] = RAFTHIP_STATE_* | RAFTHIP_PROGRESS_PAUSED,
 * next, pending (NULL = 0).  The Inflights of the range are emptied
 * (Progress.ResetState, tracker/progress.go:102-107).  Match is rafthip_upload_match's. */
int rafthip_upload_progress(rafthip_engine *e, uint64_t first, uint64_t count, const uint8_t *state,
                            const uint64_t *next, const uint64_t *pending, uint64_t src_stride);
/* One new peer — Changer.initProgress, confchange/confchange.go:257-270: the row
 * (slot, group) becomes {state, Next = next, PendingSnapshot = pending, empty
 * Inflights}; the reference creates it as StateProbe with Next = c.LastIndex
 * (confchange/confchange.go:262; 0 for a peer added to an empty log).
 * The group's other rows keep their state and Inflights (rafthip_upload_progress
 * re-initialises whole groups).  Match is rafthip_upload_match's. */
int rafthip_init_progress_row(rafthip_engine *e, uint64_t group, uint32_t slot, uint8_t state, uint64_t next,
                              uint64_t pending);
/* Any destination may be NULL.  infl_meta = Inflights.start | Inflights.count << 16,
 * infl_bytes = Inflights.bytes (tracker/inflights.go:28-40). */
int rafthip_read_progress(rafthip_engine *e, uint64_t first, uint64_t count, uint8_t *state, uint64_t *next,
                          uint64_t *pending, uint32_t *infl_meta, uint64_t *infl_bytes, uint64_t dst_stride);
/* The Inflights rings as laid out in the reference's buffer: entry k of slot s of
 * group first+g at [(k*S + s)*stride + g], k < max(max_inflight, 1). */
int rafthip_upload_inflights(rafthip_engine *e, uint64_t first, uint64_t count, const uint32_t *infl_meta,
                             const uint64_t *infl_bytes, const uint64_t *ring_index, const uint64_t *ring_bytes,
                             uint64_t src_stride);
int rafthip_read_inflights(rafthip_engine *e, uint64_t first, uint64_t count, uint64_t *ring_index,
                           uint64_t *ring_bytes, uint64_t dst_stride);
/* raftLog.firstIndex() per group, for the StateSnapshot exit of raft.go:1481 (default 0). */
int rafthip_upload_first_index(rafthip_engine *e, uint64_t first, uint64_t count, const uint64_t *first_index);

/* stepLeader's Progress branches for a batch of events — raft.go:1333-1586 —
 * with the send-side bookkeeping of maybeSendAppend (raft.go:645,653) as events
 * of the same stream:
 *   MsgAppResp        RecentActive; Reject: MaybeDecrTo(m.Index, aux) and, from
 *                     StateReplicate, BecomeProbe (raft.go:1339-1466; aux = nextProbeIdx,
 *                     i.e. RejectHint after raft.go:1362-1459's findConflictByTerm on the
 *                     host's log); else MaybeUpdate and Probe->Replicate,
 *                     Snapshot->Probe->Replicate (needs firstIndex) or Inflights.FreeLE
 *                     (raft.go:1467-1526)
 *   MsgHeartbeatResp  RecentActive, MsgAppFlowPaused = false, ReadIndex ack with
 *                     RAFTHIP_MSGF_READ_CTX (raft.go:1527-1561)
 *   MsgSnapStatus, MsgUnreachable, RAFTHIP_MSG_SENT_APP, RAFTHIP_MSG_SENT_SNAP   as above
 *   RAFTHIP_MSG_APPENDED  term_last = max(term_last, index); every such record of
 *                     the batch applies BEFORE its responses
 * Events of one Progress are stepped in stream order, events of different
 * Progress rows are independent on this path, so the result equals stepping
 * the batch one message at a time with every send deferred to the end of the
 * batch (the host performs the sends the action bytes ask for, and reports
 * them as SENT records of a later batch).  lastIndex is term_last
 * (rafthip_upload_term_range).  aux (one uint64 per event) may be NULL = all 0;
 * actions (one RAFTHIP_ACT_* byte set per event) may be NULL; both may be host or
 * device memory.  Requires rafthip_progress_config.  n <= 2^31. */
int rafthip_step_progress(rafthip_engine *e, const rafthip_msg *msgs, const uint64_t *aux, uint64_t n,
                          uint8_t *actions);

/* bcastAppend as a plan — raft.go:689-696: for every Progress row of groups
 * [first, first+count) but the leaders' own, what maybeSendAppend(to,
 * sendIfEmpty) (raft.go:600-666) decides to send right now, without building
 * the message: plan[s*stride + g] is one of RAFTHIP_SEND_*; next (may be NULL)
 * receives pr.Next of the same row — a MsgApp's Index is Next-1 and its entries
 * start at Next.  The decision reads IsPaused (tracker/progress.go:225-236),
 * Inflights.Full (inflights.go:131-133), RecentActive, raftLog.firstIndex()
 * (rafthip_upload_first_index) and lastIndex (term_last).  Rows outside the
 * config and groups without a self bit (not led here) plan nothing.  After a
 * send the host submits RAFTHIP_MSG_SENT_APP / _SENT_SNAP records as before
 * (UpdateOnEntriesSend / BecomeSnapshot, raft.go:645,653).  Both outputs host or
 * device memory. */
#define RAFTHIP_SEND_NONE 0
#define RAFTHIP_SEND_APPEND 1   /* MsgApp with entries from pr.Next, as many as maxMsgSize admits */
#define RAFTHIP_SEND_EMPTY 2    /* MsgApp without entries: conveys the commit index */
#define RAFTHIP_SEND_SNAPSHOT 3 /* term(Next-1) or the entries are gone: MsgSnap (raft.go:625-650) */
int rafthip_plan_appends(rafthip_engine *e, uint64_t first, uint64_t count, int send_if_empty, uint8_t *plan,
                         uint64_t *next, uint64_t stride);

/* rafthip_step_progress for a tick whose responses are dense — in steady state
 * every follower of every group acks and answers its heartbeat every tick:
 *   acks[s*ack_stride + g]  acknowledged index + 1 of a non-reject MsgAppResp from
 *                           slot s of group first+g, 0 = none (NULL = no acks)
 *   hb_mask[g]              slot bitmask: a MsgHeartbeatResp arrived (NULL = none);
 *   hb_ctx_mask[g]          of those, the ones carrying the pending ReadIndex context
 * Per Progress the MsgAppResp is stepped first, then the MsgHeartbeatResp
 * (raft.go:1333-1561); actions[s*act_stride + g] (may be NULL) receives the two
 * RAFTHIP_ACT_* bytes or-ed.  One lane per group streams the records, Match and
 * ack planes at HBM rate: about 5 times faster than the same responses as
 * records once a third of the Progress rows are touched.  Rejections, snapshot
 * status and send records go through rafthip_step_progress (before or after:
 * the host's choice of order).  Pointers may be host or device memory. */
int rafthip_step_dense(rafthip_engine *e, uint64_t first, uint64_t count, const uint64_t *acks,
                       uint64_t ack_stride, const void *hb_mask, const void *hb_ctx_mask, uint8_t *actions,
                       uint64_t act_stride);

/* rafthip_ingest_wire's decode and classification feeding rafthip_step_progress
 * instead of the max / or updates: serialized responses in, Progress transitions
 * and action bytes out, no host pass in between.  A message with status
 * RAFTHIP_WIRE_APPLIED is stepped (a rejection with LogTerm 0 probes at
 * m.RejectHint, raft.go:1362); RAFTHIP_WIRE_NEEDS_LOG is left to the host (which
 * steps it with rafthip_step_progress once it has the hint, then resubmits the
 * DEFERRED events of that Progress, order kept); every other status is not an
 * event (RAFTHIP_ACT_DROPPED).  Requires rafthip_progress_config, rafthip_upload_ids
 * and rafthip_upload_term.  status_counts has RAFTHIP_WIRE_STATUSES entries. */
int rafthip_step_wire(rafthip_engine *e, const uint8_t *bytes, const uint64_t *offsets,
                      const uint32_t *groups, uint64_t n, rafthip_msg *out_msgs, uint8_t *out_status,
                      uint64_t *status_counts, uint8_t *actions);

/* ProgressTracker.Committed for every group — tracker/tracker.go:179-181 =
 * JointConfig.CommittedIndex (quorum/joint.go:49-56) over
 * MajorityConfig.CommittedIndex (quorum/majority.go:126-172) with
 * matchAckIndexer (tracker/tracker.go:164-175).  Result → commit plane. */
int rafthip_commit(rafthip_engine *e);

/* rafthip_commit for n engines — the shards one multi-raft host keeps on one
 * GPU — as ONE kernel launch whose grid spans all of them, so the shards do not
 * each pay a kernel boundary (at 2^20 groups per shard that boundary is a fifth
 * of the pass).  Same results as calling rafthip_commit on each engine
 * (tracker/tracker.go:179-181).  The single launch needs every engine on one
 * GPU and either the same uniform config of 1..8 voters in all of them
 * (rafthip_set_uniform_config), or the same num_slots <= 8 in all of them — then
 * shards with per-group masks (joint configs while a conf change is in flight,
 * quorum/joint.go:49-56) and uniform shards share the launch; any other set of
 * engines is served engine by engine.  Runs on engines[0]'s
 * stream; engines on other streams are joined to it with events, before and
 * after.  Calls on any of the engines must not overlap this one. */
int rafthip_commit_many(rafthip_engine *const *engines, uint32_t n);

/* rafthip_commit_changed (advance == 0) or rafthip_commit_advance (advance != 0)
 * for n engines at once: dst[i] / max_pairs[i] / n_out[i] are engine i's pair
 * buffer (host or device memory; NULL with max_pairs 0 = count only), its room
 * and its count, exactly as in the single-engine calls (consumer: raft.maybeCommit,
 * raft.go:755-758 / raftLog.maybeCommit, log.go:447-456).  Up to 32 engines on one
 * GPU with the same uniform config of 1..8 voters, or with the same num_slots <= 8
 * (masked and uniform shards mixed), take one commit launch, one pack launch and one
 * wait together; any other set is served engine by engine.  Runs on
 * engines[0]'s stream (streams of the other engines that still hold work are joined
 * to it first); waits for the GPU.  An engine may appear only once. */
int rafthip_commit_report_many(rafthip_engine *const *engines, uint32_t n, int advance,
                               rafthip_commit_pair *const *dst, const uint64_t *max_pairs, uint64_t *n_out);

/* Same pass, but also reports the groups whose commit index differs from the
 * commit plane's previous content, sorted by group.  Waits for the GPU.
 * *n_changed = number of pairs; up to max_pairs of them are copied to dst
 * (host or device; may be NULL with max_pairs 0).  Consumer: raft.maybeCommit,
 * raft.go:755-758. */
int rafthip_commit_changed(rafthip_engine *e, rafthip_commit_pair *dst, uint64_t max_pairs,
                           uint64_t *n_changed);
/* raft.maybeCommit for every group — raft.go:755-758: mci = trk.Committed(),
 * then raftLog.maybeCommit(mci, r.Term) (log.go:447-456) fused into the same
 * pass.  The commit plane is raftLog.committed here: it moves to mci, and
 * (group, mci) is reported (sorted by group), iff mci > committed and
 * term_first <= mci <= term_last — i.e. term != 0 and
 * zeroTermOnOutOfBounds(term(mci)) == r.Term.  Only advanced groups travel
 * back, so this replaces the host's per-group term lookup.  Use either
 * rafthip_commit/_changed or this on one engine: they give the commit plane
 * different meanings. */
int rafthip_commit_advance(rafthip_engine *e, rafthip_commit_pair *dst, uint64_t max_pairs,
                           uint64_t *n_advanced);
int rafthip_read_commit(rafthip_engine *e, uint64_t first, uint64_t count, uint64_t *dst);

/* rafthip_commit_changed for callers that cannot allocate device memory: tight
 * serving loops, harnesses that forbid hipMalloc.  (Not for use inside a HIP graph
 * capture: like rafthip_commit_changed it waits for the GPU before it returns, and
 * a wait on a capturing stream invalidates the capture.)
 * No counterpart in the reference.  Same pass and same result as
 * rafthip_commit_changed, but its scratch (ballot words, per-tile counts, the
 * total) lives in a static device array of the library — 2.3 MiB per GPU, held
 * until the process ends — and the call makes no hipMalloc / hipFree or any
 * other allocating call.  Limits: engines of at most RAFTHIP_STATIC_MAX_GROUPS
 * groups (after rounding up to 4096; larger: RAFTHIP_EINVAL); dst must be
 * device memory or page-locked host memory (rafthip_host_alloc), written by the
 * kernel itself; a config must have been set (rafthip_set_uniform_config with
 * voters >= 1, or rafthip_upload_config).  Every call on one GPU shares that
 * GPU's array, which one call holds from its first kernel to its final wait: a call
 * that overlaps another on the same GPU (two engines, two threads) returns
 * RAFTHIP_EBUSY and does nothing; serialise them.  Waits for the GPU like
 * rafthip_commit_changed. */
#define RAFTHIP_STATIC_MAX_GROUPS (1ull << 24)
int rafthip_commit_changed_static(rafthip_engine *e, rafthip_commit_pair *dst, uint64_t max_pairs,
                                  uint64_t *n_changed);
/* bcastHeartbeat as serialized bytes — raft.go:698-716 (bcastHeartbeat,
 * bcastHeartbeatWithCtx), sendHeartbeat raft.go:669-685, r.send raft.go:502-533:
 * for every Progress row (slot s, group first+g) whose id is not 0, except the
 * leader's own (rafthip_upload_self), the message
 *   raftpb.Message{Type: MsgHeartbeat, To: id, From: the id of the self slot,
 *                  Term: r.Term, Commit: min(Match, raftLog.committed),
 *                  Context: the pending ReadIndex context, nil when none}
 * exactly as Message.Marshal writes it (raftpb/raft.pb.go:937-1025), at
 * out_bytes[(s*count + g) * RAFTHIP_HEARTBEAT_STRIDE], zero-padded, its length
 * in out_len[s*count + g] (0 = no message: empty slot, own slot, or a group
 * whose self mask is empty, i.e. not led here).  Both outputs host or device
 * memory (device: out_bytes 8-byte aligned).  The commit plane must hold
 * raftLog.committed (rafthip_commit_advance / rafthip_upload_commit); requires
 * rafthip_upload_ids, _term, _self (and _read_ctx when reads are pending). */
#define RAFTHIP_HEARTBEAT_STRIDE 72
int rafthip_emit_heartbeats(rafthip_engine *e, uint64_t first, uint64_t count, uint8_t *out_bytes,
                            uint8_t *out_len);
/* The same for the groups of the range that stepped MsgBeat in the last
 * rafthip_tick (tickHeartbeat, raft.go:856-861 -> stepLeader MsgBeat,
 * raft.go:1228-1230 -> bcastHeartbeat): every other group gets length 0, as a
 * group that is not led here does.  With HeartbeatTick 1 every led group beats
 * every tick and the two calls agree. */
int rafthip_emit_heartbeats_due(rafthip_engine *e, uint64_t first, uint64_t count, uint8_t *out_bytes,
                                uint8_t *out_len);


/* raft.committedEntryInCurrentTerm for every group — raft.go:2001-2005: has the
 * leader committed an entry of its own term (raftLog.committed inside
 * [term_first, term_last])?  ReadIndex requests wait for it (raft.go:1312-1317,
 * 2062-2071).  Output 1 / 0 per group; the commit plane must hold
 * raftLog.committed (rafthip_commit_advance / rafthip_upload_commit). */
int rafthip_committed_in_term(rafthip_engine *e);
int rafthip_read_committed_in_term(rafthip_engine *e, uint64_t first, uint64_t count, uint8_t *dst);

/* The commit index a heartbeat may carry to each peer — sendHeartbeat,
 * raft.go:669-685: min(Progress.Match, raftLog.committed) for every (slot,
 * group).  dst receives S rows of `count` values (row s at dst + s*dst_stride);
 * with dst NULL the result only stays on the device (RAFTHIP_PLANE_HEARTBEAT). */
int rafthip_heartbeat_commit(rafthip_engine *e, uint64_t first, uint64_t count, uint64_t *dst,
                             uint64_t dst_stride);

/* ProgressTracker.TallyVotes for every group — tracker/tracker.go:269-290 →
 * JointConfig.VoteResult (quorum/joint.go:61-75) over
 * MajorityConfig.VoteResult (quorum/majority.go:178-207). */
int rafthip_tally_votes(rafthip_engine *e);
/* result: VoteResult per group; granted / rejected (either may be NULL):
 * the informational counts of tracker.go:274-287. */
int rafthip_read_vote_result(rafthip_engine *e, uint64_t first, uint64_t count, uint8_t *result,
                             uint8_t *granted, uint8_t *rejected);

/* ProgressTracker.QuorumActive for every group — tracker/tracker.go:217-227
 * (CheckQuorum, raft.go:1231-1243).  Output 1 = active, 0 = not. */
int rafthip_quorum_active(rafthip_engine *e);
int rafthip_read_quorum_active(rafthip_engine *e, uint64_t first, uint64_t count, uint8_t *dst);
/* The whole MsgCheckQuorum step — raft.go:1231-1243: QuorumActive as above,
 * then RecentActive = false for every Progress but the leader's own
 * (rafthip_upload_self).  Read the verdict with rafthip_read_quorum_active. */
int rafthip_check_quorum(rafthip_engine *e);

/* ---- RawNode.Tick for every group — rawnode.go:64-66 ----------------------
 * The caller of the two steps above and of the heartbeat fan-out: r.tick() is
 * tickHeartbeat for a leader (raft.go:835-862) and tickElection for a follower,
 * candidate or pre-candidate (raft.go:823-832).  Per-group state, as planes:
 *   election_elapsed, heartbeat_elapsed   r.electionElapsed, r.heartbeatElapsed
 *   randomized_timeout                    r.randomizedElectionTimeout (raft.go:1988-1990,
 *                                         drawn by the host's generator)
 *   role                                  r.state, StateType of raft.go:48-53
 *                                         (0 follower, 1 candidate, 2 leader, 3 pre-candidate)
 *   flags                                 RAFTHIP_TICKF_PROMOTABLE = r.promotable() (raft.go:1883-1886),
 *                                         RAFTHIP_TICKF_TRANSFER = r.leadTransferee != None
 * and for the engine ElectionTick, HeartbeatTick, CheckQuorum of raft.Config
 * (raft.go:128-138,222-224; refused as Config.validate refuses them, raft.go:296-302).
 * Counters are uint32 where the reference has int: keep them below 2^31.
 *
 * One pass advances every group by one tick and leaves an action byte per group:
 *   RAFTHIP_TICK_HUP             tickElection fired: the host steps MsgHup (raft.go:826-831);
 *                                election_elapsed is back at 0
 *   RAFTHIP_TICK_CHECK_QUORUM    an election timeout passed on a leader with CheckQuorum: the
 *                                MsgCheckQuorum step of raft.go:1231-1243 has been done on the
 *                                group's RecentActive bits (rafthip_upload_active / _self; what
 *                                rafthip_check_quorum does, for this group only)
 *   RAFTHIP_TICK_STEP_DOWN       ... and the quorum was not active: role is follower, both timers 0,
 *                                the transfer flag and every RecentActive bit cleared
 *                                (becomeFollower -> reset, raft.go:864-871,760-789); the host does the
 *                                rest of reset: a new randomized timeout, rafthip_reset_match,
 *                                rafthip_reset_votes, the Progress rows, and clears the group's self
 *                                mask (rafthip_upload_self: the group is no longer led here)
 *   RAFTHIP_TICK_ABORT_TRANSFER  a leader's transfer did not finish within an election timeout
 *                                (raft.go:846-849); the flag is cleared
 *   RAFTHIP_TICK_BEAT            heartbeat_elapsed reached HeartbeatTick on a (still) leader: the host
 *                                steps MsgBeat -> bcastHeartbeat (rafthip_emit_heartbeats_due)
 * counts (may be NULL: nothing is copied back and the call does not wait):
 * counts[b] = groups whose action byte has bit b. */
#define RAFTHIP_TICK_HUP 1u
#define RAFTHIP_TICK_CHECK_QUORUM 2u
#define RAFTHIP_TICK_STEP_DOWN 4u
#define RAFTHIP_TICK_ABORT_TRANSFER 8u
#define RAFTHIP_TICK_BEAT 16u
#define RAFTHIP_TICK_ACTIONS 5
#define RAFTHIP_TICKF_PROMOTABLE 1u
#define RAFTHIP_TICKF_TRANSFER 2u
int rafthip_tick_config(rafthip_engine *e, uint32_t election_tick, uint32_t heartbeat_tick, int check_quorum);
/* Any of the five sources / destinations may be NULL (that plane is left alone);
 * planes never uploaded hold zeros: a follower that is not promotable. */
int rafthip_upload_tick_state(rafthip_engine *e, uint64_t first, uint64_t count, const uint32_t *election_elapsed,
                              const uint32_t *heartbeat_elapsed, const uint32_t *randomized_timeout,
                              const uint8_t *role, const uint8_t *flags);
int rafthip_read_tick_state(rafthip_engine *e, uint64_t first, uint64_t count, uint32_t *election_elapsed,
                            uint32_t *heartbeat_elapsed, uint32_t *randomized_timeout, uint8_t *role,
                            uint8_t *flags);
int rafthip_tick(rafthip_engine *e, uint64_t *counts);
/* The same tick for the shards (engines) of one GPU in ONE launch, for a host that
 * keeps several shards per GPU (as rafthip_commit_many): every shard with its own
 * planes, config and tick constants; counts (may be NULL) receives
 * RAFTHIP_TICK_ACTIONS words per engine.  Engines of one GPU and one mask width
 * share the launch (16 at a time); otherwise the call ticks them one by one.
 * Engines on different streams are joined as in rafthip_commit_many.  An engine
 * may appear in the list once only (its planes are written by the pass). */
int rafthip_tick_many(rafthip_engine *const *engines, uint32_t n, uint64_t *counts);
/* stepFollower's `r.electionElapsed = 0` (raft.go:1681-1692) for the groups that
 * heard from their leader since the last tick — MsgApp, MsgHeartbeat or MsgSnap
 * of the current term: election_elapsed of every listed group becomes 0.  A
 * list (host or device memory), not a plane: on a host that follows millions of
 * groups these are the ids its receive path has in hand anyway.  List followers
 * only: the call resets the counter of whatever group it is given.  Duplicates are
 * fine; ids outside the engine are skipped and counted in *rejected (may be
 * NULL).  Waits for the GPU. */
int rafthip_tick_heard_from_leader(rafthip_engine *e, const uint32_t *groups, uint64_t n, uint64_t *rejected);
int rafthip_read_tick_actions(rafthip_engine *e, uint64_t first, uint64_t count, uint8_t *dst);
/* The groups of the last rafthip_tick whose action byte has a bit of `mask`, in
 * group order, with their action byte: what the host walks instead of the whole
 * action plane (campaigns, step-downs and transfer timeouts are a handful per
 * tick among millions of groups).  At most max_events records are written to dst
 * (host or device memory); *n_events receives how many there are.  Engines of at
 * most 2^32 groups.  Waits for the GPU. */
typedef struct rafthip_tick_event {
    uint32_t group;
    uint32_t action; /* RAFTHIP_TICK_* bits, all of the group's, not only those of mask */
} rafthip_tick_event;
int rafthip_read_tick_events(rafthip_engine *e, uint32_t mask, rafthip_tick_event *dst, uint64_t max_events,
                             uint64_t *n_events);

/* ReadIndex ack quorum — Voters.VoteResult(acks) at raft.go:1552 with the
 * yes-only map of read_only.go:68-76.  Output: VoteResult per group. */
int rafthip_ack_quorum(rafthip_engine *e);
int rafthip_read_ack_quorum(rafthip_engine *e, uint64_t first, uint64_t count, uint8_t *dst);

/* ---- multi-GPU: the path's one exchange step, over RCCL / xGMI ----------- */

/* Groups shard over the GPUs of a node by contiguous group-id range, one
 * engine per GPU; no quorum computation couples two groups, so the only
 * exchange is this one: after the commit pass every rank's changed (or
 * advanced) (group, commit) pairs are all-gathered, so that the host loop that
 * feeds raft.maybeCommit (raft.go:755-758) — and Ready's HardState.Commit
 * (rawnode.go:141-156) — sees the whole node's set from any GPU.  Implemented
 * against librccl: one count all-gather, one padded payload all-gather, one
 * packing kernel.  The payload width of an exchange is guessed from the
 * previous one (half again as much), so in steady state the two collectives
 * and the packing run back to back with no host wait in between; a guess that
 * turns out too small is repeated once, exact.  After rafthip_comm_create an
 * exchange allocates nothing.
 *
 * A communicator binds one engine to one rank.  Ranks may be processes (one
 * per GPU; distribute the id through any channel the host has) or threads of
 * one process (each thread drives its own engine and communicator); a single
 * thread that drives several GPUs uses rafthip_comm_create_all and
 * rafthip_commit_allgather_many, which issue the collectives of all its ranks
 * inside one RCCL group.  Calls on one communicator must be serialised, and
 * every rank must make the same sequence of exchanges (it is a collective). */
typedef struct rafthip_comm rafthip_comm;
typedef struct rafthip_comm_id {
    char bytes[128]; /* an ncclUniqueId */
} rafthip_comm_id;

/* On one rank; hand the id to the others. */
int rafthip_comm_unique_id(rafthip_comm_id *id);
/* Collective over the nranks ranks.  group_base = first global group id of this
 * engine's shard (ranks own ascending ranges, so the gathered pairs come out
 * sorted by global group id).  max_pairs_per_rank = most pairs one rank may
 * contribute to an exchange, 0 = this engine's group count (the worst case);
 * every rank ends up with the largest value any rank asked for.  Device memory:
 * (2 * nranks + 1) * that many pairs of 16 bytes. */
int rafthip_comm_create(rafthip_engine *e, const rafthip_comm_id *id, int rank, int nranks, uint64_t group_base,
                        uint64_t max_pairs_per_rank, rafthip_comm **out);
/* The same for n engines on n different GPUs of this process (rank i = engines[i]). */
int rafthip_comm_create_all(rafthip_engine **engines, int n, const uint64_t *group_base, uint64_t max_pairs_per_rank,
                            rafthip_comm **out);
int rafthip_comm_destroy(rafthip_comm *c);
/* *nranks is RCCL's own count of the communicator's ranks (ncclCommCount). */
int rafthip_comm_info(const rafthip_comm *c, int *rank, int *nranks, uint64_t *max_pairs_per_rank);

/* rafthip_commit_changed (advance == 0) or rafthip_commit_advance (advance != 0)
 * on this rank's engine, then the all-gather: dst receives up to max_pairs pairs
 * of the WHOLE node, global group ids, sorted by group (host or device memory;
 * NULL with max_pairs 0 = counts only); counts[nranks] (may be NULL) = pairs each
 * rank reported; *n_total = their sum.  A rank that reports more than
 * max_pairs_per_rank contributes only that many pairs (its count still says how
 * many there were).  Waits for the GPU.  Consumer: raft.maybeCommit,
 * raft.go:755-758. */
int rafthip_commit_allgather(rafthip_comm *c, int advance, rafthip_commit_pair *dst, uint64_t max_pairs,
                             uint64_t *counts, uint64_t *n_total);
/* The same exchange for all n ranks that one thread drives: dst[i], max_pairs[i],
 * counts[i], n_total[i] belong to comms[i] (dst / counts / their entries may be
 * NULL). */
int rafthip_commit_allgather_many(rafthip_comm **comms, int n, int advance, rafthip_commit_pair **dst,
                                  const uint64_t *max_pairs, uint64_t **counts, uint64_t *n_total);

/* The dense form of the same exchange, for ticks in which most groups move (a pair
 * is 16 bytes, a plane entry 8): rafthip_commit (advance == 0) or the term-guarded
 * pass of rafthip_commit_advance (advance != 0) on this rank's engine, then ONE
 * all-gather of the commit planes: dst[r * width + g] = the commit index (advance:
 * raftLog.committed) of rank r's group g.  *width (may be NULL) = the largest plane
 * stride of any rank, the same on every rank — rafthip_plane(e, RAFTHIP_PLANE_COMMIT)
 * reports a rank's own; entries at g >= a rank's group count carry no meaning.  dst
 * (host or device memory) holds dst_elems >= nranks * width elements; with dst NULL
 * the call only reports *width (local, no collective).  Waits for the GPU.  Consumer: raft.maybeCommit, raft.go:755-758, diffing against the commit
 * indexes it already holds; SURVEY.md 8(e) "dense fallback". */
int rafthip_commit_allgather_dense(rafthip_comm *c, int advance, uint64_t *dst, uint64_t dst_elems, uint64_t *width);
/* ... and for all n ranks that one thread drives (dst[i], dst_elems[i], width[i] belong to comms[i]). */
int rafthip_commit_allgather_dense_many(rafthip_comm **comms, int n, int advance, uint64_t **dst,
                                        const uint64_t *dst_elems, uint64_t *width);

/* ---- zero-copy access for hosts that fill planes on the GPU themselves --- */

#define RAFTHIP_PLANE_MATCH 0  /* uint64 [S][stride] */
#define RAFTHIP_PLANE_COMMIT 1 /* uint64 [stride]    */
#define RAFTHIP_PLANE_VOTES 2  /* uint8  [S][stride] */
#define RAFTHIP_PLANE_HEARTBEAT 3 /* uint64 [S][stride], filled by rafthip_heartbeat_commit */
/* Device address and plane stride (in elements) of an engine plane. */
int rafthip_plane(rafthip_engine *e, int which, void **ptr, uint64_t *stride);

/* ---- stateless one-shot forms (host or device pointers) ----------------- */

/* What the new Go entry point quorum.BatchCommittedIndex binds to: for each of
 * G groups, JointConfig{in,out}.CommittedIndex(l) (quorum/joint.go:49-56) with
 * l.AckedIndex(slot) = (match[slot][g], true); pass 0 for a voter the indexer
 * does not know (quorum/majority.go:149-161 leaves it 0).  Mask width as
 * rafthip_mask_bytes for S; out_mask may be NULL (non-joint).
 * Host pointers: the planes are staged chunk by chunk through page-locked buffers
 * that the library keeps between calls (with three staging threads and one result
 * thread, parked while idle), the kernels read them over the link, the results are
 * copied to `out`; nothing is allocated per call after the first of a size, and
 * calls from several threads run side by side.  Returns when `out` is complete.
 * Device pointers: evaluated in place through an engine that lives for the call. */
int rafthip_batch_committed_index(int device, uint64_t num_groups, uint32_t num_slots,
                                  const uint64_t *match, const void *in_mask,
                                  const void *out_mask, uint64_t *out);
/* quorum.BatchVoteResult: JointConfig{in,out}.VoteResult(votes) per group
 * (quorum/joint.go:61-75); votes are vote-cell planes [S][G]. */
int rafthip_batch_vote_result(int device, uint64_t num_groups, uint32_t num_slots,
                              const uint8_t *votes, const void *in_mask, const void *out_mask,
                              uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif /* RAFTHIP_H */
