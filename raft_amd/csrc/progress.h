// progress.h — K9: the leader's per-peer replication state on the device
// (SURVEY.md §8f row 3): tracker.Progress (tracker/progress.go), its Inflights
// window (tracker/inflights.go) and the stepLeader branches that drive them
// (raft.go:1333-1586), for a stream of events.
//
// Unlike Match = max(...) these updates do not commute: a rejection followed
// by an ack is not an ack followed by a rejection.  Events of one batch are
// therefore applied in stream order per Progress, by exactly one lane:
//
//   k_progress_link   every event swaps its index into its Progress record's
//                     head word (one 32-bit atomic exchange) and keeps what it
//                     found in prev[i]: a linked list per Progress, in whatever
//                     order the exchanges ran.  Leader-log growth records (kMsgAppended)
//                     are applied here, i.e. before every response.
//   k_progress_apply  the event that still owns the head word (the last to
//                     exchange) walks the list, takes the events in ascending
//                     stream index and steps them through the state machine
//                     with the Progress in registers; plain loads / stores, no
//                     other lane touches the cell.  Lists longer than
//                     kStepLimit are left unstepped (action DEFERRED).
//
// Different Progress cells never interact on this path (the group-level
// consequences — maybeCommit, bcastAppend — are the commit pass and the host's
// sends), so the grid is embarrassingly parallel across cells.
#pragma once
#include "kernels.h"

namespace rafthip {

constexpr u32 kStepLimit = 64;  // events per Progress per batch that are stepped (= RAFTHIP_STEP_LIMIT)

enum : uint8_t {  // tracker/state.go:18-34
    kStateProbe = 0,
    kStateReplicate = 1,
    kStateSnapshot = 2,
};
enum : uint8_t {  // raftpb/raft.proto:41-69 + the local records
    kEvUnreachable = 10,
    kEvSnapStatus = 11,
    kEvSentApp = 0xF1,
    kEvSentSnap = 0xF2,
    kEvNeedsHostRec = 0xFE,  // = kEvNeedsHost of wire.h: a response only the host can step
};
enum : uint8_t {  // = RAFTHIP_ACT_*
    kActSendAppend = 1,
    kActUpdated = 2,
    kActWasPaused = 4,
    kActCaughtUp = 8,
    kActError = 32,
    kActDeferred = 64,
    kActDropped = 128,
};

// Everything of a Progress but Match (which the commit kernels stream as a
// plane) in ONE 32-byte record, so a step costs one line read and one line
// write instead of six of each: the apply pass is bound by the number of
// scattered lines it touches, not by bytes.  The list head of the link pass
// lives in the same record.
struct alignas(32) ProgRec {
    u64 next;
    u64 pending;  // PendingSnapshot
    u64 ibytes;   // Inflights.bytes
    u32 meta;     // Inflights.start (bits 0-13) | Inflights.count (14-28) | State (29-30) | MsgAppFlowPaused (31)
    u32 head;     // k_progress_link / k_progress_apply; zero between calls
};
constexpr u32 kMaxInflight = 16383;
__host__ __device__ __forceinline__ u32 rec_meta(u32 start, u32 count, u32 pstate) {
    return (start & 0x3FFFu) | ((count & 0x7FFFu) << 14) | (pstate << 29);
}

// One entry of an Inflights ring: inflight{index, bytes} (tracker/inflights.go:18-21).
struct RingEntry {
    u64 index, bytes;
};
struct ProgPlanes {
    u64* match;        // [S][stride]
    ProgRec* rec;      // [S][stride]
    // Inflights.buffer of every Progress: [S][stride][max(cap, 1)] entries, the ring of one Progress contiguous.
    // (Rounds 1-2 kept two planes [entry][slot][group]: FreeLE's walk then read one 8-byte word per line and entry —
    // five lines per ack at cap 4, a TLB entry apart at larger caps; now it is one 16-byte load per entry from one
    // or a few adjacent lines.  The C ABI still speaks [entry][slot][group] planes: k_ring_pack / k_ring_unpack.)
    RingEntry* ring;
    u64 stride;
    u32 cap;           // Inflights.size
    u32 ring_len;      // max(cap, 1): entries per Progress in `ring`
    u64 max_bytes;     // Inflights.maxBytes
};

struct Prog {  // one tracker.Progress in registers
    u64 match, next, pending, ibytes;
    u32 state, paused, istart, icount;
    bool active, read_ack;
};

__device__ __forceinline__ bool infl_full(const Prog& p, const ProgPlanes& pl) {  // inflights.go:131-133
    return p.icount == pl.cap || (pl.max_bytes != 0 && p.ibytes >= pl.max_bytes);
}
__device__ __forceinline__ void prog_reset_state(Prog& p, u32 state) {  // progress.go:102-107
    p.paused = 0;
    p.pending = 0;
    p.state = state;
    p.istart = p.icount = 0;
    p.ibytes = 0;
}
__device__ __forceinline__ u64 umax(u64 a, u64 b) { return a > b ? a : b; }
__device__ __forceinline__ u64 umin(u64 a, u64 b) { return a < b ? a : b; }
__device__ __forceinline__ void prog_become_probe(Prog& p) {  // progress.go:111-123
    if (p.state == kStateSnapshot) {
        const u64 pending = p.pending;
        prog_reset_state(p, kStateProbe);
        p.next = umax(p.match + 1, pending + 1);
    } else {
        prog_reset_state(p, kStateProbe);
        p.next = p.match + 1;
    }
}
__device__ __forceinline__ void prog_become_replicate(Prog& p) {  // progress.go:126-129
    prog_reset_state(p, kStateReplicate);
    p.next = p.match + 1;
}
__device__ __forceinline__ bool prog_is_paused(const Prog& p) {  // progress.go:225-236
    return p.state == kStateSnapshot ? true : p.paused != 0;
}

__device__ __forceinline__ RingEntry ring_load(const RingEntry* p) {
    const Vec<u64, 2> v = load_vec<u64, 2>(reinterpret_cast<const u64*>(p));
    return RingEntry{v.v[0], v.v[1]};
}

// Inflights.FreeLE, inflights.go:98-128, on the cell's ring in global memory.
__device__ __forceinline__ void infl_free_le(Prog& p, const ProgPlanes& pl, u64 cell, u64 to) {
    if (p.icount == 0) return;
    const RingEntry* ring = pl.ring + cell * pl.ring_len;
    u32 idx = p.istart, i = 0;
    u64 bytes = 0;
    for (; i < p.icount; ++i) {
        const RingEntry en = ring_load(ring + idx);
        if (to < en.index) break;
        bytes += en.bytes;
        if (++idx >= pl.cap) idx -= pl.cap;
    }
    if (i == 0) return;  // out of the left side of the window (inflights.go:99-102)
    p.icount -= i;
    p.ibytes -= bytes;
    p.istart = idx;
    if (p.icount == 0) p.istart = 0;
}

// One event on one Progress: stepLeader, raft.go:1333-1586, and the send-side
// bookkeeping of maybeSendAppend, raft.go:645,653.  Returns the action byte.
__device__ __forceinline__ uint8_t prog_step(Prog& p, const ProgPlanes& pl, u64 cell, u32 type, bool reject, u32 flags,
                                             u64 index, u64 aux, const u64* last_index, const u64* first_index) {
    uint8_t act = 0;
    switch (type) {
        case kMsgAppResp:
            p.active = true;  // :1337
            if (reject) {
                // MaybeDecrTo(m.Index, nextProbeIdx), progress.go:194-217
                bool decr;
                if (p.state == kStateReplicate) {
                    decr = index > p.match;
                    if (decr) p.next = p.match + 1;
                } else {
                    decr = p.next - 1 == index;
                    if (decr) {
                        p.next = umax(umin(index, aux + 1), 1);
                        p.paused = 0;
                    }
                }
                if (decr) {  // :1460-1466
                    if (p.state == kStateReplicate) prog_become_probe(p);
                    act |= kActSendAppend;
                }
            } else {
                const bool old_paused = prog_is_paused(p);  // :1468
                bool updated = false;                       // MaybeUpdate, progress.go:168-177
                if (p.match < index) {
                    p.match = index;
                    updated = true;
                    p.paused = 0;
                }
                p.next = umax(p.next, index + 1);
                if (updated || (p.match == index && p.state == kStateProbe)) {  // :1477
                    if (p.state == kStateProbe) {
                        prog_become_replicate(p);
                    } else if (p.state == kStateSnapshot && p.match + 1 >= (first_index ? *first_index : 0)) {
                        prog_become_probe(p);
                        prog_become_replicate(p);
                    } else if (p.state == kStateReplicate) {
                        infl_free_le(p, pl, cell, index);
                    }
                    act |= kActUpdated;
                    if (old_paused) act |= kActWasPaused;
                    if (p.match == *last_index) act |= kActCaughtUp;
                }
            }
            break;
        case kMsgHeartbeatResp:  // :1527-1561
            p.active = true;
            p.paused = 0;
            if (p.state == kStateProbe || p.match < *last_index) act |= kActSendAppend;
            if (flags & kMsgFlagCtx) p.read_ack = true;
            break;
        case kEvSnapStatus:  // :1562-1579
            if (p.state != kStateSnapshot) break;
            if (reject) p.pending = 0;
            prog_become_probe(p);
            p.paused = 1;
            break;
        case kEvUnreachable:  // :1580-1586
            if (p.state == kStateReplicate) prog_become_probe(p);
            break;
        case kEvSentApp: {  // UpdateOnEntriesSend, progress.go:141-163; index = last entry sent
            const bool entries = (flags & 1u) != 0;
            if (p.state == kStateReplicate) {
                if (entries) {
                    if (infl_full(p, pl)) {  // Inflights.Add panics, inflights.go:66-68
                        act |= kActError;
                        break;
                    }
                    p.next = index + 1;  // OptimisticUpdate
                    u32 pos = p.istart + p.icount;
                    if (pos >= pl.cap) pos -= pl.cap;
                    Vec<u64, 2> en;
                    en.v[0] = index;
                    en.v[1] = aux;
                    store_vec<u64, 2>(reinterpret_cast<u64*>(pl.ring + cell * pl.ring_len + pos), en);
                    p.icount++;
                    p.ibytes += aux;
                }
                p.paused = infl_full(p, pl) ? 1u : 0u;
            } else if (p.state == kStateProbe) {
                if (entries) p.paused = 1;
            } else {
                act |= kActError;  // "sending append in unhandled state", progress.go:160
            }
            break;
        }
        case kEvSentSnap:  // BecomeSnapshot, progress.go:133-136
            prog_reset_state(p, kStateSnapshot);
            p.pending = index;
            break;
        default: break;
    }
    return act;
}

__device__ __forceinline__ bool prog_event_type(u32 type) {
    return type == kMsgAppResp || type == kMsgHeartbeatResp || type == kEvUnreachable || type == kEvSnapStatus ||
           type == kEvSentApp || type == kEvSentSnap || type == kEvNeedsHostRec;
}

struct MsgFields {
    u32 group, slot, type, flags;
    bool reject;
    u64 index;
};
__device__ __forceinline__ MsgFields load_msg(const Msg* msgs, u64 i) {
    const Vec<u64, 2> raw = load_vec<u64, 2>(reinterpret_cast<const u64*>(msgs + i));
    MsgFields f;
    f.group = (u32)raw.v[0];
    f.slot = (u32)(raw.v[0] >> 32) & 0xFFu;
    f.type = (u32)(raw.v[0] >> 40) & 0xFFu;
    f.reject = ((raw.v[0] >> 48) & 0xFFu) != 0;
    f.flags = (u32)(raw.v[0] >> 56);
    f.index = raw.v[1];
    return f;
}

__device__ __forceinline__ void progress_link_one(u32 i, const Msg* __restrict__ msgs, u64 G, u32 S, u64 stride,
                                                  ProgRec* __restrict__ rec, u32* __restrict__ prev,
                                                  u64* __restrict__ term_last, uint8_t* __restrict__ actions) {
    const MsgFields f = load_msg(msgs, i);
    uint8_t act;
    if (f.type == kMsgAppended) {
        if ((u64)f.group < G) {
            atomicMax(term_last + f.group, f.index);
            act = 0;
        } else {
            act = kActDropped;
        }
    } else if (!prog_event_type(f.type) || (u64)f.group >= G || f.slot >= S) {
        act = kActDropped;
    } else {
        prev[i] = atomicExch(&rec[(u64)f.slot * stride + f.group].head, i + 1);
        act = kActDeferred;
    }
    actions[i] = act;
}

__global__ __launch_bounds__(kBlock) void k_progress_link(const Msg* __restrict__ msgs, u32 n, u64 G, u32 S,
                                                          u64 stride, ProgRec* __restrict__ rec,
                                                          u32* __restrict__ prev, u64* __restrict__ term_last,
                                                          uint8_t* __restrict__ actions) {
    const u32 i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) progress_link_one(i, msgs, G, S, stride, rec, prev, term_last, actions);
}

// K9p's fallback (progress_part.h): the same pass behind a device flag, as a grid-stride grid — when the flag is
// clear, the usual case, a modest grid leaves at once (one workgroup per 256 events: 17 us of empty launches at 2^24)
constexpr u32 kProgFlaggedGrid = 8192;
__global__ __launch_bounds__(kBlock) void k_progress_link_flagged(const Msg* __restrict__ msgs, u32 n, u64 G, u32 S,
                                                                  u64 stride, ProgRec* __restrict__ rec,
                                                                  u32* __restrict__ prev, u64* __restrict__ term_last,
                                                                  uint8_t* __restrict__ actions,
                                                                  const u32* __restrict__ flag) {
    if (!flag[0]) return;
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < n; i += (u64)gridDim.x * kBlock)
        progress_link_one((u32)i, msgs, G, S, stride, rec, prev, term_last, actions);
}

__device__ __forceinline__ void progress_apply_one(u32 i, const Msg* __restrict__ msgs, const u64* __restrict__ aux, u32 n,
                                                   u64 G, u32 S, const ProgPlanes& pl, const u32* __restrict__ prev,
                                                   void* __restrict__ active, void* __restrict__ acks, int mbytes,
                                                   const u64* __restrict__ term_last, const u64* __restrict__ first_index,
                                                   uint8_t* __restrict__ actions) {
    const MsgFields own = load_msg(msgs, i);
    if (!prog_event_type(own.type) || (u64)own.group >= G || own.slot >= S) return;
    const u64 cell = (u64)own.slot * pl.stride + own.group;
    u64* rw = reinterpret_cast<u64*>(pl.rec + cell);
    const Vec<u64, 2> r0 = load_vec<u64, 2>(rw), r1 = load_vec<u64, 2>(rw + 2);
    if ((u32)(r1.v[1] >> 32) != i + 1) return;  // another event of this Progress exchanged later: it owns the list

    // (a call that fails between the two passes marks the engine; the next call clears every head word first, so
    // no link of an earlier batch is ever met here; links above n are still treated as the end)
    u32 len = 0;
    for (u32 j = i + 1; j != 0 && j <= n && len <= kStepLimit; j = prev[j - 1]) ++len;
    if (len > kStepLimit) {  // all of them stay DEFERRED
        pl.rec[cell].head = 0;
        return;
    }
    Prog p;
    const u64 match0 = pl.match[cell];
    const u32 meta = (u32)r1.v[1];
    p.match = match0;
    p.next = r0.v[0];
    p.pending = r0.v[1];
    p.ibytes = r1.v[0];
    p.state = (meta >> 29) & 3u;
    p.paused = meta >> 31;
    p.istart = meta & 0x3FFFu;
    p.icount = (meta >> 14) & 0x7FFFu;
    p.active = p.read_ack = false;
    const u64* last_index = term_last + own.group;
    const u64* first = first_index ? first_index + own.group : nullptr;

    u32 done = 0;  // stream index + 1 of the last event stepped
    bool blocked = false;  // a host-only event was met: it and what follows it on this Progress stay DEFERRED
    for (u32 k = 0; k < len; ++k) {
        u32 cur = i + 1;
        if (len > 1) {  // the smallest index above `done`: lists are short, a re-walk beats a sort
            cur = 0xFFFFFFFFu;
            u32 j = i + 1;
            for (u32 c = 0; c < len; ++c, j = prev[j - 1])  // exactly the len links counted above
                if (j > done && j < cur) cur = j;
        }
        const MsgFields f = cur == i + 1 ? own : load_msg(msgs, cur - 1);
        blocked = blocked || f.type == kEvNeedsHostRec;
        if (!blocked)
            actions[cur - 1] = prog_step(p, pl, cell, f.type, f.reject, f.flags, f.index, aux ? aux[cur - 1] : 0,
                                         last_index, first);
        done = cur;
    }
    if (p.match != match0) pl.match[cell] = p.match;
    Vec<u64, 2> w0, w1;
    w0.v[0] = p.next;
    w0.v[1] = p.pending;
    w1.v[0] = p.ibytes;
    w1.v[1] = (u64)rec_meta(p.istart, p.icount, p.state | (p.paused << 2));  // head = 0
    store_vec<u64, 2>(rw, w0);
    store_vec<u64, 2>(rw + 2, w1);
    if (p.active) or_slot_bit(active, mbytes, own.group, own.slot);
    if (p.read_ack) or_slot_bit(acks, mbytes, own.group, own.slot);
}

__global__ __launch_bounds__(kBlock) void k_progress_apply(const Msg* __restrict__ msgs,
                                                           const u64* __restrict__ aux, u32 n, u64 G, u32 S,
                                                           ProgPlanes pl, const u32* __restrict__ prev,
                                                           void* __restrict__ active, void* __restrict__ acks,
                                                           int mbytes, const u64* __restrict__ term_last,
                                                           const u64* __restrict__ first_index,
                                                           uint8_t* __restrict__ actions) {
    const u32 i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) progress_apply_one(i, msgs, aux, n, G, S, pl, prev, active, acks, mbytes, term_last, first_index, actions);
}

__global__ __launch_bounds__(kBlock) void k_progress_apply_flagged(const Msg* __restrict__ msgs,
                                                                   const u64* __restrict__ aux, u32 n, u64 G, u32 S,
                                                                   ProgPlanes pl, const u32* __restrict__ prev,
                                                                   void* __restrict__ active, void* __restrict__ acks,
                                                                   int mbytes, const u64* __restrict__ term_last,
                                                                   const u64* __restrict__ first_index,
                                                                   uint8_t* __restrict__ actions,
                                                                   const u32* __restrict__ flag) {
    if (!flag[0]) return;
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < n; i += (u64)gridDim.x * kBlock)
        progress_apply_one((u32)i, msgs, aux, n, G, S, pl, prev, active, acks, mbytes, term_last, first_index, actions);
}

// K9d — the same stepping when a tick's responses are dense: an ack plane
// (acks[slot][group] = acknowledged index + 1, 0 = no MsgAppResp from that peer)
// and a heartbeat-response bitmask per group.  One event of each kind per
// Progress at most, so there is nothing to order between lanes: lane = group,
// loop over slots, every access a coalesced stream (32-byte records, Match, ack
// and ring planes).  Per Progress: the MsgAppResp first, then the
// MsgHeartbeatResp.  A wave writes a slot's records back only if one of its
// lanes stepped (whole-line stores).  out[slot][group] = the two action bytes
// or-ed (they share no bit: an accepted ack never asks for sendAppend, a
// heartbeat response asks for nothing else).
template <typename MaskT>
__global__ __launch_bounds__(kBlock) void k_progress_dense(ProgPlanes pl, u64 first, u64 count, u32 S,
                                                           const u64* __restrict__ acks_plane, u64 ack_stride,
                                                           const MaskT* __restrict__ hb, const MaskT* __restrict__ hb_ctx,
                                                           MaskT* __restrict__ active, MaskT* __restrict__ rd_acks,
                                                           const u64* __restrict__ term_last,
                                                           const u64* __restrict__ first_index,
                                                           uint8_t* __restrict__ out, u64 out_stride) {
    const u64 i = (u64)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < count;
    const u64 g = first + (live ? i : 0);
    const u64 hbm = live && hb ? (u64)hb[i] : 0, ctxm = live && hb_ctx ? (u64)hb_ctx[i] : 0;
    u64 act_bits = 0, ack_bits = 0;
    for (u32 s = 0; s < S; ++s) {
        const u64 cell = (u64)s * pl.stride + g;
        const u64 av = live && acks_plane ? acks_plane[(u64)s * ack_stride + i] : 0;
        const bool has_ack = av != 0, has_hb = ((hbm >> s) & 1ull) != 0;
        const bool any = has_ack || has_hb;
        if (!__any(any)) {
            if (live && out) out[(u64)s * out_stride + i] = 0;
            continue;
        }
        u64* rw = reinterpret_cast<u64*>(pl.rec + cell);
        Vec<u64, 2> r0, r1;
        if (live) {
            r0 = load_vec<u64, 2>(rw);
            r1 = load_vec<u64, 2>(rw + 2);
        }
        uint8_t act = 0;
        if (any) {
            Prog p;
            const u64 match0 = pl.match[cell];
            const u32 meta = (u32)r1.v[1];
            p.match = match0;
            p.next = r0.v[0];
            p.pending = r0.v[1];
            p.ibytes = r1.v[0];
            p.state = (meta >> 29) & 3u;
            p.paused = meta >> 31;
            p.istart = meta & 0x3FFFu;
            p.icount = (meta >> 14) & 0x7FFFu;
            p.active = p.read_ack = false;
            const u64* fi = first_index ? first_index + g : nullptr;
            if (has_ack) act |= prog_step(p, pl, cell, kMsgAppResp, false, 0, av - 1, 0, term_last + g, fi);
            if (has_hb)
                act |= prog_step(p, pl, cell, kMsgHeartbeatResp, false, (u32)((ctxm >> s) & 1ull), 0, 0, term_last + g, fi);
            if (p.match != match0) pl.match[cell] = p.match;
            r0.v[0] = p.next;
            r0.v[1] = p.pending;
            r1.v[0] = p.ibytes;
            r1.v[1] = (u64)rec_meta(p.istart, p.icount, p.state | (p.paused << 2)) | (r1.v[1] & 0xFFFFFFFF00000000ull);
            act_bits |= (u64)p.active << s;
            ack_bits |= (u64)p.read_ack << s;
        }
        if (live) {
            store_vec<u64, 2>(rw, r0);
            store_vec<u64, 2>(rw + 2, r1);
            if (out) out[(u64)s * out_stride + i] = act;
        }
    }
    if (live) {
        if (act_bits) active[g] = (MaskT)(active[g] | (MaskT)act_bits);
        if (ack_bits) rd_acks[g] = (MaskT)(rd_acks[g] | (MaskT)ack_bits);
    }
}

// ---------------------------------------------------------------------------
// K11 — bcastAppend as a plan (raft.go:689-696): which peers of a led group get
// an append now, and of which kind — maybeSendAppend's decision
// (raft.go:600-666) without building the message:
//   IsPaused()                                     -> none      (:602-604)
//   errt = term(Next-1) fails: compacted / beyond lastIndex      (:606-607, log.go:381-397)
//   entries wanted unless a throttled StateReplicate (:617-619); raftLog.entries(Next):
//       Next > lastIndex -> none, no error; Next < firstIndex -> ErrCompacted (log.go:409-414,543-557)
//   no entries and !sendIfEmpty                    -> none      (:621-623)
//   errt or erre: a snapshot, if RecentActive      -> snapshot / none  (:625-650)
//   otherwise a MsgApp, with entries or empty      -> append / empty   (:652-665)
// Lane = group, loop over slots; records, first / last index and the bit
// planes are coalesced streams.  The leader's own row, rows outside the config
// and groups without a self bit (not led here) get none.
// ---------------------------------------------------------------------------
enum : uint8_t { kSendNone = 0, kSendAppend = 1, kSendEmpty = 2, kSendSnapshot = 3 };  // = RAFTHIP_SEND_*

__device__ __forceinline__ uint8_t plan_append(u64 next, u32 state, bool paused, bool full, bool recent_active,
                                               u64 first_index, u64 last_index, bool send_if_empty) {
    if (state == kStateSnapshot || paused) return kSendNone;  // IsPaused, progress.go:225-236
    const u64 li = next - 1;
    const bool errt = li + 1 < first_index || li > last_index;
    bool have = false, erre = false;
    if (state != kStateReplicate || !full) {
        if (next > last_index) {
        } else if (next < first_index) {
            erre = true;
        } else {
            have = true;
        }
    }
    if (!have && !send_if_empty) return kSendNone;
    if (errt || erre) return recent_active ? kSendSnapshot : kSendNone;
    return have ? kSendAppend : kSendEmpty;
}

template <typename MaskT>
__global__ __launch_bounds__(kBlock) void k_plan_appends(ProgPlanes pl, u64 first, u64 count, u32 S,
                                                         const MaskT* __restrict__ in_mask /* NULL: every slot holds a row */,
                                                         const MaskT* __restrict__ out_mask,
                                                         const MaskT* __restrict__ learner_mask /* may be NULL */,
                                                         const MaskT* __restrict__ self_mask,
                                                         const MaskT* __restrict__ active,
                                                         const u64* __restrict__ first_index,
                                                         const u64* __restrict__ last_index, bool send_if_empty,
                                                         uint8_t* __restrict__ plan, u64* __restrict__ next_out,
                                                         u64 out_stride) {
    const u64 i = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const u64 g = first + i;
    const u64 self = (u64)self_mask[g];
    const u64 have_row = in_mask ? ((u64)in_mask[g] | (u64)out_mask[g] | (learner_mask ? (u64)learner_mask[g] : 0)) : ~0ull;
    const u64 act = (u64)active[g];
    const u64 fi = first_index[g], la = last_index[g];
    for (u32 s = 0; s < S; ++s) {
        const u64* rw = reinterpret_cast<const u64*>(pl.rec + (u64)s * pl.stride + g);
        const Vec<u64, 2> r0 = load_vec<u64, 2>(rw), r1 = load_vec<u64, 2>(rw + 2);
        const u32 meta = (u32)r1.v[1];
        const u32 icount = (meta >> 14) & 0x7FFFu;
        const bool full = icount == pl.cap || (pl.max_bytes != 0 && r1.v[0] >= pl.max_bytes);  // inflights.go:131-133
        uint8_t out = kSendNone;
        if (self != 0 && !((self >> s) & 1ull) && ((have_row >> s) & 1ull))
            out = plan_append(r0.v[0], (meta >> 29) & 3u, (meta >> 31) != 0, full, ((act >> s) & 1ull) != 0, fi, la,
                              send_if_empty);
        plan[(u64)s * out_stride + i] = out;
        if (next_out) next_out[(u64)s * out_stride + i] = r0.v[0];
    }
}

// ---- SoA <-> record conversion for the upload / read entry points -----------
// (the C ABI speaks [slot][group] planes like every other call)

// (Re)initialise records of groups [first, first+count): State / paused / Next /
// PendingSnapshot from planes of row pitch `count`, fresh Inflights
// (Progress.ResetState, progress.go:102-107).
// Head words of every Progress record back to 0: run before a link pass when an earlier call failed
// between its link and apply passes and may have left list heads behind (rafthip_engine::heads_dirty).
__global__ __launch_bounds__(kBlock) void k_progress_clear_heads(ProgRec* __restrict__ rec, u64 cells) {
    const u64 i = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (i < cells) rec[i].head = 0;
}

// One Progress row as confchange's initProgress leaves it (confchange/confchange.go:257-270): the given
// State / Next / PendingSnapshot, an empty Inflights; every other row of the group untouched.
__global__ void k_progress_init_row(ProgRec* __restrict__ rec, u64 stride, u64 group, u32 slot, u32 pstate, u64 next,
                                    u64 pending) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    ProgRec r;
    r.next = next;
    r.pending = pending;
    r.ibytes = 0;
    r.meta = rec_meta(0, 0, pstate & 7u);
    r.head = 0;
    rec[(u64)slot * stride + group] = r;
}

__global__ __launch_bounds__(kBlock) void k_progress_pack(ProgRec* __restrict__ rec, u64 stride, u64 first, u64 count,
                                                          u32 S, const uint8_t* __restrict__ pstate,
                                                          const u64* __restrict__ next,
                                                          const u64* __restrict__ pending) {
    const u64 g = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (g >= count) return;
    for (u32 s = 0; s < S; ++s) {
        ProgRec r;
        r.next = next[(u64)s * count + g];
        r.pending = pending ? pending[(u64)s * count + g] : 0;
        r.ibytes = 0;
        r.meta = rec_meta(0, 0, pstate[(u64)s * count + g] & 7u);
        r.head = 0;
        rec[(u64)s * stride + first + g] = r;
    }
}

// Inflights.start / count / bytes of a range from planes (meta = start | count << 16).
__global__ __launch_bounds__(kBlock) void k_progress_set_inflights(ProgRec* __restrict__ rec, u64 stride, u64 first,
                                                                   u64 count, u32 S, const u32* __restrict__ meta,
                                                                   const u64* __restrict__ bytes) {
    const u64 g = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (g >= count) return;
    for (u32 s = 0; s < S; ++s) {
        ProgRec& r = rec[(u64)s * stride + first + g];
        const u32 m = meta[(u64)s * count + g];
        r.meta = rec_meta(m & 0xFFFFu, m >> 16, r.meta >> 29);
        r.ibytes = bytes[(u64)s * count + g];
    }
}

__global__ __launch_bounds__(kBlock) void k_progress_unpack(const ProgRec* __restrict__ rec, u64 stride, u64 first,
                                                            u64 count, u32 S, uint8_t* __restrict__ pstate,
                                                            u64* __restrict__ next, u64* __restrict__ pending,
                                                            u32* __restrict__ meta, u64* __restrict__ bytes) {
    const u64 g = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (g >= count) return;
    for (u32 s = 0; s < S; ++s) {
        const ProgRec r = rec[(u64)s * stride + first + g];
        const u64 o = (u64)s * count + g;
        pstate[o] = (uint8_t)(r.meta >> 29);
        next[o] = r.next;
        pending[o] = r.pending;
        meta[o] = (r.meta & 0x3FFFu) | (((r.meta >> 14) & 0x7FFFu) << 16);
        bytes[o] = r.ibytes;
    }
}

// Inflights rings between the ABI's [entry][slot][group] planes (row pitch `pitch`, group 0 = `first`) and the
// engine's [slot][group][entry] records.
__global__ __launch_bounds__(kBlock) void k_ring_pack(RingEntry* __restrict__ ring, u64 stride, u64 first, u64 count, u32 S,
                                                      u32 ring_len, const u64* __restrict__ index,
                                                      const u64* __restrict__ bytes, u64 pitch) {
    const u64 g = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (g >= count) return;
    for (u32 s = 0; s < S; ++s)
        for (u32 k = 0; k < ring_len; ++k) {
            Vec<u64, 2> en;
            en.v[0] = index[((u64)k * S + s) * pitch + g];
            en.v[1] = bytes[((u64)k * S + s) * pitch + g];
            store_vec<u64, 2>(reinterpret_cast<u64*>(ring + ((u64)s * stride + first + g) * ring_len + k), en);
        }
}
__global__ __launch_bounds__(kBlock) void k_ring_unpack(const RingEntry* __restrict__ ring, u64 stride, u64 first, u64 count,
                                                        u32 S, u32 ring_len, u64* __restrict__ index /* may be NULL */,
                                                        u64* __restrict__ bytes /* may be NULL */, u64 pitch) {
    const u64 g = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (g >= count) return;
    for (u32 s = 0; s < S; ++s)
        for (u32 k = 0; k < ring_len; ++k) {
            const RingEntry en = ring_load(ring + ((u64)s * stride + first + g) * ring_len + k);
            if (index) index[((u64)k * S + s) * pitch + g] = en.index;
            if (bytes) bytes[((u64)k * S + s) * pitch + g] = en.bytes;
        }
}

}  // namespace rafthip
