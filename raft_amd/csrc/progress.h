// progress.h — K9: the leader's per-peer replication state on the device
// (SURVEY.md §8f row 3): tracker.Progress (tracker/progress.go), its Inflights
// window (tracker/inflights.go) and the stepLeader branches that drive them
// (raft.go:1333-1586), for a stream of events.
//
// Unlike Match = max(...) these updates do not commute: a rejection followed
// by an ack is not an ack followed by a rejection.  Events of one batch are
// therefore applied in stream order per Progress, by exactly one lane:
//
//   k_progress_link   every event swaps its index into head[cell] (one 32-bit
//                     atomic exchange) and keeps what it found in prev[i]: a
//                     linked list per Progress, in whatever order the
//                     exchanges ran.  Leader-log growth records (kMsgAppended)
//                     are applied here, i.e. before every response.
//   k_progress_apply  the event that still owns head[cell] (the last to
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

struct ProgPlanes {
    u64* match;        // [S][stride]
    u64* next;         // [S][stride]
    uint8_t* pstate;   // [S][stride] bits 0-1 State, bit 2 MsgAppFlowPaused
    u64* pending;      // [S][stride] PendingSnapshot
    u32* infl_meta;    // [S][stride] start | count << 16
    u64* infl_bytes;   // [S][stride]
    u64* ring_index;   // [cap][S][stride]
    u64* ring_bytes;   // [cap][S][stride]
    u64 stride;
    u64 plane;         // S * stride
    u32 cap;           // Inflights.size
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

// Inflights.FreeLE, inflights.go:98-128, on the cell's ring in global memory.
__device__ void infl_free_le(Prog& p, const ProgPlanes& pl, u64 cell, u64 to) {
    if (p.icount == 0 || to < pl.ring_index[(u64)p.istart * pl.plane + cell]) return;
    u32 idx = p.istart, i = 0;
    u64 bytes = 0;
    for (; i < p.icount; ++i) {
        if (to < pl.ring_index[(u64)idx * pl.plane + cell]) break;
        bytes += pl.ring_bytes[(u64)idx * pl.plane + cell];
        if (++idx >= pl.cap) idx -= pl.cap;
    }
    p.icount -= i;
    p.ibytes -= bytes;
    p.istart = idx;
    if (p.icount == 0) p.istart = 0;
}

// One event on one Progress: stepLeader, raft.go:1333-1586, and the send-side
// bookkeeping of maybeSendAppend, raft.go:645,653.  Returns the action byte.
__device__ uint8_t prog_step(Prog& p, const ProgPlanes& pl, u64 cell, u32 type, bool reject, u32 flags, u64 index,
                             u64 aux, u64 last_index, u64 first_index) {
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
                    } else if (p.state == kStateSnapshot && p.match + 1 >= first_index) {
                        prog_become_probe(p);
                        prog_become_replicate(p);
                    } else if (p.state == kStateReplicate) {
                        infl_free_le(p, pl, cell, index);
                    }
                    act |= kActUpdated;
                    if (old_paused) act |= kActWasPaused;
                    if (p.match == last_index) act |= kActCaughtUp;
                }
            }
            break;
        case kMsgHeartbeatResp:  // :1527-1561
            p.active = true;
            p.paused = 0;
            if (p.match < last_index || p.state == kStateProbe) act |= kActSendAppend;
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
                    pl.ring_index[(u64)pos * pl.plane + cell] = index;
                    pl.ring_bytes[(u64)pos * pl.plane + cell] = aux;
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
           type == kEvSentApp || type == kEvSentSnap;
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

__global__ __launch_bounds__(kBlock) void k_progress_link(const Msg* __restrict__ msgs, u32 n, u64 G, u32 S,
                                                          u64 stride, u32* __restrict__ head,
                                                          u32* __restrict__ prev, u64* __restrict__ term_last,
                                                          uint8_t* __restrict__ actions) {
    const u32 i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
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
        prev[i] = atomicExch(head + (u64)f.slot * stride + f.group, i + 1);
        act = kActDeferred;
    }
    actions[i] = act;
}

__global__ __launch_bounds__(kBlock) void k_progress_apply(const Msg* __restrict__ msgs,
                                                           const u64* __restrict__ aux, u32 n, u64 G, u32 S,
                                                           ProgPlanes pl, u32* __restrict__ head,
                                                           const u32* __restrict__ prev, void* __restrict__ active,
                                                           void* __restrict__ acks, int mbytes,
                                                           const u64* __restrict__ term_last,
                                                           const u64* __restrict__ first_index,
                                                           uint8_t* __restrict__ actions) {
    const u32 i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const MsgFields own = load_msg(msgs, i);
    if (!prog_event_type(own.type) || (u64)own.group >= G || own.slot >= S) return;
    const u64 cell = (u64)own.slot * pl.stride + own.group;
    if (head[cell] != i + 1) return;  // another event of this Progress exchanged later: it owns the list

    u32 len = 0;
    for (u32 j = i + 1; j != 0 && len <= kStepLimit; j = prev[j - 1]) ++len;
    if (len > kStepLimit) {  // all of them stay DEFERRED
        head[cell] = 0;
        return;
    }
    Prog p;
    p.match = pl.match[cell];
    p.next = pl.next[cell];
    p.pending = pl.pending[cell];
    p.ibytes = pl.infl_bytes[cell];
    const u32 ps = pl.pstate[cell], meta = pl.infl_meta[cell];
    p.state = ps & 3u;
    p.paused = (ps >> 2) & 1u;
    p.istart = meta & 0xFFFFu;
    p.icount = meta >> 16;
    p.active = p.read_ack = false;
    const u64 last_index = term_last[own.group];
    const u64 first = first_index ? first_index[own.group] : 0;

    u32 done = 0;  // stream index + 1 of the last event stepped
    for (u32 k = 0; k < len; ++k) {
        u32 cur = i + 1;
        if (len > 1) {  // the smallest index above `done`: lists are short, a re-walk beats a sort
            cur = 0xFFFFFFFFu;
            for (u32 j = i + 1; j != 0; j = prev[j - 1])
                if (j > done && j < cur) cur = j;
        }
        const MsgFields f = cur == i + 1 ? own : load_msg(msgs, cur - 1);
        actions[cur - 1] = prog_step(p, pl, cell, f.type, f.reject, f.flags, f.index, aux ? aux[cur - 1] : 0,
                                     last_index, first);
        done = cur;
    }
    pl.match[cell] = p.match;
    pl.next[cell] = p.next;
    pl.pending[cell] = p.pending;
    pl.infl_bytes[cell] = p.ibytes;
    pl.pstate[cell] = (uint8_t)(p.state | (p.paused << 2));
    pl.infl_meta[cell] = p.istart | (p.icount << 16);
    if (p.active) or_slot_bit(active, mbytes, own.group, own.slot);
    if (p.read_ack) or_slot_bit(acks, mbytes, own.group, own.slot);
    head[cell] = 0;
}

// Fresh Inflights for a range of groups (Progress upload = ResetState, progress.go:102-107).
__global__ __launch_bounds__(kBlock) void k_progress_clear_inflights(u32* __restrict__ meta, u64* __restrict__ bytes,
                                                                     u64 stride, u64 first, u64 count, u32 S) {
    const u64 g = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (g >= count) return;
    for (u32 s = 0; s < S; ++s) {
        meta[(u64)s * stride + first + g] = 0;
        bytes[(u64)s * stride + first + g] = 0;
    }
}

}  // namespace rafthip
