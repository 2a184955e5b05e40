// wire.h — K8: raftpb.Message bytes -> stepLeader bookkeeping, on the device.
//
// One lane decodes one gogo-protobuf serialized raftpb.Message
// (raftpb/raft.proto:71-108) the way Message.Unmarshal does
// (raftpb/raft.pb.go:1801-2179, skipRaft :3027-3104), applies raft.Step's term
// gate (raft.go:1051-1139), looks the sender up in the group's id table
// (`r.trk.Progress[m.From]`, raft.go:1327) and performs the same max / or
// updates as K7.  A workgroup's messages are one contiguous byte range: it is
// staged into LDS with 16-byte loads and parsed from there; a range that does
// not fit is parsed straight from global memory.
#pragma once
#include "kernels.h"
#include "progress.h"

namespace rafthip {

using i64 = long long;

constexpr int kWireBlock = 256;         // messages per workgroup
constexpr u32 kWireLdsBytes = 24576;    // staged bytes per workgroup (96 B per message on average)

enum : uint8_t {  // = RAFTHIP_WIRE_*
    kWireApplied = 0,
    kWireOtherType = 1,
    kWireNoProgress = 2,
    kWireStaleTerm = 3,
    kWireHigherTerm = 4,
    kWireNested = 5,
    kWireMalformed = 6,
    kWireBadGroup = 7,
    kWireNeedsLog = 8,  // rafthip_step_wire only
    kWireVote = 9,      // a vote response of the group's own campaign: recorded (rafthip_upload_campaign)
    kWireStatuses = 10,
};
constexpr uint8_t kEvNeedsHost = 0xFE;  // record type of a kWireNeedsLog message (progress.h)
constexpr uint8_t kMsgFlagHasCtx = 2;
constexpr uint8_t kWireNoSlot = 0xFF;

struct WireFields {
    u32 type;      // int32 MessageType: the low 32 bits of the varint
    u64 from, term, index, log_term, reject_hint;
    bool reject;
    bool nested;   // a non-empty entries / snapshot / responses payload was seen
    u64 ctx;       // first min(ctx_len, 8) bytes of Context, little-endian
    u64 ctx_len;
};

// Byte sources of the parser.  Positions are the reader's own type: 64-bit offsets into the caller's buffer for
// global memory, 32-bit offsets into the staged window for LDS — the byte loop is all position arithmetic and
// compares, and in 64 bits it cost twice the VALU instructions and spilled its exec masks (round 3).
struct GlobalBytes {
    using pos_t = i64;
    const uint8_t* p;
    __device__ __forceinline__ u32 operator()(i64 pos) const { return p[pos]; }
};
struct LdsBytes {
    using pos_t = u32;
    const uint8_t* sm;  // sm[i] holds byte i of the window
    __device__ __forceinline__ u32 operator()(u32 pos) const { return sm[pos]; }
};

// The parser is written with ONE exit per loop and error flags instead of early returns: on the GPU every
// `return false` inside nested divergent loops becomes lane-mask bookkeeping on the scalar unit, and the first form
// of this parser (early returns throughout) spent 2200 scalar instructions per wavefront against 1100 vector ones
// — the decode kernel ran at the scalar unit's issue rate (0.94 ms of 1.09 ms per 2^24 messages, SQ counters).
//
// The varint loop every field of Unmarshal uses (e.g. raft.pb.go:1807-1820): at most 10 bytes, bits past 63 are
// shifted out; running off the buffer (io.ErrUnexpectedEOF) or an 11th byte (ErrIntOverflowRaft) is an error.
// The loop carries ONE boolean (more) and the result (good).  Do not fold the three conditions into the `while`
// with the last byte kept live across the exit: hipcc 7.2 built that form so that a lane whose varint ended one
// byte before its neighbour's came out wrong (tests/test_gpu_wire.py::test_neighbouring_lanes_...).
template <class R>
__device__ __forceinline__ bool wire_varint(const R& rd, typename R::pos_t& pos, typename R::pos_t end, u64& v) {
    v = 0;
    u32 shift = 7;
    bool good = false, more = false;
    if (pos < end) {  // the first byte outside the loop: tags and most small values end here
        const u32 b = rd(pos++);
        v = b & 0x7Fu;
        good = b < 0x80u;
        more = !good && pos < end;
    }
    while (more) {
        const u32 b = rd(pos++);
        v |= (u64)(b & 0x7Fu) << shift;  // shift <= 63 here
        shift += 7;
        good = b < 0x80u;
        more = !good && shift < 70u && pos < end;
    }
    return good;
}

// skipRaft (raft.pb.go:3027-3104) on [pos, end): skips one unknown field,
// groups included.  The caller's `iNdEx + skippy > l` check is folded in.
template <class R>
__device__ bool wire_skip(const R& rd, typename R::pos_t& pos, typename R::pos_t end) {
    typename R::pos_t i = pos;
    int depth = 0;
    bool ok = true, done = false;
    while (ok && !done) {
        ok = i < end;  // io.ErrUnexpectedEOF (inside a group, after the first field)
        u64 wire = 0;
        ok = ok && wire_varint(rd, i, end, wire);
        const u32 wt = (u32)wire & 7u;
        // one arm per wire type; each leaves ok / i / depth for the common tail
        u64 val = 0;
        const bool has_varint = ok && (wt == 0 || wt == 2);
        if (has_varint) ok = wire_varint(rd, i, end, val);
        if (ok) {
            typename R::pos_t adv = 0;
            if (wt == 1) adv = 8;
            if (wt == 5) adv = 4;
            if (wt == 2) {
                ok = val <= (u64)(end - i);  // ErrInvalidLengthRaft (negative as int) / past the end
                adv = ok ? (typename R::pos_t)val : 0;
            }
            i += adv;
            if (wt == 3) ++depth;
            if (wt == 4) {
                ok = depth != 0;  // ErrUnexpectedEndOfGroupRaft
                --depth;
            }
            if (wt > 5) ok = false;  // illegal wireType 6, 7
        }
        ok = ok && i <= end;
        done = ok && depth == 0;
    }
    if (done) pos = i;
    return done;
}

// Message.Unmarshal's top level.  Returns false where the reference returns an
// error.  Nested messages (fields 7, 9, 14) are measured, not decoded: a
// non-empty one sets `nested` and the message is left to the host.
template <class R>
__device__ bool wire_parse(const R& rd, typename R::pos_t pos, typename R::pos_t end, WireFields& f) {
    using P = typename R::pos_t;
    f.type = 0;
    f.from = f.term = f.index = f.log_term = f.reject_hint = 0;
    f.reject = f.nested = false;
    f.ctx = f.ctx_len = 0;
    bool ok = true;
    while (ok && pos < end) {
        const P pre = pos;
        u64 wire = 0;
        ok = wire_varint(rd, pos, end, wire);
        const int32_t field = (int32_t)(u32)(wire >> 3);  // fieldNum := int32(wire >> 3)
        const u32 wt = (u32)wire & 7u;
        ok = ok && wt != 4 && field > 0;  // end group for non-group; illegal tag
        // fields 1-6, 8, 10, 11, 13: varints; 7, 9, 12, 14: length-delimited; everything else is skipped
        const bool scalar = field > 0 && field <= 13 && ((0x2D7Eu >> field) & 1u) != 0;
        const bool delimited = field > 0 && field <= 14 && ((0x5280u >> field) & 1u) != 0;
        if (ok && scalar) {
            u64 v = 0;
            ok = wt == 0 && wire_varint(rd, pos, end, v);  // wrong wireType
            if (ok) {
                f.type = field == 1 ? (u32)v : f.type;
                f.from = field == 3 ? v : f.from;
                f.term = field == 4 ? v : f.term;
                f.log_term = field == 5 ? v : f.log_term;
                f.index = field == 6 ? v : f.index;
                f.reject = field == 10 ? v != 0 : f.reject;
                f.reject_hint = field == 11 ? v : f.reject_hint;
            }
        } else if (ok && delimited) {
            u64 len = 0;
            ok = wt == 2 && wire_varint(rd, pos, end, len);
            ok = ok && len <= (u64)(end - pos);  // ErrInvalidLengthRaft (negative as int) / io.ErrUnexpectedEOF
            if (ok) {
                if (field == 12) {  // last occurrence wins (:2101)
                    f.ctx_len = len;
                    u64 c = 0;
#pragma unroll
                    for (u32 j = 0; j < 8; ++j)
                        if ((u64)j < len) c |= (u64)rd(pos + (P)j) << (8 * j);
                    f.ctx = c;
                } else if (len > 0) {
                    f.nested = true;
                }
                pos += (P)len;
            }
        } else if (ok) {
            pos = pre;
            ok = wire_skip(rd, pos, end);
        }
    }
    return ok;
}

// A message outside the staged window (a workgroup whose 256 messages span more than the window): the same
// parser over global memory, kept out of line so that the common path's code and registers stay small.
__device__ __noinline__ bool wire_parse_global(const uint8_t* bytes, i64 pos, i64 end, WireFields& f) {
    return wire_parse(GlobalBytes{bytes}, pos, end, f);
}

// The per-group lookup state of the wire path as ONE row per group, H = 3 + S
// rounded up to 4 words (32-byte aligned rows): [term][ctx][ctx_len][ids...].
// A decoded message gathers its term, its sender's slot and the pending
// ReadIndex context from one or two lines instead of five.
// Word 2 packs two bytes: bits 0-7 the length of the pending ReadIndex context, bits 8-15 the vote
// response type the group's campaign counts (0 = not campaigning).
constexpr u32 kHdrTerm = 0, kHdrCtx = 1, kHdrCtxLen = 2, kHdrIds = 3;
constexpr u32 kHdrCampaignShift = 8;
// Rows are whole aligned lines: a message's lookup is a random gather, bound by the lines and bytes it touches —
// 64 bytes up to 5 slots (the three- and five-voter groups most hosts run), 128 bytes up to 13 slots, multiples of
// 128 above (96-byte rows at 32-byte alignment straddled two 128-byte lines half of the time).
__host__ __device__ __forceinline__ u32 wire_hdr_words(u32 S) { return S <= 5 ? 8u : (kHdrIds + S + 15u) & ~15u; }
constexpr u32 kHdrLineIds = 16 - kHdrIds;  // ids in the first line of a row

// The first line of a group's row, all of it requested at once (eight 16-byte loads in flight instead of a
// chain of dependent 8-byte ones), and the slot search over it.
struct HdrLine {
    u64 w[16];
    __device__ __forceinline__ void load(const u64* row, u32 H) {  // H: words per row, 8 or a multiple of 16 (wave-uniform)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Vec<u64, 2> v = load_vec<u64, 2>(row + 2 * k);
            w[2 * k] = v.v[0];
            w[2 * k + 1] = v.v[1];
        }
        if (H > 8) {
#pragma unroll
            for (int k = 4; k < 8; ++k) {
                const Vec<u64, 2> v = load_vec<u64, 2>(row + 2 * k);
                w[2 * k] = v.v[0];
                w[2 * k + 1] = v.v[1];
            }
        } else {
#pragma unroll
            for (int k = 8; k < 16; ++k) w[k] = 0;  // never an id: find() stops at S <= 5
        }
    }
    // the first slot whose id is `from` (r.trk.Progress[m.From]); ids beyond the first line are read from the row
    __device__ __forceinline__ u32 find(const u64* row, u32 S, u64 from) const {
        u32 slot = kWireNoSlot;
#pragma unroll
        for (int s = (int)kHdrLineIds - 1; s >= 0; --s)
            if ((u32)s < S && w[kHdrIds + s] == from) slot = (u32)s;
        if (slot == kWireNoSlot)
            for (u32 s = kHdrLineIds; s < S; ++s)
                if (row[kHdrIds + s] == from) return s;
        return slot;
    }
};

__global__ __launch_bounds__(kWireBlock) void k_ingest_wire(
    const uint8_t* __restrict__ bytes, u64 nbytes, const u64* __restrict__ offsets,
    const u32* __restrict__ groups, u64 n, u64 G, u32 S, const u64* __restrict__ hdr /* [G][H] */, u32 H,
    u64* __restrict__ match, u64 stride,
    void* __restrict__ active, void* __restrict__ acks, int mbytes, Msg* __restrict__ out_msgs,
    uint8_t* __restrict__ out_status, u64* __restrict__ counters /* [kWireStatuses] */,
    u64* __restrict__ out_aux /* non-NULL: decode only, the records go on to the Progress stepper */,
    ProgRec* __restrict__ rec, u32* __restrict__ prev, uint8_t* __restrict__ actions, u32 index_base
    /* with out_aux: k_progress_link's work for message i, list index index_base + i */) {
    __shared__ __align__(16) uint8_t sm[kWireLdsBytes + 32];
    __shared__ u32 hist[kWireStatuses];
    const u64 i0 = (u64)blockIdx.x * kWireBlock;
    const u64 cnt = n - i0 < (u64)kWireBlock ? n - i0 : (u64)kWireBlock;
    if (threadIdx.x < kWireStatuses) hist[threadIdx.x] = 0;

    // the workgroup's byte window [w0, w1)
    const u64 w0 = offsets[i0], w1 = offsets[i0 + cnt];
    const bool window = w0 <= w1 && w1 <= nbytes && w1 - w0 <= kWireLdsBytes;
    u32 lead = 0;  // sm[lead] holds byte w0
    if (window && w1 > w0) {
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(bytes + w0);
        const uintptr_t a1 = reinterpret_cast<uintptr_t>(bytes + w1);
        const uintptr_t base = a0 & ~(uintptr_t)15;
        lead = (u32)(a0 - base);
        const uintptr_t full = a1 & ~(uintptr_t)15;  // 16-byte granules wholly below a1
        for (uintptr_t a = base + (uintptr_t)threadIdx.x * 16; a < full; a += (uintptr_t)kWireBlock * 16) {
            // the first granule may start below bytes + w0: same 16-byte granule as a valid byte
            *reinterpret_cast<uint4*>(sm + (a - base)) = *reinterpret_cast<const uint4*>(a);
        }
        if (full >= base) {  // tail bytes of the last, partial granule
            const uintptr_t t0 = full > a0 ? full : a0;
            for (uintptr_t a = t0 + threadIdx.x; a < a1; a += kWireBlock)
                sm[a - base] = *reinterpret_cast<const uint8_t*>(a);
        }
    }
    __syncthreads();

    const u64 i = i0 + threadIdx.x;
    if (i < n) {
        const u64 start = offsets[i], end = offsets[i + 1];
        const u32 group = groups[i];
        uint8_t status = kWireApplied;
        WireFields f;
        bool decoded = false;
        if ((u64)group >= G) {
            status = kWireBadGroup;
        } else if (!(start <= end && end <= nbytes)) {
            status = kWireMalformed;
        } else {
            if (window && start >= w0 && end <= w1) {
                decoded = wire_parse(LdsBytes{sm + lead}, (u32)(start - w0), (u32)(end - w0), f);
            } else {
                WireFields out_of_line;  // its address leaves the kernel: a copy, so that `f` itself stays in registers
                decoded = wire_parse_global(bytes, (i64)start, (i64)end, out_of_line);
                f = out_of_line;
            }
            if (!decoded) status = kWireMalformed;
        }
        u32 slot = kWireNoSlot;
        uint8_t flags = 0;
        u32 rec_type = 0xFFu;
        u64 aux = 0;
        const bool apply = out_aux == nullptr;
        if (decoded) {
            rec_type = f.type <= 0xFFu ? f.type : 0xFFu;
            if (f.ctx_len > 0) flags |= kMsgFlagHasCtx;
            const bool app = f.type == kMsgAppResp, hb = f.type == kMsgHeartbeatResp;
            const bool vote = f.type == kMsgVoteResp || f.type == kMsgPreVoteResp;
            if (!app && !hb && !vote) {
                status = kWireOtherType;
            } else if (f.nested) {
                status = kWireNested;
            } else {
                // requesting the line before the parse, so that its latency runs under the byte loop, was measured:
                // slower (1172 -> 1239 us per 2^24 messages; the kernel is bound by what it moves, not by latency)
                const u64* row = hdr + (u64)group * H;  // one line holds all the group's lookup state (S <= 13)
                HdrLine line;
                line.load(row, H);
                const u64 term = line.w[kHdrTerm];
                // raft.go:1068-1076: a granted MsgPreVoteResp carries the pre-candidate's future term and does
                // not make it a follower
                if (f.term != 0 && f.term > term && !(f.type == kMsgPreVoteResp && !f.reject)) {
                    status = kWireHigherTerm;
                } else if (f.term != 0 && f.term < term) {
                    status = kWireStaleTerm;
                } else if (vote) {
                    // stepCandidate, raft.go:1628-1648: only the response type of the campaign in progress is
                    // polled; leaders, followers and the other candidate state ignore it
                    const u32 mine = (u32)(line.w[kHdrCtxLen] >> kHdrCampaignShift) & 0xFFu;
                    if (mine == 0 || f.type != mine) {
                        status = kWireOtherType;
                    } else {
                        if (f.from != 0) slot = line.find(row, S, f.from);
                        // a vote from an id without a row can change neither the result (config members
                        // only, majority.go:188-197) nor the counts (tracker.go:274-287)
                        status = slot == kWireNoSlot ? kWireNoProgress : kWireVote;
                    }
                } else {
                    if (f.from != 0) slot = line.find(row, S, f.from);
                    if (slot == kWireNoSlot) {
                        status = kWireNoProgress;
                    } else {
                        if (apply) or_slot_bit(active, mbytes, group, slot);  // raft.go:1337,1528
                        if (app) {
                            if (!apply) {
                                // nextProbeIdx is RejectHint unless LogTerm > 0 asks for
                                // findConflictByTerm on the leader's log (raft.go:1362-1459): the host's
                                if (f.reject) {
                                    aux = f.reject_hint;
                                    if (f.log_term > 0) {
                                        status = kWireNeedsLog;
                                        rec_type = kEvNeedsHost;
                                    }
                                }
                            } else if (!f.reject) {
                                atomicMax(match + (u64)slot * stride + group, f.index);  // :1477
                            }
                        } else {
                            const u64 want = line.w[kHdrCtxLen] & 0xFFu;
                            if (want != 0 && f.ctx_len == want && f.ctx == line.w[kHdrCtx]) {
                                flags |= kMsgFlagCtx;
                                if (apply) or_slot_bit(acks, mbytes, group, slot);  // read_only.go:68-76
                            }
                        }
                    }
                }
            }
        }
        if (out_msgs) {
            Vec<u64, 2> o;
            o.v[0] = (u64)group | ((u64)(slot & 0xFFu) << 32) | ((u64)rec_type << 40) |
                     ((u64)(decoded && f.reject ? 1 : 0) << 48) | ((u64)flags << 56);
            o.v[1] = decoded ? f.index : 0;
            store_vec<u64, 2>(reinterpret_cast<u64*>(out_msgs + i), o);
        }
        if (out_status) out_status[i] = status;
        if (out_aux) {
            out_aux[i] = aux;
            if (slot != kWireNoSlot && status != kWireVote) {  // applied or needs-log: an event of Progress (slot, group)
                const u32 li = index_base + (u32)i;
                if (rec) prev[li] = atomicExch(&rec[(u64)slot * stride + group].head, li + 1);  // NULL: K9p links per tile
                actions[li] = kActDeferred;
            } else {
                actions[index_base + (u32)i] = kActDropped;
            }
        }
        atomicAdd(&hist[status], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kWireStatuses && hist[threadIdx.x] != 0) atomicAdd(counters + threadIdx.x, (u64)hist[threadIdx.x]);
}

// ---------------------------------------------------------------------------
// K10 — bcastHeartbeat as serialized bytes (raft.go:698-716 -> sendHeartbeat
// :669-685 -> r.send :502-533): for every Progress row of a leader's group but
// its own, the raftpb.Message
//   {Type: MsgHeartbeat, To: id, From: r.id, Term: r.Term,
//    Commit: min(pr.Match, raftLog.committed), Context: the last pending
//    ReadIndex context or nil}
// as Message.Marshal writes it (raftpb/raft.pb.go:937-1025: every non-nullable
// scalar in field order, Context only when non-nil).  Lane = group, loop over
// slots; a workgroup builds its 256 messages of one slot in LDS (each lane
// zeroes its 72-byte slot, then stores the message byte by byte) and writes them out as one
// contiguous run.  At 1 byte read per 3 written the kernel moves 4.6 TB/s, 92 % of what a
// device-to-device copy reaches on the same GPU.
// A group whose self mask is empty (not led here) emits nothing.
// ---------------------------------------------------------------------------
constexpr u32 kHeartbeatStride = 72;  // longest: 2+11+11+11+2+2+11+2+2+(2+8)+2 = 66 bytes
constexpr u32 kMsgHeartbeat = 8;

struct ByteSink {  // appends bytes to a lane's message slot in LDS (the slot is zeroed first)
    uint8_t* dst;
    u32 n;
    __device__ __forceinline__ void put(u32 b) { dst[n++] = (uint8_t)b; }
    // encodeVarintRaft, raft.pb.go:1233-1243.  Values below 2^32 (ids, terms, most indexes) stay in 32-bit
    // arithmetic: 64-bit shifts run at a fraction of the 32-bit rate.
    __device__ __forceinline__ void varint(u64 v) {
        while ((v >> 32) != 0) {
            put(((u32)v & 0x7Fu) | 0x80u);
            v >>= 7;
        }
        u32 x = (u32)v;
        while (x >= 0x80u) {
            put((x & 0x7Fu) | 0x80u);
            x >>= 7;
        }
        put(x);
    }
};

template <class MaskT>
__global__ __launch_bounds__(kBlock) void k_emit_heartbeats(
    const u64* __restrict__ match, u64 stride, const u64* __restrict__ commit, const u64* __restrict__ hdr, u32 H,
    const MaskT* __restrict__ self_mask, u64 first, u64 count, u32 S, uint8_t* __restrict__ out_bytes /* [S][count][72] */,
    uint8_t* __restrict__ out_len /* [S][count] */,
    const uint8_t* __restrict__ beat /* NULL, or the action bytes of the last rafthip_tick: only groups that stepped MsgBeat */) {
    constexpr u32 W = kHeartbeatStride / 8;
    __shared__ __align__(16) u64 sm2[2][kBlock * W];  // two buffers: one barrier per slot
    const u64 g0 = (u64)blockIdx.x * kBlock;
    const u64 gl = g0 + threadIdx.x;  // group within the range
    const bool live = gl < count;
    const u64 g = first + (live ? gl : 0);
    const u64* row = hdr + g * H;
    u64 term = 0, ctx = 0, from = 0, com = 0;
    u32 ctx_len = 0, self_slot = kWireNoSlot;
    if (live) {
        const u64 self = (u64)self_mask[g];
        if (self != 0 && (!beat || (beat[g] & 16u /* RAFTHIP_TICK_BEAT */))) {
            self_slot = (u32)__ffsll((unsigned long long)self) - 1;
            if (self_slot < S) from = row[kHdrIds + self_slot];
            term = row[kHdrTerm];
            ctx = row[kHdrCtx];
            ctx_len = (u32)(row[kHdrCtxLen] & 0xFFu);
            if (ctx_len > 8) ctx_len = 0;  // never stored (k_hdr_set_field); the message slot holds 8 context bytes
            com = commit[g];
        }
    }
    const u64 in_block = count - g0 < (u64)kBlock ? count - g0 : (u64)kBlock;
    for (u32 s = 0; s < S; ++s) {
        u64* sm = sm2[s & 1];
#pragma unroll
        for (u32 k = 0; k < W; ++k) sm[threadIdx.x * W + k] = 0;  // the rest of the slot reads 0
        ByteSink w{reinterpret_cast<uint8_t*>(sm + threadIdx.x * W), 0};
        const u64 to = live && self_slot < S && s != self_slot ? row[kHdrIds + s] : 0;
        if (to != 0 && from != 0) {
            const u64 m = __builtin_nontemporal_load(match + (u64)s * stride + g);
            w.put(0x08), w.put(kMsgHeartbeat);
            w.put(0x10), w.varint(to);
            w.put(0x18), w.varint(from);
            w.put(0x20), w.varint(term);
            w.put(0x28), w.put(0);  // LogTerm
            w.put(0x30), w.put(0);  // Index
            w.put(0x40), w.varint(m < com ? m : com);  // raft.go:676
            w.put(0x50), w.put(0);  // Reject
            w.put(0x58), w.put(0);  // RejectHint
            if (ctx_len != 0) {     // bcastHeartbeat passes nil when nothing is pending (raft.go:699-706)
                w.put(0x62), w.put(ctx_len);
                for (u32 j = 0; j < ctx_len; ++j) w.put((u32)(ctx >> (8 * j)));
            }
            w.put(0x68), w.put(0);  // Vote
        }
        const u32 len = w.n;
        if (live) out_len[(u64)s * count + gl] = (uint8_t)len;
        __syncthreads();
        // the block's messages of this slot are one contiguous run of in_block * 72 bytes
        uint8_t* dst = out_bytes + ((u64)s * count + g0) * kHeartbeatStride;
        const u32 words = (u32)in_block * W;
        u64* d64 = reinterpret_cast<u64*>(dst);  // 72 * anything is 8-byte aligned when out_bytes is
        for (u32 k = threadIdx.x; k < words; k += kBlock) __builtin_nontemporal_store(sm[k], d64 + k);  // written once
        // no second barrier: the next slot fills the other buffer, and this one is not refilled before every
        // lane has passed the next slot's barrier, i.e. finished this copy
    }
}

// ids [S][count] planes (the layout every other upload uses) -> the rows.
__global__ __launch_bounds__(kBlock) void k_hdr_set_ids(const u64* __restrict__ src, u64 count, u32 S, u32 H,
                                                        u64* __restrict__ hdr /* row of the first group */) {
    const u64 g = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (g >= count) return;
    for (u32 s = 0; s < S; ++s) hdr[g * H + kHdrIds + s] = src[(u64)s * count + g];
}
// One word (and optionally a byte-wide second field) per group -> its row.
// A byte above max8 is stored as 0 and its word as 0: a ReadIndex context longer than the 8 bytes a row
// holds is "nothing pending" (the host entry point rejects such a length; this guards device-resident input).
__global__ __launch_bounds__(kBlock) void k_hdr_set_field(const u64* __restrict__ src, const uint8_t* __restrict__ src8,
                                                          u64 count, u32 H, u32 field, u32 field8, u32 shift8,
                                                          u32 max8, u64* __restrict__ hdr) {
    const u64 g = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (g >= count) return;
    u32 b = src8 ? src8[g] : 0u;
    const bool over = b > max8;
    if (over) b = 0;
    if (src) hdr[g * H + field] = over ? 0ull : src[g];
    if (src8) {  // one byte of word field8, at bit `shift8`
        u64& w = hdr[g * H + field8];
        w = (w & ~((u64)0xFFu << shift8)) | ((u64)b << shift8);
    }
}

}  // namespace rafthip
