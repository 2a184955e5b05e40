// progress_part.h — K9p, the partitioned form of K9 (progress.h): tracker.Progress / tracker.Inflights
// driven by stepLeader's branches (raft.go:1333-1586) for a stream of events, in stream order per Progress.
//
// K9 pays one scattered 32-bit exchange per event to link it into its Progress's list (26.6e9/s) and then
// 3.5 scattered memory requests per event in the apply pass (39e9 requests/s: random 64-byte lines of HBM).
// Here the events are first brought into tile order with K5p's two LDS-staged partition passes
// (scatter_part.h; the records travel packed, 16 bytes with the stream index inside), and one workgroup per
// tile of T groups then sorts its events by Progress cell and steps them in that order: the Progress
// records, Match cells, last indexes, Inflights rings and bit-plane words a wave touches are neighbours —
// lines shared between lanes and visited once, instead of one random line per request.  (Tile order alone is
// not enough: with ~1000 tiles in flight their working sets do not fit the L2s, and events stepped in arrival
// order inside a tile fetched as many lines as K9 did)
//
//   count    classifies every record (kMsgAppended raises the group's last index here, i.e. before every
//            response, as K9's link pass does), writes DROPPED / 0 / the guessed outcome (part_action_guess)
//            into its action byte and histograms the events by tile
//   scan     k_part_scan
//   pass 1   k_part_partition<1, kPartEvents>: by bucket, packing the records
//   pass 2   k_part_partition<2, kPartEvents>: by tile
//   apply    one workgroup per tile: a counting sort by Progress cell (counters in LDS) makes the events of
//            one Progress adjacent and hands consecutive cells to consecutive lanes; the first event of a
//            cell takes the cell's events in ascending STREAM index (the partition passes do not keep
//            stream order; the index in the key does) and steps them with prog_step — the same function K9
//            runs, so states, Inflights and action bytes are bit-identical
//
// A batch the scan finds piled onto one tile takes K9's link and apply kernels instead (flag on the device).
#pragma once
#include "progress.h"
#include "scatter_part.h"

namespace rafthip {

constexpr int kProgPartBlock = 512;

// The action byte the count pass writes for an event before anything is stepped: what the event most often ends
// with in steady state — an accepted ack updates its Progress, everything else asks for nothing.  The apply pass
// then stores an action byte only where the event's real outcome differs (and DEFERRED where an event is not
// stepped): the byte goes to actions[stream index], one scattered store each, and at 2^24 events those stores
// were a fifth of the pass.
__device__ __forceinline__ uint8_t part_action_guess(u32 type, bool reject) {
    return type == kMsgAppResp && !reject ? (uint8_t)kActUpdated : (uint8_t)0;
}

// Classify + histogram.  Grid of `count_wgs` workgroups, each over a contiguous slice.
__global__ __launch_bounds__(kPartCountBlock) void k_prog_part_count(const Msg* __restrict__ msgs, u64 n, u64 slice, u64 G, u32 S,
                                                                     u32 tile_shift, u32 tiles_pad, u32* __restrict__ tile_count,
                                                                     u64* __restrict__ term_last, uint8_t* __restrict__ actions) {
    extern __shared__ u32 part_hist[];
    for (u32 t = threadIdx.x; t < tiles_pad; t += kPartCountBlock) part_hist[t] = 0;
    __syncthreads();
    const u64 lo = (u64)blockIdx.x * slice, hi = lo + slice < n ? lo + slice : n;
    for (u64 i = lo + threadIdx.x; i < hi; i += kPartCountBlock) {
        typedef u64 v2 __attribute__((ext_vector_type(2)));
        const v2 r = __builtin_nontemporal_load(reinterpret_cast<const v2*>(msgs + i));
        const u32 group = (u32)r[0], slot = (u32)(r[0] >> 32) & 0xFFu, type = (u32)(r[0] >> 40) & 0xFFu;
        uint8_t act;
        if (type == kMsgAppended) {
            act = kActDropped;
            if ((u64)group < G) {
                atomicMax(term_last + group, r[1]);
                act = 0;
            }
        } else if (part_event_code(type) >= 7u || (u64)group >= G || slot >= S) {
            act = kActDropped;
        } else {
            atomicAdd(&part_hist[group >> tile_shift], 1u);
            act = part_action_guess(type, ((r[0] >> 48) & 0xFFu) != 0);
        }
        actions[i] = act;
    }
    __syncthreads();
    for (u32 t = threadIdx.x; t < tiles_pad; t += kPartCountBlock) {
        const u32 c = part_hist[t];
        if (c) atomicAdd(tile_count + t, c);
    }
}

struct PackedEvent {
    u32 g_in_tile, slot, type, flags, stream;
    bool reject;
    u64 index;
};
__device__ __forceinline__ u32 part_event_type(u32 code) {
    // inverse of part_event_code
    return code == 0 ? (u32)kMsgAppResp : code == 1 ? (u32)kMsgHeartbeatResp : code == 2 ? (u32)kEvUnreachable
         : code == 3 ? (u32)kEvSnapStatus : code == 4 ? (u32)kEvSentApp : code == 5 ? (u32)kEvSentSnap : (u32)kEvNeedsHostRec;
}
__device__ __forceinline__ PackedEvent unpack_event(u64 key, u64 index, u32 tile_mask) {
    PackedEvent f;
    f.g_in_tile = (u32)key & tile_mask;
    f.slot = ((u32)key >> kEvKeySlotShift) & 63u;
    f.type = part_event_type(((u32)key >> kEvKeyCodeShift) & 7u);
    f.reject = (((u32)key >> kEvKeyRejectShift) & 1u) != 0;
    f.flags = ((u32)key >> kEvKeyFlagsShift) & 3u;
    f.stream = (u32)(key >> 32);
    f.index = index;
    return f;
}

// One workgroup per tile.  The tile's events are first sorted by Progress cell (slot-major, then group) with a
// counting sort whose counters live in LDS — cnt[slot << tile_shift | group in tile]: counts, then exclusive
// starts, then (after the placement) ends — into the tile's own run of `sorted`; the events of one Progress are
// then adjacent, and consecutive lanes own consecutive cells, so the Progress records, Match cells, last indexes
// and bit-plane words a wave touches share their lines.
__global__ __launch_bounds__(kProgPartBlock) void k_prog_part_apply(const Ack* __restrict__ bins, Ack* __restrict__ sorted,
                                                                    const u64* __restrict__ aux, u32 S, u32 tile_shift,
                                                                    ProgPlanes pl, PartBufs pb, void* __restrict__ active,
                                                                    void* __restrict__ acks, int mbytes,
                                                                    const u64* __restrict__ term_last,
                                                                    const u64* __restrict__ first_index,
                                                                    uint8_t* __restrict__ actions) {
    extern __shared__ u32 part_cnt[];  // [S << tile_shift]
    __shared__ u32 wsum[kProgPartBlock / 64];
    if (pb.flag[0]) return;
    const u32 tile = blockIdx.x;
    const u32 lo = pb.tile_base[tile], hi = pb.tile_base[tile + 1];
    if (lo == hi) return;
    const u32 T = 1u << tile_shift, cells = S << tile_shift;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 g0 = (u64)tile << tile_shift;
    const u64* bw = reinterpret_cast<const u64*>(bins);
    auto cell_of = [&](u32 key) { return (((key >> kEvKeySlotShift) & 63u) << tile_shift) | (key & (T - 1)); };
    for (u32 i = tid; i < cells; i += kProgPartBlock) part_cnt[i] = 0;
    __syncthreads();
    for (u32 i = lo + tid; i < hi; i += kProgPartBlock) atomicAdd(&part_cnt[cell_of((u32)bw[2 * (u64)i])], 1u);
    __syncthreads();
    {  // exclusive scan over the cells: a run of consecutive cells per lane
        const u32 per = (cells + kProgPartBlock - 1) / kProgPartBlock, c0 = tid * per;
        u32 sum = 0;
        for (u32 k = 0; k < per; ++k) sum += c0 + k < cells ? part_cnt[c0 + k] : 0;
        const u32 incl = wave_scan_incl(sum);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        u32 run = incl - sum;
        for (u32 w = 0; w < wave; ++w) run += wsum[w];
        for (u32 k = 0; k < per; ++k) {
            if (c0 + k >= cells) break;
            const u32 c = part_cnt[c0 + k];
            part_cnt[c0 + k] = run;
            run += c;
        }
    }
    __syncthreads();
    for (u32 i = lo + tid; i < hi; i += kProgPartBlock) {
        const Vec<u64, 2> r = load_vec<u64, 2>(bw + 2 * (u64)i);
        const u32 pos = atomicAdd(&part_cnt[cell_of((u32)r.v[0])], 1u);
        store_vec<u64, 2>(reinterpret_cast<u64*>(sorted + lo + pos), r);
    }
    __syncthreads();  // part_cnt[c] = end of cell c; the tile's run of `sorted` is visible to the whole workgroup
    const u64* sw = reinterpret_cast<const u64*>(sorted + lo);
    const u32 m = hi - lo;
    for (u32 pos = tid; pos < m; pos += kProgPartBlock) {
        const Vec<u64, 2> raw = load_vec<u64, 2>(sw + 2 * (u64)pos);
        const PackedEvent own = unpack_event(raw.v[0], raw.v[1], T - 1);
        const u32 c = (own.slot << tile_shift) | own.g_in_tile;
        const u32 first_pos = c ? part_cnt[c - 1] : 0;
        const u32 len = part_cnt[c] - first_pos;
        if (len > kStepLimit) {  // none of them is stepped
            actions[own.stream] = kActDeferred;
            continue;
        }
        if (pos != first_pos) continue;  // the first event of a cell steps all of them
        const u64 group = g0 + own.g_in_tile;
        const u64 cell = (u64)own.slot * pl.stride + group;
        u64* rw = reinterpret_cast<u64*>(pl.rec + cell);
        const Vec<u64, 2> r0 = load_vec<u64, 2>(rw), r1 = load_vec<u64, 2>(rw + 2);
        const u64 match0 = pl.match[cell];
        Prog p;
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
        const u64* last_index = term_last + group;
        const u64* first = first_index ? first_index + group : nullptr;

        u64 done = 0;  // stream index + 1 of the last event stepped
        bool blocked = false;  // a host-only event was met: it and what follows it on this Progress stay DEFERRED
        for (u32 k = 0; k < len; ++k) {
            PackedEvent f = own;
            if (len > 1) {  // the smallest stream index not yet stepped: the cell's events are adjacent and few
                u64 best = ~0ull;
                u32 at = pos;
                for (u32 j = pos; j < pos + len; ++j) {
                    const u64 s1 = (sw[2 * (u64)j] >> 32) + 1;
                    if (s1 > done && s1 < best) {
                        best = s1;
                        at = j;
                    }
                }
                if (at != pos) {
                    const Vec<u64, 2> o = load_vec<u64, 2>(sw + 2 * (u64)at);
                    f = unpack_event(o.v[0], o.v[1], T - 1);
                }
            }
            blocked = blocked || f.type == kEvNeedsHostRec;
            uint8_t act = kActDeferred;
            if (!blocked) {
                // aux: nextProbeIdx of a rejection, the bytes of a send — nothing else reads it
                const bool wants_aux = (f.type == kMsgAppResp && f.reject) || f.type == kEvSentApp;
                const u64 a = aux && wants_aux ? aux[f.stream] : 0;
                act = prog_step(p, pl, cell, f.type, f.reject, f.flags, f.index, a, last_index, first);
            }
            if (act != part_action_guess(f.type, f.reject)) actions[f.stream] = act;
            done = (u64)f.stream + 1;
        }
        if (p.match != match0) pl.match[cell] = p.match;
        Vec<u64, 2> w0, w1;
        w0.v[0] = p.next;
        w0.v[1] = p.pending;
        w1.v[0] = p.ibytes;
        w1.v[1] = (u64)rec_meta(p.istart, p.icount, p.state | (p.paused << 2));  // head = 0, as K9 leaves it
        store_vec<u64, 2>(rw, w0);
        store_vec<u64, 2>(rw + 2, w1);
        if (p.active) or_slot_bit(active, mbytes, group, own.slot);
        if (p.read_ack) or_slot_bit(acks, mbytes, group, own.slot);
    }
}

}  // namespace rafthip
