// scatter_part.h — K5p, the partitioned form of K5 (Progress.MaybeUpdate's Match part for a stream
// of acks, tracker/progress.go:168-177 via raft.go:1477) for batches that touch a large share of the
// engine's cells.  K5 pays one scattered 64-bit atomic per ack (26.6e9/s on MI355X, 19e9/s once the
// planes exceed the TLB reach); here the acks are first brought into tile order with two LDS-staged
// partition passes that write whole runs, and then every tile of Match is updated in LDS and written
// back with whole-line stores:
//
//   count   every workgroup histograms its slice of the records by tile (LDS), then adds its histogram
//           into the global per-tile counts (contiguous atomics); invalid records are counted as dropped
//   scan    one workgroup: exclusive scan -> tile bases; bucket and tile cursors; the chunk table of
//           pass 2; the largest tile count decides whether the batch is too skewed for this form
//   pass 1  by bucket (128 tiles).  A workgroup takes chunks of 8192 records: per-bucket ranks from LDS
//           atomics, one global reservation per bucket and chunk, the records staged bucket by bucket
//           in LDS and written out as whole runs (consecutive lanes -> consecutive addresses); the next
//           chunk's records are already in flight while the current one is staged and written
//   pass 2  the same inside every bucket, by tile
//   apply   one workgroup per tile: the tile's S x T cells in LDS, its records applied with LDS 64-bit
//           max, the tile written back
//
// max is commutative and idempotent, so the order in which a tile's records arrive is immaterial and
// the result is bit-identical to K5's.  A batch whose acks pile onto one tile (more than kPartSkewLimit
// records in a tile) would serialise in that tile's workgroup: the scan raises a flag, the partition
// and apply kernels leave at once, and k_scatter_acks_flagged applies the original records with K5's
// atomics — decided on the device, no host round trip.
#pragma once
#include "kernels.h"

namespace rafthip {

constexpr int kPartBlock = 1024;       // lanes of a partition workgroup
constexpr int kPartApplyBlock = 512;   // lanes of a tile's workgroup in the apply pass
// records per chunk: 128 KB of LDS staging, one 1024-lane workgroup per CU.  A chunk is cut into one run per bucket, so
// with 128 buckets (cfg5: 16 384 tiles) 4096-record chunks gave 512-byte runs and 8192-record ones give 1 KB: the two
// passes at 2^26 acks 515 + 562 -> together 140 us less; with 32 buckets (2^22 groups) the runs were long already and
// nothing changes (round 3; 512 lanes x 4096 records, two workgroups per CU, before)
constexpr int kPartChunk = 8192;
constexpr int kPartR = kPartChunk / kPartBlock;
constexpr u32 kPartLB = 7;             // tiles per bucket = 128
constexpr u32 kPartMaxBuckets = 256;
constexpr u32 kPartMaxTiles = kPartMaxBuckets << kPartLB;  // 32768: the count pass keeps one LDS counter per tile
constexpr u32 kPartSkewLimit = 1u << 17;  // records in one tile beyond which the batch takes the atomic form
constexpr u32 kPartCountBlock = 1024;

struct PartBufs {
    u32* tile_count;     // [tiles_pad]  zero between calls
    u32* tile_base;      // [tiles_pad + 1]
    u32* tile_cursor;    // [tiles_pad]
    u32* bucket_cursor;  // [kPartMaxBuckets]
    u32* chunk_first;    // [kPartMaxBuckets + 1]
    u32* flag;           // [0] 1 = too skewed: atomic form
};

// inclusive scan over the 64 lanes of a wavefront
__device__ __forceinline__ u32 wave_scan_incl(u32 v) {
    const u32 lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 o = (u32)__shfl_up((int)v, d, 64);
        if (lane >= (u32)d) v += o;
    }
    return v;
}

__global__ __launch_bounds__(kPartCountBlock) void k_part_count(const Ack* __restrict__ recs, u64 n, u64 slice, u64 G, u32 S,
                                                                u32 tile_shift, u32 tiles_pad, u32* __restrict__ tile_count,
                                                                u64* __restrict__ rejected) {
    extern __shared__ u32 part_hist[];
    for (u32 t = threadIdx.x; t < tiles_pad; t += kPartCountBlock) part_hist[t] = 0;
    __syncthreads();
    const u64 lo = (u64)blockIdx.x * slice, hi = lo + slice < n ? lo + slice : n;
    u32 bad = 0;
    for (u64 i = lo + threadIdx.x; i < hi; i += kPartCountBlock) {
        const u64 w = __builtin_nontemporal_load(reinterpret_cast<const u64*>(recs + i));
        const u32 group = (u32)w, slot = (u32)(w >> 32);
        if ((u64)group < G && slot < S) {
            atomicAdd(&part_hist[group >> tile_shift], 1u);
        } else {
            ++bad;
        }
    }
    const u64 b = __ballot(bad != 0);
    if (b != 0) {  // rare: records without a Progress row (raft.go:1327-1331)
        u32 tot = bad;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) tot += (u32)__shfl_xor((int)tot, d, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(rejected, (u64)tot);
    }
    __syncthreads();
    for (u32 t = threadIdx.x; t < tiles_pad; t += kPartCountBlock) {
        const u32 c = part_hist[t];
        if (c) atomicAdd(tile_count + t, c);
    }
}

// One workgroup of 1024 lanes; tiles_pad <= 32768 is a multiple of 128.  The counts pass through LDS
// (tiles_pad words of dynamic LDS) so that every global access is coalesced: a lane scanning its own run of
// consecutive tiles straight from global memory cost 35 us at 2^14 tiles, one partial line per lane and store.
__global__ __launch_bounds__(1024) void k_part_scan(PartBufs pb, u32 tiles_pad, u32 nb) {
    extern __shared__ u32 scan_buf[];  // [tiles_pad] counts, then exclusive prefix sums
    __shared__ u32 wsum[16];
    __shared__ u32 bstart[kPartMaxBuckets + 1];
    __shared__ u32 s_max;
    const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const u32 per = (tiles_pad + 1023) / 1024;  // <= 32 consecutive tiles per lane
    const u32 lo = t * per;
    if (t == 0) s_max = 0;
    for (u32 i = t; i < tiles_pad; i += 1024) {
        scan_buf[i] = pb.tile_count[i];
        pb.tile_count[i] = 0;  // for the next call
    }
    __syncthreads();
    u32 sum = 0, mx = 0;
    for (u32 k = 0; k < per; ++k) {
        const u32 c = lo + k < tiles_pad ? scan_buf[lo + k] : 0;
        sum += c;
        mx = c > mx ? c : mx;
    }
    const u32 incl = wave_scan_incl(sum);
    if (lane == 63) wsum[wave] = incl;
    if (mx) atomicMax(&s_max, mx);
    __syncthreads();
    u32 before = 0;
    for (u32 w = 0; w < wave; ++w) before += wsum[w];
    u32 run = before + incl - sum;
    for (u32 k = 0; k < per; ++k) {
        if (lo + k >= tiles_pad) break;
        const u32 c = scan_buf[lo + k];
        scan_buf[lo + k] = run;
        if (((lo + k) & ((1u << kPartLB) - 1u)) == 0) bstart[(lo + k) >> kPartLB] = run;
        run += c;
    }
    if (t == 1023) {
        const u32 total = before + incl;
        pb.tile_base[tiles_pad] = total;
        bstart[nb] = total;
    }
    __syncthreads();
    for (u32 i = t; i < tiles_pad; i += 1024) {
        const u32 v = scan_buf[i];
        pb.tile_base[i] = v;
        pb.tile_cursor[i] = v;
    }
    // bucket cursors and the chunk table of pass 2: chunk_first[b] = chunks of the buckets before b (nb <= 256)
    u32 mine = 0;
    if (t < nb) {
        pb.bucket_cursor[t] = bstart[t];
        mine = (bstart[t + 1] - bstart[t] + kPartChunk - 1) / kPartChunk;
    }
    const u32 ci = wave_scan_incl(mine);
    __syncthreads();  // wsum is free again
    if (t < kPartMaxBuckets && lane == 63) wsum[wave] = ci;
    __syncthreads();
    if (t < kPartMaxBuckets) {
        u32 b4 = 0;
        for (u32 w = 0; w < wave; ++w) b4 += wsum[w];
        if (t < nb) pb.chunk_first[t] = b4 + ci - mine;
        if (t == nb - 1) pb.chunk_first[nb] = b4 + ci;
    }
    if (t == 0) pb.flag[0] = s_max > kPartSkewLimit ? 1u : 0u;
}

// What a partition pass moves.  kPartAcks: 16-byte ack records (group, slot, index), unchanged from pass to pass.
// kPartEvents (K9p, progress_part.h): pass 1 reads the 16-byte event records of rafthip_step_progress and writes
// them PACKED — the bucket is known from a record's position after pass 1, so the key keeps only the group
// within its bucket and has room for the stream index, which orders the events of one Progress:
//   bits 0..16  group & (2^(tile_shift + 7) - 1)     bits 17..22 slot      bits 23..25 event code
//   bit  26     reject     bits 27..28 flags & 3      bits 32..63 stream index;  second word: index
enum { kPartAcks = 0, kPartEvents = 1 };
constexpr u32 kEvKeySlotShift = 17, kEvKeyCodeShift = 23, kEvKeyRejectShift = 26, kEvKeyFlagsShift = 27;
// event code of a record type the Progress stepper takes (progress.h: prog_event_type), 7 = not an event
__device__ __forceinline__ u32 part_event_code(u32 type) {
    switch (type) {
        case 4: return 0;      // MsgAppResp
        case 9: return 1;      // MsgHeartbeatResp
        case 10: return 2;     // MsgUnreachable
        case 11: return 3;     // MsgSnapStatus
        case 0xF1: return 4;   // sent an append
        case 0xF2: return 5;   // sent a snapshot
        case 0xFE: return 6;   // a response only the host can step
        default: return 7;
    }
}

// One partition pass over chunks of kPartChunk records; persistent workgroups, the next chunk's
// records are requested before the current one is staged.
//   LEVEL 1: chunk c = records [c * chunk, +chunk) of src; digit = tile >> kPartLB; cursors = bucket_cursor
//   LEVEL 2: chunk c lies inside one bucket (chunk table); digit = tile & 127; cursors = tile_cursor + bucket * 128
template <int LEVEL, int KIND = kPartAcks>
__global__ __launch_bounds__(kPartBlock) void k_part_partition(const Ack* __restrict__ src, u64 n, u64 G, u32 S, u32 tile_shift, u32 nb,
                                                               PartBufs pb, Ack* __restrict__ dst) {
    extern __shared__ ulonglong2 part_stage[];  // [kPartChunk] (+ kPartChunk digit bytes for pass 1 over events)
    __shared__ u32 hist[kPartMaxBuckets], lstart[kPartMaxBuckets + 1], gbase[kPartMaxBuckets], wsum[4];
    static_assert(kPartMaxBuckets <= 256, "the per-bucket tables are filled by the first four waves");
    constexpr bool kPack = KIND == kPartEvents && LEVEL == 1;   // raw event records in, packed ones out
    constexpr bool kPacked = KIND == kPartEvents && LEVEL == 2;  // packed in, packed out: every record is valid
    uint8_t* const stage_dig = reinterpret_cast<uint8_t*>(part_stage + kPartChunk);  // kPack only
    if (pb.flag[0]) return;
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 nchunks = LEVEL == 1 ? (u32)((n + kPartChunk - 1) / kPartChunk) : pb.chunk_first[nb];
    u32* const cursor = LEVEL == 1 ? pb.bucket_cursor : pb.tile_cursor;
    const u32 bins = LEVEL == 1 ? nb : (1u << kPartLB);

    auto range = [&](u32 c, u64& lo, u64& hi, u32& cur0) {
        if (LEVEL == 1) {
            lo = (u64)c * kPartChunk;
            hi = lo + kPartChunk < n ? lo + kPartChunk : n;
            cur0 = 0;
        } else {
            u32 b = 0;  // the last bucket whose first chunk <= c: binary search over nb <= 256 entries (wave-uniform)
            for (u32 step = kPartMaxBuckets / 2; step >= 1; step >>= 1)
                if (b + step < nb && pb.chunk_first[b + step] <= c) b += step;
            const u32 b_lo = pb.tile_base[b << kPartLB], b_hi = pb.tile_base[(b + 1) << kPartLB];
            lo = (u64)b_lo + (u64)(c - pb.chunk_first[b]) * kPartChunk;
            hi = lo + kPartChunk < b_hi ? lo + kPartChunk : b_hi;
            cur0 = b << kPartLB;
        }
    };
    auto fetch = [&](u64 lo, u64 hi, ulonglong2 (&rec)[kPartR]) {
#pragma unroll
        for (int j = 0; j < kPartR; ++j) {
            const u64 i = lo + (u64)j * kPartBlock + tid;
            rec[j].x = ~0ull;  // group 2^32-1, slot 2^32-1: never valid
            rec[j].y = 0;
            if (i < hi) {
                typedef u64 v2 __attribute__((ext_vector_type(2)));
                const v2 r = __builtin_nontemporal_load(reinterpret_cast<const v2*>(src + i));
                rec[j].x = r[0];
                rec[j].y = r[1];
            }
        }
    };

    if (tid < kPartMaxBuckets) hist[tid] = 0;
    u32 c = blockIdx.x;
    if (c >= nchunks) return;
    u64 lo, hi;
    u32 cur0;
    range(c, lo, hi, cur0);
    ulonglong2 nxt[kPartR];
    fetch(lo, hi, nxt);
    __syncthreads();
    for (;;) {
        ulonglong2 rec[kPartR];
#pragma unroll
        for (int j = 0; j < kPartR; ++j) rec[j] = nxt[j];
        const u32 my_cur0 = cur0;
        const u64 my_lo = lo;
        const u32 cn = c + gridDim.x;
        const bool more = cn < nchunks;
        if (more) {
            range(cn, lo, hi, cur0);
            fetch(lo, hi, nxt);
        }
        u32 rank[kPartR], dig[kPartR];
#pragma unroll
        for (int j = 0; j < kPartR; ++j) {
            dig[j] = 0xFFFFFFFFu;
            if constexpr (kPack) {
                const u64 w = rec[j].x;
                const u32 group = (u32)w, slot = (u32)(w >> 32) & 0xFFu, code = part_event_code((u32)(w >> 40) & 0xFFu);
                if ((u64)group < G && slot < S && code < 7u) {  // an unfetched record reads as group 2^32-1
                    dig[j] = (group >> tile_shift) >> kPartLB;
                    const u64 stream = my_lo + (u64)j * kPartBlock + tid;
                    rec[j].x = (u64)(group & ((1u << (tile_shift + kPartLB)) - 1u)) | ((u64)slot << kEvKeySlotShift) |
                               ((u64)code << kEvKeyCodeShift) | ((u64)(((w >> 48) & 0xFFu) != 0) << kEvKeyRejectShift) |
                               (((w >> 56) & 3u) << kEvKeyFlagsShift) | (stream << 32);
                    rank[j] = atomicAdd(&hist[dig[j]], 1u);
                }
            } else if constexpr (kPacked) {
                if (rec[j].x != ~0ull) {  // the filler of an unfetched record; a packed key never has bits 29..31 set
                    dig[j] = ((u32)rec[j].x >> tile_shift) & ((1u << kPartLB) - 1u);
                    rank[j] = atomicAdd(&hist[dig[j]], 1u);
                }
            } else {
                const u32 group = (u32)rec[j].x, slot = (u32)(rec[j].x >> 32);
                if ((u64)group < G && slot < S) {
                    const u32 tile = group >> tile_shift;
                    dig[j] = LEVEL == 1 ? tile >> kPartLB : tile & ((1u << kPartLB) - 1u);
                    rank[j] = atomicAdd(&hist[dig[j]], 1u);
                }
            }
        }
        __syncthreads();  // A: ranks complete
        u32 mine = 0, incl = 0;
        if (tid < kPartMaxBuckets) {
            mine = tid < bins ? hist[tid] : 0;
            hist[tid] = 0;  // for the next chunk
            gbase[tid] = mine ? atomicAdd(cursor + my_cur0 + tid, mine) : 0;
            incl = wave_scan_incl(mine);
            if (lane == 63) wsum[wave] = incl;
        }
        __syncthreads();  // B
        if (tid < kPartMaxBuckets) {
            u32 before = 0;
            for (u32 w = 0; w < wave; ++w) before += wsum[w];
            lstart[tid] = before + incl - mine;
            if (tid == kPartMaxBuckets - 1) lstart[kPartMaxBuckets] = before + incl;
        }
        __syncthreads();  // C
#pragma unroll
        for (int j = 0; j < kPartR; ++j)
            if (dig[j] != 0xFFFFFFFFu) {
                const u32 at = lstart[dig[j]] + rank[j];
                part_stage[at] = rec[j];
                if constexpr (kPack) stage_dig[at] = (uint8_t)dig[j];  // the packed key no longer names its bucket
            }
        __syncthreads();  // D
        const u32 total = lstart[kPartMaxBuckets];
        for (u32 i = tid; i < total; i += kPartBlock) {
            const ulonglong2 r = part_stage[i];
            u32 d;
            if constexpr (kPack) {
                d = stage_dig[i];
            } else {
                const u32 tile = (u32)r.x >> tile_shift;
                d = LEVEL == 1 ? tile >> kPartLB : tile & ((1u << kPartLB) - 1u);
            }
            *reinterpret_cast<ulonglong2*>(dst + (u64)gbase[d] + (i - lstart[d])) = r;
        }
        if (!more) break;
        c = cn;
        __syncthreads();  // E: the stage and the tables are free again
    }
}

// One workgroup per tile of T = 1 << tile_shift groups: cells[slot][group in tile] in LDS.
__global__ __launch_bounds__(kPartApplyBlock) void k_part_apply(u64* __restrict__ match, u64 stride, u32 S, u32 tile_shift,
                                                           const Ack* __restrict__ bins, PartBufs pb) {
    extern __shared__ u64 part_cells[];  // [S << tile_shift]
    if (pb.flag[0]) return;
    const u32 tile = blockIdx.x;
    const u32 lo = pb.tile_base[tile], hi = pb.tile_base[tile + 1];
    if (lo == hi) return;
    const u32 T = 1u << tile_shift, cells = S << tile_shift;
    const u64 g0 = (u64)tile << tile_shift;
    for (u32 i = threadIdx.x * 2; i < cells; i += kPartApplyBlock * 2) {
        const u32 s = i >> tile_shift, t = i & (T - 1);
        const Vec<u64, 2> v = load_stream<u64, 2>(match + (u64)s * stride + g0 + t);
        part_cells[i] = v.v[0];
        part_cells[i + 1] = v.v[1];
    }
    __syncthreads();
    for (u32 i = lo + threadIdx.x; i < hi; i += kPartApplyBlock) {
        typedef u64 v2 __attribute__((ext_vector_type(2)));
        const v2 r = __builtin_nontemporal_load(reinterpret_cast<const v2*>(bins + i));
        const u32 group = (u32)r[0], slot = (u32)(r[0] >> 32);
        atomicMax(&part_cells[(slot << tile_shift) + (group & (T - 1))], r[1]);
    }
    __syncthreads();
    for (u32 i = threadIdx.x * 2; i < cells; i += kPartApplyBlock * 2) {
        const u32 s = i >> tile_shift, t = i & (T - 1);
        Vec<u64, 2> v;
        v.v[0] = part_cells[i];
        v.v[1] = part_cells[i + 1];
        store_stream<u64, 2>(match + (u64)s * stride + g0 + t, v);
    }
}

// K5 on the original records when the scan found the batch too skewed (flag set); otherwise nothing.
// Invalid records were counted by k_part_count already.
__global__ __launch_bounds__(kBlock) void k_scatter_acks_flagged(u64* __restrict__ match, u64 stride, u64 G, u32 S,
                                                                 const Ack* __restrict__ recs, u64 n, const u32* __restrict__ flag) {
    if (!flag[0]) return;  // the usual case: a modest grid leaves at once (one workgroup per 256 records cost 58 us at 2^26)
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < n; i += (u64)gridDim.x * kBlock) {
        const Vec<u64, 2> raw = load_vec<u64, 2>(reinterpret_cast<const u64*>(recs + i));
        const u32 group = (u32)raw.v[0], slot = (u32)(raw.v[0] >> 32);
        if ((u64)group < G && slot < S) atomicMax(match + (u64)slot * stride + group, raw.v[1]);
    }
}
constexpr u32 kPartFlaggedGrid = 8192;  // workgroups of the grid-stride fallback: 32 waves on each of 256 CUs

}  // namespace rafthip
