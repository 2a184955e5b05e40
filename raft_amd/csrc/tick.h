// tick.h — K12, RawNode.Tick for every group of an engine in one pass.
//
// The caller of the CheckQuorum tally and of the heartbeat fan-out: rawnode.go:64-66 calls
// r.tick(), which is tickHeartbeat for a leader (raft.go:835-862) and tickElection for everyone else
// (raft.go:823-832).  A multi-raft host ticks every group it holds every tick interval; per group that is two
// counters, three comparisons and — once per election timeout on a leader with CheckQuorum — the QuorumActive tally
// of K3 plus the RecentActive reset of raft.go:1236-1242.  All of it is local to the group, so one lane handles four
// consecutive groups: 16-byte accesses on the three u32 timer planes, one dword on each byte plane.
//
// What the kernel leaves to the host is what needs the host: Step(MsgHup) (the campaign), bcastHeartbeat for the
// groups whose action byte has kTickBeat (rafthip_emit_heartbeats), and for a leader that steps down the rest of
// raft.reset (raft.go:760-789: a fresh randomized timeout from the host's generator, the Progress rows, the votes).
#pragma once
#include "kernels.h"

namespace rafthip {

// StateType, raft.go:47-53
constexpr u32 kStateFollower = 0, kStateLeader = 2;
// rafthip.h: RAFTHIP_TICK_*
constexpr u32 kTickHup = 1, kTickCheckQuorum = 2, kTickStepDown = 4, kTickAbortTransfer = 8, kTickBeat = 16;
constexpr int kTickActions = 5;
constexpr u32 kTickPromotable = 1, kTickTransfer = 2;

struct TickPlanes {
    u32* __restrict__ election_elapsed;
    u32* __restrict__ heartbeat_elapsed;
    const u32* __restrict__ randomized_timeout;
    uint8_t* __restrict__ role;
    uint8_t* __restrict__ flags;
    uint8_t* __restrict__ action;
};

// counts: how many groups of this pass carry each action bit (the host skips what did not happen); summed per
// workgroup in LDS first, then added into one of kTickCountCopies sets of counters, a cache line each (one set for
// all workgroups made the pass three times as long: 65 536 workgroups x 4 atomics on the same five words); the host
// adds the sets up.
constexpr int kTickCountCopies = 256, kTickCountStride = 8;  // words per set: 64 bytes
// One workgroup's 1024 groups of one engine; `block` is the workgroup's index within that engine.
template <typename MaskT>
__device__ __forceinline__ void tick_block(const TickPlanes t, u64 G, int S, u32 election_timeout, u32 heartbeat_timeout,
                                           u32 check_quorum, MaskT* __restrict__ active, const MaskT* __restrict__ in_mask,
                                           const MaskT* __restrict__ out_mask, const MaskT* __restrict__ learner_mask,
                                           int uniform_voters, const MaskT* __restrict__ self_mask, u64* __restrict__ counts,
                                           u32 block) {
    __shared__ u32 wg_counts[kTickActions];
    if (threadIdx.x < kTickActions) wg_counts[threadIdx.x] = 0;
    __syncthreads();
    const u64 g0 = ((u64)block * kBlock + threadIdx.x) * 4;  // < stride (a multiple of 4096)
    const u64 slots = S >= 64 ? kU64Max : ((1ull << S) - 1ull);
    Vec<u32, 4> EL = load_stream<u32, 4>(t.election_elapsed + g0);
    Vec<u32, 4> HB = load_stream<u32, 4>(t.heartbeat_elapsed + g0);
    const Vec<u32, 4> RT = load_stream<u32, 4>(t.randomized_timeout + g0);
    const u32 role_in = *reinterpret_cast<const u32*>(t.role + g0);
    const u32 flags_in = *reinterpret_cast<const u32*>(t.flags + g0);
    // Which of the four groups run MsgCheckQuorum in this tick is known from the timers alone; the masks are only
    // read by a wavefront that holds such a group (once per election timeout per leader).
    bool due = false;
    if (check_quorum) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            due |= ((role_in >> (8 * j)) & 0xFFu) == kStateLeader && EL.v[j] + 1u >= election_timeout;
    }
    Vec<MaskT, 4> A, M0, M1, ML, K;
    const bool wave_due = __any(due);
    const bool lrn = in_mask && learner_mask;
    if (wave_due) {
        A = load_vec<MaskT, 4>(active + g0);
        K = load_vec<MaskT, 4>(self_mask + g0);
        if (in_mask) {
            M0 = load_vec<MaskT, 4>(in_mask + g0);
            M1 = load_vec<MaskT, 4>(out_mask + g0);
            if (lrn) ML = load_vec<MaskT, 4>(learner_mask + g0);
        }
    }
    u32 role_out = 0, flags_out = 0, act_out = 0;
    u32 tally[kTickActions] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u32 role = (role_in >> (8 * j)) & 0xFFu, fl = (flags_in >> (8 * j)) & 0xFFu;
        u32 el = EL.v[j], hb = HB.v[j], act = 0;
        if (role == kStateLeader) {  // tickHeartbeat, raft.go:835-862
            ++hb;
            ++el;
            if (el >= election_timeout) {
                el = 0;
                if (check_quorum) {  // Step(MsgCheckQuorum), raft.go:1231-1243
                    act |= kTickCheckQuorum;
                    u64 m0, m1, yes = (u64)A.v[j];
                    if (in_mask) {
                        m0 = (u64)M0.v[j] & slots;
                        m1 = (u64)M1.v[j] & slots;
                        if (lrn) yes &= ~(u64)ML.v[j];  // learners cast no vote (tracker.go:220-222)
                    } else {
                        m0 = (uniform_voters >= 64 ? kU64Max : ((1ull << uniform_voters) - 1ull)) & slots;
                        m1 = 0;
                    }
                    const bool quorum_active = vote_joint(vote_half(m0, yes, ~yes), vote_half(m1, yes, ~yes)) == 3;
                    if (!quorum_active) {
                        // becomeFollower(r.Term, None) -> reset, raft.go:864-871, 760-789: both timers to zero, the
                        // transfer aborted, every Progress replaced by a fresh one (RecentActive false, the leader's
                        // own included)
                        act |= kTickStepDown;
                        role = kStateFollower;
                        hb = 0;
                        fl &= ~kTickTransfer;
                        A.v[j] = 0;
                    } else {
                        A.v[j] = A.v[j] & K.v[j];  // raft.go:1236-1242: everyone but ourselves
                    }
                }
                if (role == kStateLeader && (fl & kTickTransfer)) {  // raft.go:846-849
                    act |= kTickAbortTransfer;
                    fl &= ~kTickTransfer;
                }
            }
            if (role == kStateLeader && hb >= heartbeat_timeout) {  // raft.go:852-861
                hb = 0;
                act |= kTickBeat;
            }
        } else {  // tickElection, raft.go:823-832
            ++el;
            if ((fl & kTickPromotable) && el >= RT.v[j]) {  // promotable() && pastElectionTimeout()
                el = 0;
                act |= kTickHup;
            }
        }
        if (g0 + j >= G) act = 0;  // padding rows of the planes: never reported
        EL.v[j] = el;
        HB.v[j] = hb;
        role_out |= role << (8 * j);
        flags_out |= fl << (8 * j);
        act_out |= act << (8 * j);
#pragma unroll
        for (int b = 0; b < kTickActions; ++b) tally[b] += (act >> b) & 1u;
    }
    store_stream<u32, 4>(t.election_elapsed + g0, EL);
    store_stream<u32, 4>(t.heartbeat_elapsed + g0, HB);
    *reinterpret_cast<u32*>(t.action + g0) = act_out;
    if (role_out != role_in) *reinterpret_cast<u32*>(t.role + g0) = role_out;
    if (flags_out != flags_in) *reinterpret_cast<u32*>(t.flags + g0) = flags_out;
    if (due) store_vec<MaskT, 4>(active + g0, A);
    if (counts) {
#pragma unroll
        for (int b = 0; b < kTickActions; ++b) {
            u32 s = tally[b];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if ((threadIdx.x & 63) == 0 && s) atomicAdd(&wg_counts[b], s);
        }
        __syncthreads();
        if (threadIdx.x < kTickActions && wg_counts[threadIdx.x])
            atomicAdd(counts + (block % kTickCountCopies) * kTickCountStride + threadIdx.x, (u64)wg_counts[threadIdx.x]);
    }
}

template <typename MaskT>
__global__ __launch_bounds__(kBlock) void k_tick(const TickPlanes t, u64 G, int S, u32 election_timeout,
                                                 u32 heartbeat_timeout, u32 check_quorum, MaskT* __restrict__ active,
                                                 const MaskT* __restrict__ in_mask, const MaskT* __restrict__ out_mask,
                                                 const MaskT* __restrict__ learner_mask, int uniform_voters,
                                                 const MaskT* __restrict__ self_mask, u64* __restrict__ counts) {
    tick_block<MaskT>(t, G, S, election_timeout, heartbeat_timeout, check_quorum, active, in_mask, out_mask, learner_mask,
                      uniform_voters, self_mask, counts, blockIdx.x);
}

// K12m — the same pass over the shards of several engines of one GPU in ONE launch (rafthip_tick_many), for the
// host that keeps several shards per GPU (K1m's reason: a 2^20-group shard's tick moves 24 MB and ends after a few
// microseconds, as long as the boundary between two kernels).  The shard table travels by value in the kernel
// arguments; every shard keeps its own planes, config, tick constants and counter sets.
constexpr int kTickManyMax = 16;
struct TickShards {
    u32 n;
    u32 first_block[kTickManyMax + 1];
    TickPlanes planes[kTickManyMax];
    u64 groups[kTickManyMax];
    void* active[kTickManyMax];
    const void* in_mask[kTickManyMax];
    const void* out_mask[kTickManyMax];
    const void* learner_mask[kTickManyMax];
    const void* self_mask[kTickManyMax];
    u64* counts[kTickManyMax];
    u32 election_timeout[kTickManyMax], heartbeat_timeout[kTickManyMax];
    uint8_t check_quorum[kTickManyMax], slots[kTickManyMax], uniform_voters[kTickManyMax];
};

template <typename MaskT>
__global__ __launch_bounds__(kBlock) void k_tick_many(const TickShards sh) {
    u32 k = 0;
    for (u32 i = 1; i < sh.n; ++i) k = blockIdx.x >= sh.first_block[i] ? i : k;  // wave-uniform: scalar loads + compares
    tick_block<MaskT>(sh.planes[k], sh.groups[k], (int)sh.slots[k], sh.election_timeout[k], sh.heartbeat_timeout[k],
                      sh.check_quorum[k], static_cast<MaskT*>(sh.active[k]), static_cast<const MaskT*>(sh.in_mask[k]),
                      static_cast<const MaskT*>(sh.out_mask[k]), static_cast<const MaskT*>(sh.learner_mask[k]),
                      (int)sh.uniform_voters[k], static_cast<const MaskT*>(sh.self_mask[k]), sh.counts[k],
                      blockIdx.x - sh.first_block[k]);
}

// stepFollower's `r.electionElapsed = 0` for the groups that heard from their leader (MsgApp, MsgHeartbeat, MsgSnap:
// raft.go:1681-1692), as a list of group ids: a store of zero, idempotent, so order and duplicates do not matter.
__global__ __launch_bounds__(kBlock) void k_tick_heard(u32* __restrict__ election_elapsed, u64 G, const u32* __restrict__ groups,
                                                       u64 n, u64* __restrict__ rejected) {
    const u64 i = (u64)blockIdx.x * kBlock + threadIdx.x;
    bool bad = false;
    if (i < n) {
        const u32 g = groups[i];
        if ((u64)g < G) {
            election_elapsed[g] = 0;
        } else {
            bad = true;
        }
    }
    const u64 b = __ballot(bad);
    if (b != 0 && (threadIdx.x & 63) == 0) atomicAdd(rejected, (u64)__popcll(b));
}

// ---------------------------------------------------------------------------
// The groups of the last pass whose action byte has a bit of `mask`, as (group, action) records in group order
// (rafthip_read_tick_events): what the host walks instead of the whole action plane — campaigns, step-downs and
// transfer timeouts are a handful per tick among millions of groups.  Three small passes over the action bytes,
// 16 per lane: per-workgroup counts, one workgroup scans them, then every workgroup writes its records at its
// offset.  No atomics order the output: it is sorted by group and reproducible.
// ---------------------------------------------------------------------------
struct TickEvent {  // = rafthip_tick_event
    u32 group;
    u32 action;
};
constexpr int kEvPerLane = 16, kEvTile = kBlock * kEvPerLane;  // 4096 groups per workgroup: divides the plane stride

__device__ __forceinline__ u32 tick_hits(const uint8_t* __restrict__ action, u64 g0, u64 G, u32 mask, Vec<u32, 4>& a) {
    a = load_vec<u32, 4>(reinterpret_cast<const u32*>(action + g0));
    u32 n = 0;
#pragma unroll
    for (int i = 0; i < kEvPerLane; ++i)
        n += (((a.v[i >> 2] >> (8 * (i & 3))) & mask) != 0 && g0 + i < G) ? 1u : 0u;
    return n;
}

__global__ __launch_bounds__(kBlock) void k_tick_event_count(const uint8_t* __restrict__ action, u64 G, u32 mask,
                                                             u32* __restrict__ tile_counts) {
    __shared__ u32 wave_tot[kBlock / 64];
    Vec<u32, 4> a;
    u32 n = tick_hits(action, ((u64)blockIdx.x * kBlock + threadIdx.x) * kEvPerLane, G, mask, a);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 s = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) s += wave_tot[w];
        tile_counts[blockIdx.x] = s;
    }
}

// One workgroup: tile_counts[i] becomes the number of records in front of tile i; *total = all of them.
__global__ __launch_bounds__(kBlock) void k_tick_event_scan(u32* __restrict__ tile_counts, u32 tiles, u64* __restrict__ total) {
    __shared__ u32 part[kBlock];
    const u32 per = (tiles + kBlock - 1) / kBlock, lo = threadIdx.x * per, hi = lo + per < tiles ? lo + per : tiles;
    u32 s = 0;
    for (u32 i = lo; i < hi; ++i) s += tile_counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (int i = 0; i < kBlock; ++i) {
            const u32 v = part[i];
            part[i] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    u32 run = part[threadIdx.x];
    for (u32 i = lo; i < hi; ++i) {
        const u32 v = tile_counts[i];
        tile_counts[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(kBlock) void k_tick_event_pack(const uint8_t* __restrict__ action, u64 G, u32 mask,
                                                            const u32* __restrict__ tile_offsets, u64 group_base,
                                                            TickEvent* __restrict__ out, u64 max_events) {
    __shared__ u32 wave_tot[kBlock / 64];
    const u64 g0 = ((u64)blockIdx.x * kBlock + threadIdx.x) * kEvPerLane;
    Vec<u32, 4> a;
    const u32 n = tick_hits(action, g0, G, mask, a);
    // exclusive prefix of n over the wavefront, then over the workgroup's wavefronts
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    u64 at = tile_offsets[blockIdx.x] + (incl - n);
    for (int w = 0; w < wave; ++w) at += wave_tot[w];
    if (n == 0) return;
#pragma unroll
    for (int i = 0; i < kEvPerLane; ++i) {
        const u32 act = (a.v[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        if ((act & mask) != 0 && g0 + i < G) {
            if (at < max_events) out[at] = TickEvent{(u32)(group_base + g0 + i), act};
            ++at;
        }
    }
}

}  // namespace rafthip
