// kernels.h — gfx950 device code of the batched quorum engine.
//
// Everything here is integer order-statistic / popcount work bound by HBM
// bandwidth (no MFMA: there is no contraction on this path).  Layout rule:
// planes are [slot][group], so lane i of a wavefront touches group i and every
// global access is a contiguous 16-byte-per-lane stream.
//
//   K1  k_commit_uniform   non-joint, same voter slots in every group, V <= 8
//   K2  k_commit_masked    per-group incoming/outgoing slot masks, S <= 8
//   K4  k_commit_wide      9..64 slots: LDS-staged transpose + wave bitonic sort
//   K3  k_tally_*          VoteResult / QuorumActive / ReadIndex-ack tallies
//   K5  k_scatter_acks     MsgAppResp stream -> atomicMax on Match
//       k_pack_changed     ordered packing of changed (group, commit) pairs
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "sortnet.h"

namespace rafthip {

using u64 = unsigned long long;
using u32 = unsigned int;

constexpr int kBlock = 256;        // threads per workgroup: 4 wavefronts of 64
constexpr u64 kPlaneAlign = 4096;  // plane stride granule (elements)
constexpr u64 kU64Max = ~0ull;

struct Pair {  // = rafthip_commit_pair
    u64 group;
    u64 commit;
};
struct Ack {  // = rafthip_ack
    u32 group;
    u32 slot;
    u64 index;
};

struct Msg {  // = rafthip_msg
    u32 group;
    uint8_t slot;
    uint8_t type;
    uint8_t reject;
    uint8_t flags;
    u64 index;
};
constexpr uint8_t kMsgAppResp = 4, kMsgHeartbeatResp = 9;  // raftpb/raft.proto:41-69
constexpr uint8_t kMsgAppended = 0xF0;                     // local: the leader's log grew to `index`
constexpr uint8_t kMsgFlagCtx = 1;                         // heartbeat response carries the pending ReadIndex ctx

template <typename T, int N>
struct alignas(sizeof(T) * N) Vec {
    T v[N];
};

template <typename T, int N>
__device__ __forceinline__ Vec<T, N> load_vec(const T* p) {
    return *reinterpret_cast<const Vec<T, N>*>(p);
}
template <typename T, int N>
__device__ __forceinline__ void store_vec(T* p, const Vec<T, N>& x) {
    *reinterpret_cast<Vec<T, N>*>(p) = x;
}

// Streaming (non-temporal) forms.  Every plane is read once and written once
// per pass, so keeping its lines in L2 / Infinity Cache only evicts useful
// data; measured on MI355X (tools/kbench.hip, 2^20 x 5 voters, cold batches)
// `nt` loads+stores are worth 9-10 % over the default policy.
template <typename T, int N>
__device__ __forceinline__ Vec<T, N> load_stream(const T* p) {
    typedef T vt __attribute__((ext_vector_type(N)));
    const vt x = __builtin_nontemporal_load(reinterpret_cast<const vt*>(p));
    Vec<T, N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = x[i];
    return r;
}
template <typename T, int N>
__device__ __forceinline__ void store_stream(T* p, const Vec<T, N>& r) {
    typedef T vt __attribute__((ext_vector_type(N)));
    vt x;
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = r.v[i];
    __builtin_nontemporal_store(x, reinterpret_cast<vt*>(p));
}

// ---------------------------------------------------------------------------
// Register sorting network (ascending).  Replaces insertionSort,
// quorum/majority.go:115-122.
// ---------------------------------------------------------------------------
template <int A, int B, int N>
__device__ __forceinline__ void cex(u64 (&v)[N]) {
    const u64 x = v[A], y = v[B];
    const bool lt = x < y;  // unsigned: Match values may be >= 2^63
    v[A] = lt ? x : y;
    v[B] = lt ? y : x;
}
template <int N, int... K>
__device__ __forceinline__ void sort_impl(u64 (&v)[N], std::integer_sequence<int, K...>) {
    (cex<SortNet<N>::ce[K].a, SortNet<N>::ce[K].b, N>(v), ...);
}
template <int N>
__device__ __forceinline__ void sort_asc(u64 (&v)[N]) {
    if constexpr (SortNet<N>::n > 0) sort_impl<N>(v, std::make_integer_sequence<int, SortNet<N>::n>{});
}

template <int N, int... K>
__device__ __forceinline__ void batcher_impl(u64 (&v)[N], std::integer_sequence<int, K...>) {
    (cex<Batcher<N>::list.ce[K].a, Batcher<N>::list.ce[K].b, N>(v), ...);
}
// Ascending sort of N keys (9..64) in registers: Batcher's odd-even merge sort.
// (Hand-scheduled borrow-chain compare-exchanges in inline asm were tried and
// lost to the compiler's v_cmp_u64 + v_cndmask form on MI355X, tools/cexbench.hip.)
template <int N>
__device__ __forceinline__ void sort_asc_batcher(u64 (&v)[N]) {
    batcher_impl<N>(v, std::make_integer_sequence<int, Batcher<N>::n>{});
}

// The same network on the f64 unit: gfx950 has no 64-bit integer min / max, so an integer compare-exchange is
// v_cmp_u64 + four v_cndmask (5 issue slots), but v_min_f64 + v_max_f64 is 2 — and both return one operand's bits
// unchanged (f64 denormals are preserved in HIP's default mode; no NaN, no -0 is ever fed in).  to_f64_order maps a
// u64 to the f64 bit pattern at the same rank: [0, 0x7FF0...) onto the negative doubles from -inf up to the smallest
// negative denormal, [0x7FF0..., 0xFFE0...] onto +0 .. +inf.  The top 2^53 - 1 values have no pattern left (they
// would be NaNs): the caller checks f64_order_ok over its keys and takes another form for such a wave (no log index
// comes near; the bench distribution, rand.Int63n, stays below 2^63).
constexpr u64 kF64NegInf = 0xFFF0000000000000ull, kF64PosInf = 0x7FF0000000000000ull;
__device__ __forceinline__ u64 to_f64_order(u64 u) {
    return (u32)(u >> 32) < (u32)(kF64PosInf >> 32) ? kF64NegInf - u : u - kF64PosInf;
}
__device__ __forceinline__ u64 from_f64_order(u64 p) { return (long long)p < 0 ? kF64NegInf - p : p + kF64PosInf; }
// max over the keys' high words -> every key has a pattern
__device__ __forceinline__ bool f64_order_ok(u32 top_hi) { return top_hi < 0xFFE00000u; }
// the same bound for a key that is packed as (key << 2) | tag first
constexpr u32 kJointF64KeyHi = 0xFFE00000u >> 2;

template <int A, int B, int N>
__device__ __forceinline__ void cex_f64(u64 (&v)[N]) {
    const double x = __longlong_as_double((long long)v[A]), y = __longlong_as_double((long long)v[B]);
    v[A] = (u64)__double_as_longlong(__builtin_fmin(x, y));
    v[B] = (u64)__double_as_longlong(__builtin_fmax(x, y));
}
template <int N, int... K>
__device__ __forceinline__ void batcher_f64_impl(u64 (&v)[N], std::integer_sequence<int, K...>) {
    (cex_f64<Batcher<N>::list.ce[K].a, Batcher<N>::list.ce[K].b, N>(v), ...);
}
template <int N>
__device__ __forceinline__ void sort_asc_batcher_f64(u64 (&v)[N]) {
    batcher_f64_impl<N>(v, std::make_integer_sequence<int, Batcher<N>::n>{});
}

// Minimal networks up to 8 keys, Batcher's above.
template <int N>
__device__ __forceinline__ void sort_keys(u64 (&v)[N]) {
    if constexpr (N <= 8) {
        sort_asc<N>(v);
    } else {
        sort_asc_batcher<N>(v);
    }
}

// MajorityConfig.CommittedIndex for a config of exactly V voters held in v[]:
// sort, take srt[n-(n/2+1)] — quorum/majority.go:165-171.
template <int V>
__device__ __forceinline__ u64 committed_fixed(u64 (&v)[V]) {
    sort_asc<V>(v);
    return v[V - (V / 2 + 1)];
}

// MajorityConfig.CommittedIndex for the voters selected by slot mask m out of
// S slots.  Non-members are replaced by 0, the smallest key, so after the
// sort the n members occupy the top n places value-wise and the q-th largest
// member sits at ascending position S-q.  n == 0 -> MaxUint64
// (quorum/majority.go:128-132).
template <int S>
__device__ __forceinline__ u64 committed_masked(const u64 (&v)[S], u32 m) {
    m &= (1u << S) - 1u;
    u64 a[S];
#pragma unroll
    for (int s = 0; s < S; ++s) a[s] = ((m >> s) & 1u) ? v[s] : 0ull;
    sort_keys<S>(a);
    const int n = __popc(m);
    const int pos = S - (n / 2 + 1);
    constexpr int kLo = S - (S / 2 + 1);  // smallest position any n can select
    u64 r = a[kLo];
#pragma unroll
    for (int i = kLo + 1; i < S; ++i) r = (pos == i) ? a[i] : r;
    return m == 0 ? kU64Max : r;
}

template <int N, int... K>
__device__ __forceinline__ void sort_f64_impl(u64 (&v)[N], std::integer_sequence<int, K...>) {
    (cex_f64<SortNet<N>::ce[K].a, SortNet<N>::ce[K].b, N>(v), ...);
}
template <int N>
__device__ __forceinline__ void sort_keys_f64(u64 (&v)[N]) {
    if constexpr (N <= 8) {
        if constexpr (SortNet<N>::n > 0) sort_f64_impl<N>(v, std::make_integer_sequence<int, SortNet<N>::n>{});
    } else {
        sort_asc_batcher_f64<N>(v);
    }
}

// committed_masked on the f64 unit (9..16 slots): t[] = to_f64_order of the keys, done once for both halves of a joint
// config; a non-member becomes the pattern of key 0.
template <int S>
__device__ __forceinline__ u64 committed_masked_f64(const u64 (&t)[S], u32 m) {
    m &= (1u << S) - 1u;
    u64 a[S];
#pragma unroll
    for (int s = 0; s < S; ++s) a[s] = ((m >> s) & 1u) ? t[s] : kF64NegInf;
    sort_keys_f64<S>(a);
    const int n = __popc(m);
    const int pos = S - (n / 2 + 1);
    constexpr int kLo = S - (S / 2 + 1);
    u64 r = a[kLo];
#pragma unroll
    for (int i = kLo + 1; i < S; ++i) r = (pos == i) ? a[i] : r;
    return m == 0 ? kU64Max : from_f64_order(r);
}

// JointConfig.CommittedIndex from ONE sort, for keys below 2^62 (every real log index is): the key
// moves up two bits and carries "member of the incoming half" / "of the outgoing half" in the two
// bits it vacates, so the membership rides through the plain u64 network for free; after the sort a
// scan from the largest key down counts members per half and picks each half's q-th largest
// (quorum/joint.go:49-56 over quorum/majority.go:126-172).  Keys of slots in neither half are 0.
// An empty half is never hit and yields MaxUint64, the identity of min.  The caller checks the 2^62
// bound for the whole wavefront and takes another form otherwise (bit-exact either way).
template <int N>
__device__ __forceinline__ u64 joint_from_packed_batcher(u64 (&a)[N], u32 q0, u32 q1) {
    sort_asc_batcher<N>(a);
    // the selected entries stay packed until the end (one shift per half, not one per key)
    u32 c0 = 0, c1 = 0;
    u64 p0 = 0, p1 = 0;  // packed entries carry a non-zero tag: 0 = "no member reached the quorum-th place"
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        const u32 t = (u32)a[i] & 3u, in0 = t & 1u, in1 = t >> 1;
        c0 += in0;
        c1 += in1;
        p0 = (in0 != 0 && c0 == q0) ? a[i] : p0;
        p1 = (in1 != 0 && c1 == q1) ? a[i] : p1;
    }
    const u64 r0 = p0 ? (p0 >> 2) : kU64Max, r1 = p1 ? (p1 >> 2) : kU64Max;
    return r0 < r1 ? r0 : r1;
}
// ---------------------------------------------------------------------------
// Launch shape of the lane-per-group commit kernels (K1/K2).  A workgroup of
// kCommitBlock lanes covers a tile of kTileGroups consecutive groups as kCH
// chunks; in chunk c lane t owns the two groups tile + c*2*BLOCK + 2t, +1, so
// every wave instruction is one contiguous 1 KiB access (16 B per lane) and a
// lane keeps kCH x V such loads in flight.  Chosen by in-process A/B on
// MI355X (tools/kbench.hip): 512 x 2 chunks beat 256 x 1 by 13 % at 2^20
// groups; persistent grid-stride grids were slower.
// ---------------------------------------------------------------------------
constexpr int kCommitBlock = 512;
constexpr int kCH = 2;
constexpr int kTileGroups = kCommitBlock * 2 * kCH;  // 2048; divides kPlaneAlign
static_assert(kPlaneAlign % kTileGroups == 0, "a tile must never run off the end of a plane");

// ---------------------------------------------------------------------------
// Reporting changed (group, commit) pairs, pass 1 of 2 (inside the commit
// kernels).  A tile is split into spans of one wavefront's groups; per span the
// wave stores its __ballot words (which lanes changed), per tile the count of
// changed groups, and per run of kSuperTiles tiles one atomic add of that count.
// Nothing else: the pairs themselves are written by k_pack_changed, which finds
// every tile's final offset from those counts and reads the new commit value
// back from the commit plane — so a pair costs its 16 output bytes plus one
// 8-byte re-read, instead of a staged copy (16 written + 16 read + 16 written).
// No atomics order the output: it is sorted by group and reproducible.
// ---------------------------------------------------------------------------
constexpr u32 kSuperTiles = 64;

struct Report {
    u64* __restrict__ bits;    // ballot words: paired tiles 2 per span of 128 groups, lane tiles 1 per span of 64
    u32* __restrict__ counts;  // [tiles] changed groups per tile
    u32* __restrict__ super;   // [ceil(tiles / kSuperTiles)] sums of counts; zero before the pass
    // The engine keeps two `super` arrays and alternates: a pass adds into one and, on the side, zeroes the
    // other for its successor, which saves the memset launch in front of every pass.  NULL: the caller zeroes.
    u32* __restrict__ super_next;
    u32 super_words;
};

// Workgroup 0 zeroes the other `super` array (no pass is reading it: passes of one engine are stream-ordered).
__device__ __forceinline__ void zero_next_super(const Report& rep, u32 tile) {
    if (tile == 0 && rep.super_next)
        for (u32 i = threadIdx.x; i < rep.super_words; i += blockDim.x) rep.super_next[i] = 0;
}

// Lane-pair tiles (K1/K2): in chunk c lane t of wave w owns groups span*128 + 2t, +1
// with span = c*NW + w; word 2*span holds the ballot of the even groups, 2*span+1
// of the odd ones.
template <int CH, int BLOCK>
__device__ __forceinline__ void emit_changed(const bool (&ch)[CH][2], Report rep, u32 tile) {  // tile: index within the engine
    constexpr int NW = BLOCK / 64;
    __shared__ u32 wave_tot[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    zero_next_super(rep, tile);
    u32 tot = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const u64 b0 = __ballot(ch[c][0]), b1 = __ballot(ch[c][1]);
        tot += __popcll(b0) + __popcll(b1);
        if (lane == 0) {
            Vec<u64, 2> w;
            w.v[0] = b0;
            w.v[1] = b1;
            store_vec<u64, 2>(rep.bits + ((u64)tile * (CH * NW) + (u64)(c * NW + wave)) * 2, w);
        }
    }
    if (lane == 0) wave_tot[wave] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) run += wave_tot[w];
        rep.counts[tile] = run;
        if (run) atomicAdd(rep.super + tile / kSuperTiles, run);
    }
}

// What a commit pass does with a group's quorum commit index (mci):
//   kStore    commit[g] = mci                                  (ProgressTracker.Committed)
//   kChanged  same, and report (g, mci) where mci != old commit[g]
//   kAdvance  raftLog.maybeCommit fused in (log.go:447-456 via raft.go:755-758):
//             commit[g] is raftLog.committed; it moves to mci, and (g, mci) is
//             reported, iff mci > commit[g] and the entry at mci carries the
//             leader's term, i.e. term_start[g] <= mci <= last_index[g]
//             (log terms never decrease and the leader appends only in its own
//             term, so the entries of the current term are exactly that range;
//             beyond last_index raftLog.term is ErrUnavailable -> 0, log.go:381-407).
constexpr int kStore = 0, kChanged = 1, kAdvance = 2;

struct LogBounds {  // per-group planes used by kAdvance only
    const u64* __restrict__ term_start;
    const u64* __restrict__ last_index;
};

// Store the tile's results; report changed / advanced pairs as MODE says.
template <int CH, int BLOCK, int MODE>
__device__ __forceinline__ void finish_tile(const Vec<u64, 2> (&res)[CH], u64 base, u64 G,
                                            u64* __restrict__ commit, LogBounds lb, Report rep, u32 tile) {
    if constexpr (MODE != kStore) {
        bool ch[CH][2];
        Vec<u64, 2> nxt[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const u64 g0 = base + (u64)c * BLOCK * 2;
            const Vec<u64, 2> old = load_stream<u64, 2>(commit + g0);
            Vec<u64, 2> ts, li;
            if constexpr (MODE == kAdvance) {
                ts = load_stream<u64, 2>(lb.term_start + g0);
                li = load_stream<u64, 2>(lb.last_index + g0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u64 r = res[c].v[j];
                if constexpr (MODE == kAdvance) {
                    ch[c][j] = (g0 + j < G) && (r > old.v[j]) && (r >= ts.v[j]) && (r <= li.v[j]);
                    nxt[c].v[j] = ch[c][j] ? r : old.v[j];
                } else {
                    ch[c][j] = (g0 + j < G) && (old.v[j] != r);
                    nxt[c].v[j] = r;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) store_stream<u64, 2>(commit + base + (u64)c * BLOCK * 2, nxt[c]);
        emit_changed<CH, BLOCK>(ch, rep, tile);
    } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) store_stream<u64, 2>(commit + base + (u64)c * BLOCK * 2, res[c]);
    }
}

// ---------------------------------------------------------------------------
// K1 — ProgressTracker.Committed, uniform non-joint config of V voters.
// ---------------------------------------------------------------------------
template <int V, int MODE>
__global__ __launch_bounds__(kCommitBlock) void k_commit_uniform(const u64* __restrict__ match,
                                                                 u64 stride, u64 G,
                                                                 u64* __restrict__ commit, LogBounds lb,
                                                                 Report rep) {
    const u64 base = (u64)blockIdx.x * kTileGroups + threadIdx.x * 2;
    Vec<u64, 2> in[kCH][V];
#pragma unroll
    for (int c = 0; c < kCH; ++c)
#pragma unroll
        for (int s = 0; s < V; ++s)
            in[c][s] = load_stream<u64, 2>(match + (u64)s * stride + base + (u64)c * kCommitBlock * 2);
    Vec<u64, 2> res[kCH];
#pragma unroll
    for (int c = 0; c < kCH; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u64 a[V];
#pragma unroll
            for (int s = 0; s < V; ++s) a[s] = in[c][s].v[j];
            res[c].v[j] = committed_fixed<V>(a);
        }
    finish_tile<kCH, kCommitBlock, MODE>(res, base, G, commit, lb, rep, blockIdx.x);
}

// ---------------------------------------------------------------------------
// K1m — the same pass over the shards of several engines of one GPU in ONE
// launch (rafthip_commit_many).  A 2^20-group shard is a single wave of 512
// workgroups and ends after 9 us, as long as the 1.5-1.9 us boundary between two
// dependent kernels is a fifth of its run; a grid that spans the shards pays
// that boundary once.  The shard table travels by value in the kernel arguments
// (no device allocation); a workgroup finds its shard with a scalar search over
// at most kManyMax tile offsets.
// ---------------------------------------------------------------------------
constexpr int kManyMax = 32;
struct ManyShards {
    u32 n;
    u32 first_tile[kManyMax + 1];  // first_tile[k] = tiles of shards 0..k-1
    const u64* match[kManyMax];
    u64* commit[kManyMax];
    u64 stride[kManyMax];
};

template <int V>
__global__ __launch_bounds__(kCommitBlock) void k_commit_uniform_many(const ManyShards sh) {
    u32 k = 0;
    for (u32 i = 1; i < sh.n; ++i) k = blockIdx.x >= sh.first_tile[i] ? i : k;  // wave-uniform: scalar loads + compares
    const u64* __restrict__ match = sh.match[k];
    u64* __restrict__ commit = sh.commit[k];
    const u64 stride = sh.stride[k];
    const u64 base = (u64)(blockIdx.x - sh.first_tile[k]) * kTileGroups + threadIdx.x * 2;
    Vec<u64, 2> in[kCH][V];
#pragma unroll
    for (int c = 0; c < kCH; ++c)
#pragma unroll
        for (int s = 0; s < V; ++s)
            in[c][s] = load_stream<u64, 2>(match + (u64)s * stride + base + (u64)c * kCommitBlock * 2);
#pragma unroll
    for (int c = 0; c < kCH; ++c) {
        Vec<u64, 2> res;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u64 a[V];
#pragma unroll
            for (int s = 0; s < V; ++s) a[s] = in[c][s].v[j];
            res.v[j] = committed_fixed<V>(a);
        }
        store_stream<u64, 2>(commit + base + (u64)c * kCommitBlock * 2, res);
    }
}

// K1m in a reporting mode (rafthip_commit_report_many): the shards' changed / advanced pairs are reported
// per shard, exactly as the single-engine pass reports them — every shard has its own ballot words, counts
// and sums, and the tile index that addresses them is the tile's index within its shard.
struct ManyReport {
    u32 n;
    u32 first_tile[kManyMax + 1];
    const u64* match[kManyMax];
    u64* commit[kManyMax];
    u64 stride[kManyMax];
    u64 groups[kManyMax];
    const u64* term_start[kManyMax];  // kAdvance only
    const u64* last_index[kManyMax];
    u64* bits[kManyMax];
    u32* counts[kManyMax];
    u32* super[kManyMax];
    u32* super_next[kManyMax];
    u32 super_words[kManyMax];
};

template <int V, int MODE>
__global__ __launch_bounds__(kCommitBlock) void k_commit_uniform_many_report(const ManyReport sh) {
    static_assert(MODE != kStore, "the plain pass is k_commit_uniform_many");
    u32 k = 0;
    for (u32 i = 1; i < sh.n; ++i) k = blockIdx.x >= sh.first_tile[i] ? i : k;
    const u64* __restrict__ match = sh.match[k];
    const u64 stride = sh.stride[k];
    const u32 tile = blockIdx.x - sh.first_tile[k];
    const u64 base = (u64)tile * kTileGroups + threadIdx.x * 2;
    Vec<u64, 2> in[kCH][V];
#pragma unroll
    for (int c = 0; c < kCH; ++c)
#pragma unroll
        for (int s = 0; s < V; ++s)
            in[c][s] = load_stream<u64, 2>(match + (u64)s * stride + base + (u64)c * kCommitBlock * 2);
    Vec<u64, 2> res[kCH];
#pragma unroll
    for (int c = 0; c < kCH; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u64 a[V];
#pragma unroll
            for (int s = 0; s < V; ++s) a[s] = in[c][s].v[j];
            res[c].v[j] = committed_fixed<V>(a);
        }
    finish_tile<kCH, kCommitBlock, MODE>(res, base, sh.groups[k], sh.commit[k], LogBounds{sh.term_start[k], sh.last_index[k]},
                                         Report{sh.bits[k], sh.counts[k], sh.super[k], sh.super_next[k], sh.super_words[k]}, tile);
}

// ---------------------------------------------------------------------------
// K2 — ProgressTracker.Committed with per-group joint configs, S <= 16 slots:
// min(CommittedIndex(incoming), CommittedIndex(outgoing)), quorum/joint.go:49-56.
// Both halves are sorted from the same loaded keys.  S <= 8: 512 lanes x 2
// chunks and one mask byte per group; 9..16: 256 lanes x 1 chunk (4S VGPRs of
// keys per lane) and 16-bit masks; a uniform config of more than 8 voters
// (in_mask == NULL) runs here too.
// ---------------------------------------------------------------------------
// slots from which K2 sorts on the f64 unit (same box, half of the groups joint, 2^24 groups: 4 slots 113 -> 115 us,
// 5 slots 137 -> 135, 6 slots -4 %, 8 slots 213 -> 209: the transform pays from six keys on)
constexpr int kMaskedF64From = 6;
// One tile of K2: res[c].v[j] = the joint commit index of group base + c*2*BLOCK + j.
template <int S, typename MaskT, int BLOCK, int CH>
__device__ __forceinline__ void masked_tile(const u64* __restrict__ match, u64 stride, const MaskT* __restrict__ in_mask,
                                            const MaskT* __restrict__ out_mask, u32 uniform_voters, u64 base,
                                            Vec<u64, 2> (&res)[CH]) {
    Vec<u64, 2> in[CH][S];
    Vec<MaskT, 2> m0[CH], m1[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const u64 g0 = base + (u64)c * BLOCK * 2;
#pragma unroll
        for (int s = 0; s < S; ++s) in[c][s] = load_stream<u64, 2>(match + (u64)s * stride + g0);
        if (in_mask) {
            m0[c] = load_stream<MaskT, 2>(in_mask + g0);
            m1[c] = load_stream<MaskT, 2>(out_mask + g0);
        } else {
            m0[c].v[0] = m0[c].v[1] = (MaskT)((1u << uniform_voters) - 1u);
            m1[c].v[0] = m1[c].v[1] = 0;
        }
    }
    bool any_out = false;
#pragma unroll
    for (int c = 0; c < CH; ++c) any_out |= (m1[c].v[0] | m1[c].v[1]) != 0;
    // Joint configs are transient (confchange.go:51-126); when no lane of the
    // wavefront holds one, the outgoing half is skipped for the whole wave.
    const bool wave_joint = __any(any_out);
    // (Both halves from one sort with the membership in the key's low bits — regs_joint_packed — was tried
    // here as well: no gain at 9..16 slots, where the two sorts hide behind the loads, and at <= 8 slots the
    // extra code cost two waves per SIMD of occupancy.  Measured, not kept: profiles/r02_b_kernels.md.)
    if constexpr (S >= kMaskedF64From) {
        // The network on the f64 unit (K4r's, kernels.h to_f64_order) unless some key of the wavefront, member or not,
        // lies in the top 2^53 and has no pattern: the keys are transformed once, both halves of a joint group sort
        // with v_min_f64 / v_max_f64.  Same box, A/B/A/B: cfg3's joint mix at 6 slots 164 -> 158 us per 2^24 groups,
        // twelve shards of it in one launch 124 -> 120, one launch each 186 -> 167; 9 slots 60 / 61 either way; at 16
        // slots a joint group costs no more than a plain one now (112 -> 94 us per 2^22 groups).
        u32 top = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const u32 h0 = (u32)(in[c][s].v[0] >> 32), h1 = (u32)(in[c][s].v[1] >> 32);
                top = h0 > top ? h0 : top;
                top = h1 > top ? h1 : top;
            }
        if (__all(f64_order_ok(top))) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    u64 t[S];
#pragma unroll
                    for (int s = 0; s < S; ++s) t[s] = to_f64_order(in[c][s].v[j]);
                    u64 r = committed_masked_f64<S>(t, m0[c].v[j]);
                    if (wave_joint) {
                        const u64 r1 = committed_masked_f64<S>(t, m1[c].v[j]);
                        r = r1 < r ? r1 : r;
                    }
                    res[c].v[j] = r;
                }
            return;
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u64 a[S];
#pragma unroll
            for (int s = 0; s < S; ++s) a[s] = in[c][s].v[j];
            u64 r = committed_masked<S>(a, m0[c].v[j]);
            if (wave_joint) {
                const u64 r1 = committed_masked<S>(a, m1[c].v[j]);
                r = r1 < r ? r1 : r;
            }
            res[c].v[j] = r;
        }
}

template <int S, typename MaskT, int MODE, int BLOCK, int CH>
__global__ __launch_bounds__(BLOCK) void k_commit_masked(const u64* __restrict__ match, u64 stride, u64 G,
                                                         const MaskT* __restrict__ in_mask,
                                                         const MaskT* __restrict__ out_mask,
                                                         u32 uniform_voters, u64* __restrict__ commit,
                                                         LogBounds lb, Report rep) {
    const u64 base = (u64)blockIdx.x * (BLOCK * 2 * CH) + threadIdx.x * 2;
    Vec<u64, 2> res[CH];
    masked_tile<S, MaskT, BLOCK, CH>(match, stride, in_mask, out_mask, uniform_voters, base, res);
    finish_tile<CH, BLOCK, MODE>(res, base, G, commit, lb, rep, blockIdx.x);
}

// K2m — K2 (S <= 8) over the shards of several engines in one launch: the shards of a host in the middle of
// a conf change carry per-group masks (joint configs, confchange.go:51-126) and would otherwise fall out of
// rafthip_commit_many / _report_many to one launch each.  A shard without mask planes (in_mask NULL) is a
// uniform one: voters = slots [0, uniform_voters).  Same tile shape as K1m, so the same packing pass follows.
struct ManyMasks {
    const uint8_t* in_mask[kManyMax];
    const uint8_t* out_mask[kManyMax];
    uint8_t uniform_voters[kManyMax];
};

template <int S>
__global__ __launch_bounds__(kCommitBlock) void k_commit_masked_many(const ManyShards sh, const ManyMasks mk) {
    u32 k = 0;
    for (u32 i = 1; i < sh.n; ++i) k = blockIdx.x >= sh.first_tile[i] ? i : k;
    u64* __restrict__ commit = sh.commit[k];
    const u64 base = (u64)(blockIdx.x - sh.first_tile[k]) * kTileGroups + threadIdx.x * 2;
    Vec<u64, 2> res[kCH];
    masked_tile<S, uint8_t, kCommitBlock, kCH>(sh.match[k], sh.stride[k], mk.in_mask[k], mk.out_mask[k], mk.uniform_voters[k], base, res);
#pragma unroll
    for (int c = 0; c < kCH; ++c) store_stream<u64, 2>(commit + base + (u64)c * kCommitBlock * 2, res[c]);
}

template <int S, int MODE>
__global__ __launch_bounds__(kCommitBlock) void k_commit_masked_many_report(const ManyReport sh, const ManyMasks mk) {
    static_assert(MODE != kStore, "the plain pass is k_commit_masked_many");
    u32 k = 0;
    for (u32 i = 1; i < sh.n; ++i) k = blockIdx.x >= sh.first_tile[i] ? i : k;
    const u32 tile = blockIdx.x - sh.first_tile[k];
    const u64 base = (u64)tile * kTileGroups + threadIdx.x * 2;
    Vec<u64, 2> res[kCH];
    masked_tile<S, uint8_t, kCommitBlock, kCH>(sh.match[k], sh.stride[k], mk.in_mask[k], mk.out_mask[k], mk.uniform_voters[k], base, res);
    finish_tile<kCH, kCommitBlock, MODE>(res, base, sh.groups[k], sh.commit[k], LogBounds{sh.term_start[k], sh.last_index[k]},
                                         Report{sh.bits[k], sh.counts[k], sh.super[k], sh.super_next[k], sh.super_words[k]}, tile);
}
constexpr int kMidBlock = 256;                 // tile shape of K2 at 9..16 slots
constexpr int kMidTileGroups = kMidBlock * 2;  // 512; divides kPlaneAlign

// Tail of the one-group-per-lane kernels (block = kBlock lanes = 4 waves):
// store / diff / term-guard as MODE says, with the same ordered in-block
// compaction of reported pairs as emit_changed.
template <int MODE>
__device__ __forceinline__ void finish_lane(u64 res, u64 g, u64 G, u64* __restrict__ commit, LogBounds lb, Report rep) {
    if constexpr (MODE != kStore) {
        __shared__ u32 wave_tot[kBlock / 64];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const u64 old = commit[g];
        bool chg;
        if constexpr (MODE == kAdvance) {
            chg = (g < G) && (res > old) && (res >= lb.term_start[g]) && (res <= lb.last_index[g]);
            if (chg) commit[g] = res;
        } else {
            chg = (g < G) && (old != res);
            commit[g] = res;
        }
        const u64 b = __ballot(chg);
        zero_next_super(rep, blockIdx.x);
        if (lane == 0) {
            rep.bits[(u64)blockIdx.x * (kBlock / 64) + wave] = b;  // one span of 64 groups per wave
            wave_tot[wave] = __popcll(b);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            u32 all = 0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) all += wave_tot[w];
            rep.counts[blockIdx.x] = all;
            if (all) atomicAdd(rep.super + blockIdx.x / kSuperTiles, all);
        }
    } else {
        commit[g] = res;
    }
}

// ---------------------------------------------------------------------------
// K4r — wide voter sets (9..64 slots), one group per lane, all keys in
// registers: the N-key Batcher network above (63 / 191 / 543 compare-exchanges
// at N = 16 / 32 / 64) sorts the masked keys; position N-q is the q-th largest
// member.  Every compare-exchange is done once by one lane (the cross-lane
// bitonic network of K4 does each twice and pays a DPP / LDS-crossbar move per
// step), loads are the same coalesced plane streams as K1, and there is no LDS
// staging.  A joint group costs a second pass (keys re-read; first pass leaves
// them in L2); a wave with no joint group skips it.
// ---------------------------------------------------------------------------
// a u64 in global memory: global_load with an SGPR base.  (Against the flat loads the compiler emits when it has lost
// track of the address space, same box, A/B/A/B: within the 5 % the runs scatter by — kept because it is what the
// comments below say.)
typedef __attribute__((address_space(1))) u64 GlobalU64;

template <int N, bool NT>
__device__ __forceinline__ bool regs_half(const u64* __restrict__ match, u64 stride, int S, u64 m, u64& out) {
    u64 a[N];
    u32 top = 0;
    // wave-uniform base + small per-lane offset: the N addresses stay in SGPRs
    const u64* base = match + (u64)blockIdx.x * kBlock;
    const u32 t = threadIdx.x;
#pragma unroll
    for (int s = 0; s < N; ++s) {
        // rows >= S do not exist: re-read row 0 (a cache hit; m has no bit there)
        const u64* p = base + (u64)(s < S ? s : 0) * stride;
        // keep the row base an SGPR pair (saddr + lane-offset addressing: no VGPR
        // pair per row), and opaque, so the joint loop re-loads instead of keeping
        // a second copy of every key live
        asm volatile("" : "+s"(p));
        const GlobalU64* gp = (const GlobalU64*)p;  // behind the asm the compiler no longer knows: flat loads otherwise
        a[s] = NT ? __builtin_nontemporal_load(gp + t) : gp[t];
    }
    // One key at a time, each chained to the next through an empty asm, `top` included: left to itself the compiler
    // sinks the max over the raw keys below the sort, keeps every raw key for it, and the 64-key kernel spills them
    // one by one between its loads (1122 us against 587).
#pragma unroll
    for (int s = 0; s < N; ++s) {
        // non-members become key 0; arithmetic mask (no per-key lane-mask SGPR pair)
        const u64 key = a[s] & (0ull - ((m >> s) & 1ull));
        const u32 hi = (u32)(key >> 32);
        top = hi > top ? hi : top;
        a[s] = to_f64_order(key);
        if (s + 1 < N) {
            asm volatile("" : "+v"(a[s]), "+v"(a[s + 1]), "+v"(top));
        } else {
            asm volatile("" : "+v"(a[s]), "+v"(top));
        }
    }
    sort_asc_batcher_f64<N>(a);
    const int n = __popcll(m);
    const int pos = N - (n / 2 + 1);
    constexpr int kLo = N - (N / 2 + 1);
    u64 r = a[kLo];
#pragma unroll
    for (int i = kLo + 1; i < N; ++i) r = (pos == i) ? a[i] : r;
    out = n == 0 ? kU64Max : from_f64_order(r);
    // tested AFTER the sort, like regs_joint_packed's bound: a wave with a key in the top 2^53 has sorted in vain
    return __all(f64_order_ok(top));
}

// joint_from_packed_batcher on the f64 unit: a[] holds to_f64_order of the packed entries.  Of a pattern below zero
// (an entry below 0x7FF0...) the two low bits are those of MINUS the entry: tag 1 and tag 3 change places, so the
// incoming bit is read as it is and the outgoing one is flipped where both low bits say "incoming".
template <int N>
__device__ __forceinline__ u64 joint_from_packed_f64(u64 (&a)[N], u32 q0, u32 q1) {
    sort_asc_batcher_f64<N>(a);
    u32 c0 = 0, c1 = 0;
    u64 p0 = 0, p1 = 0;  // a member's pattern is never 0 (its tag is not): 0 = "no member reached the quorum-th place"
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        const u32 lo = (u32)a[i], neg = (u32)(a[i] >> 63);
        const u32 in0 = lo & 1u, in1 = ((lo >> 1) ^ (neg & lo)) & 1u;
        c0 += in0;
        c1 += in1;
        p0 = (in0 != 0 && c0 == q0) ? a[i] : p0;
        p1 = (in1 != 0 && c1 == q1) ? a[i] : p1;
    }
    const u64 r0 = p0 ? (from_f64_order(p0) >> 2) : kU64Max, r1 = p1 ? (from_f64_order(p1) >> 2) : kU64Max;
    return r0 < r1 ? r0 : r1;
}

// 48 / 64 keys: ask for 3 waves per SIMD (<= 168 VGPRs); left alone the
// compiler takes 186 and drops to 2 (measured 632 -> 576 us at 2^22 x 64).
// Both halves of a joint config from ONE plain sort: joint_from_packed's tags in
// the two low bits of the shifted key.  (Round 1 carried the tags in a second
// register per key: 7 operations per compare-exchange against 5, and above 32
// keys the tag registers spilled, so those waves sorted twice and re-read the
// keys: 977 us against 547 us non-joint at 2^22 x 64.)  Returns false — wave-
// uniformly — when some member key is 0x3FF8000000000000 (2^62 - 2^51) or more: the packed entry would have no f64
// pattern (to_f64_order); the caller then sorts once per half (regs_half).
template <int N>
__device__ __forceinline__ bool regs_joint_packed(const u64* __restrict__ match, u64 stride, int S, u64 m0, u64 m1,
                                                  u64& out) {
    u64 a[N];
    const u64* base = match + (u64)blockIdx.x * kBlock;
    const u32 t = threadIdx.x;
    u32 top = 0;
#pragma unroll
    for (int s = 0; s < N; ++s) {
        const u64* p = base + (u64)(s < S ? s : 0) * stride;  // rows >= S: re-read row 0, tag 0 (masks have no bit there)
        asm volatile("" : "+s"(p));
        const GlobalU64* gp = (const GlobalU64*)p;
        a[s] = __builtin_nontemporal_load(gp + t);
    }
#pragma unroll
    for (int s = 0; s < N; ++s) {  // one key at a time, as in regs_half
        const u64 tag = ((m0 >> s) & 1ull) | (((m1 >> s) & 1ull) << 1);
        const u64 key = a[s] & (0ull - (u64)(tag != 0));
        const u32 hi = (u32)(key >> 32);
        top = hi > top ? hi : top;
        a[s] = to_f64_order((key << 2) | tag);
        if (s + 1 < N) {
            asm volatile("" : "+v"(a[s]), "+v"(a[s + 1]), "+v"(top));
        } else {
            asm volatile("" : "+v"(a[s]), "+v"(top));
        }
    }
    // the bound is tested AFTER the sort: a branch where every key is live costs a hundred spilled registers;
    // a wave that fails it (no real log does) has sorted in vain
    out = joint_from_packed_f64<N>(a, (u32)__popcll(m0) / 2 + 1, (u32)__popcll(m1) / 2 + 1);
    return __all(top < kJointF64KeyHi);
}

// A wave with member keys beyond the one-sort forms' bounds (no real log index is): both halves by counting, in
// rolled loops that hold no key array — the q-th largest member x of a half is the member with
// #(members > x) < q <= #(members >= x).  S*S loads per lane, all cache hits; slow and never taken in
// practice, but the kernel stays bit-exact for every u64 (and keeps no second sorting network alive
// beside the one-sort path, which cost 100 spilled registers).
__device__ __forceinline__ u64 joint_by_counting(const u64* __restrict__ match, u64 stride, int S, u64 g, u64 m0, u64 m1) {
    const u32 q0 = (u32)__popcll(m0) / 2 + 1, q1 = (u32)__popcll(m1) / 2 + 1;
    u64 r0 = kU64Max, r1 = kU64Max;  // an empty half: MaxUint64, the identity of min
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        if (!(((m0 | m1) >> s) & 1ull)) continue;
        const u64 x = match[(u64)s * stride + g];
        u32 gt0 = 0, ge0 = 0, gt1 = 0, ge1 = 0;
#pragma unroll 1
        for (int t = 0; t < S; ++t) {
            const u64 y = match[(u64)t * stride + g];
            const u32 in0 = (u32)((m0 >> t) & 1ull), in1 = (u32)((m1 >> t) & 1ull);
            gt0 += in0 & (u32)(y > x);
            ge0 += in0 & (u32)(y >= x);
            gt1 += in1 & (u32)(y > x);
            ge1 += in1 & (u32)(y >= x);
        }
        if (((m0 >> s) & 1ull) && gt0 < q0 && q0 <= ge0) r0 = x;
        if (((m1 >> s) & 1ull) && gt1 < q1 && q1 <= ge1) r1 = x;
    }
    return r0 < r1 ? r0 : r1;
}

// (24 / 32 keys asked to fit 5 or 6 waves per SIMD spill a little and lose a quarter: 192 -> 250 / 238 us at S = 32.)
template <int N, typename MaskT, int MODE>
__global__ __launch_bounds__(kBlock, (N > 32 ? 3 : 1)) void k_commit_regs(const u64* __restrict__ match, u64 stride, u64 G,
                                                        int S, const MaskT* __restrict__ in_mask,
                                                        const MaskT* __restrict__ out_mask,
                                                        int uniform_voters, u64* __restrict__ commit,
                                                        LogBounds lb, Report rep) {
    const u64 g = (u64)blockIdx.x * kBlock + threadIdx.x;  // < stride (stride is a multiple of 4096)
    u64 m0, m1;
    if (in_mask) {
        m0 = (u64)in_mask[g];
        m1 = (u64)out_mask[g];
    } else {
        m0 = uniform_voters >= 64 ? kU64Max : ((1ull << uniform_voters) - 1ull);
        m1 = 0;
    }
    if (S < 64) {
        const u64 valid = (1ull << S) - 1ull;
        m0 &= valid;
        m1 &= valid;
    }
    u64 res;
    const bool joint = __any(m1 != 0);
    // Joint wave: one sort serves both halves when every member key is below 2^62 - 2^51 (regs_joint_packed).
    bool per_half = true;
    if (joint) per_half = !regs_joint_packed<N>(match, stride, S, m0, m1, res);
    if (per_half) {
        // A wave with no joint group, or a joint one with keys beyond that bound (rand.Int63n's, bench_test.go:32, are):
        // one sort per half in a ROLLED loop — a single network in the code, keys re-loaded per half (the second read
        // hits L2).  An empty half yields MaxUint64, the identity of min.  Keys in the top 2^53: by counting.
        u64 r = kU64Max;
        bool ok = true;
        const int halves = joint ? 2 : 1;
#pragma unroll 1
        for (int h = 0; h < halves; ++h) {
            u64 x;
            ok &= regs_half<N, true>(match, stride, S, h ? m1 : m0, x);
            r = x < r ? x : r;
        }
        res = ok ? r : joint_by_counting(match, stride, S, g, m0, m1);
    }
    finish_lane<MODE>(res, g, G, commit, lb, rep);
}

// ---------------------------------------------------------------------------
// K4 — wide voter sets (9..64 slots).  A wavefront stages a 64-group tile of
// every plane in LDS (coalesced reads), then turns the tile: lane = slot.
// W = 16/32/64 lanes form one group's sort segment, so 64/W groups are sorted
// per step by a cross-lane bitonic network (DPP for distance 1-2, ds_swizzle
// for 4-16, a permute for 32).
// ---------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ u32 xor_lane32(u32 x) {
    if constexpr (J == 1) {
        return (u32)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    } else if constexpr (J == 2) {
        return (u32)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    } else if constexpr (J <= 16) {
        return (u32)__builtin_amdgcn_ds_swizzle((int)x, (J << 10) | 0x1F);  // bit mode: lane ^ J
    } else {
        return (u32)__shfl_xor((int)x, J, 64);
    }
}
template <int J>
__device__ __forceinline__ u64 xor_lane64(u64 x) {
    const u32 lo = xor_lane32<J>((u32)x), hi = xor_lane32<J>((u32)(x >> 32));
    return ((u64)hi << 32) | lo;
}

template <int K, int J>
__device__ __forceinline__ u64 bitonic_step(u64 key, int sl) {
    const u64 other = xor_lane64<J>(key);
    const bool asc = (sl & K) == 0;
    const bool lower = (sl & J) == 0;
    const bool keep_min = (lower == asc);
    const bool lt = key < other;
    return (keep_min == lt) ? key : other;
}
template <int K, int J>
__device__ __forceinline__ u64 bitonic_merge(u64 key, int sl) {
    key = bitonic_step<K, J>(key, sl);
    if constexpr (J > 1) key = bitonic_merge<K, J / 2>(key, sl);
    return key;
}
template <int W, int K>
__device__ __forceinline__ u64 bitonic_sort(u64 key, int sl) {
    key = bitonic_merge<K, K / 2>(key, sl);
    if constexpr (K < W) key = bitonic_sort<W, K * 2>(key, sl);
    return key;
}

// q-th largest of the member keys of one W-lane segment (ascending sort,
// position W-q); n == 0 -> MaxUint64.
template <int W>
__device__ __forceinline__ u64 wide_half(u64 val, u64 mask, int sl, int seg_base) {
    const u64 key = ((mask >> sl) & 1ull) ? val : 0ull;
    const u64 sorted = bitonic_sort<W, 2>(key, sl);
    const int n = __popcll(mask);
    const int src = seg_base + (W - (n / 2 + 1));
    const u32 lo = (u32)__shfl((int)(u32)sorted, src, 64);
    const u32 hi = (u32)__shfl((int)(u32)(sorted >> 32), src, 64);
    const u64 r = ((u64)hi << 32) | lo;
    return n == 0 ? kU64Max : r;
}

constexpr int kTileStride = 65;  // u64 per LDS tile row: 64 groups + 1 pad (bank spread)

template <int W, typename MaskT, int MODE>
__global__ __launch_bounds__(kBlock) void k_commit_wide(const u64* __restrict__ match, u64 stride,
                                                        u64 G, int S, const MaskT* __restrict__ in_mask,
                                                        const MaskT* __restrict__ out_mask,
                                                        int uniform_voters, u64* __restrict__ commit,
                                                        LogBounds lb, Report rep) {
    extern __shared__ u64 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64* tile = lds + (size_t)wave * (size_t)S * kTileStride;
    const u64 g = (u64)blockIdx.x * kBlock + threadIdx.x;  // < stride (stride is a multiple of 4096)
    for (int s = 0; s < S; ++s) tile[s * kTileStride + lane] = __builtin_nontemporal_load(match + (u64)s * stride + g);
    u64 m0, m1;
    if (in_mask) {
        m0 = (u64)in_mask[g];
        m1 = (u64)out_mask[g];
    } else {
        m0 = uniform_voters >= 64 ? kU64Max : ((1ull << uniform_voters) - 1ull);
        m1 = 0;
    }
    if (S < 64) {
        const u64 valid = (1ull << S) - 1ull;
        m0 &= valid;
        m1 &= valid;
    }
    __syncthreads();
    constexpr int GPI = 64 / W;  // groups sorted per step
    const int sub = lane / W, sl = lane % W, seg_base = sub * W;
    const bool wave_joint = __any(m1 != 0);
    u64 res = 0;
#pragma unroll 1
    for (int it = 0; it < 64 / GPI; ++it) {
        const int gl = it * GPI + sub;  // tile-local group this segment sorts
        const u64 val = (sl < S) ? tile[sl * kTileStride + gl] : 0ull;
        const u64 mm0 = ((u64)(u32)__shfl((int)(u32)(m0 >> 32), gl, 64) << 32) |
                        (u64)(u32)__shfl((int)(u32)m0, gl, 64);
        u64 r = wide_half<W>(val, mm0, sl, seg_base);
        if (wave_joint) {
            const u64 mm1 = ((u64)(u32)__shfl((int)(u32)(m1 >> 32), gl, 64) << 32) |
                            (u64)(u32)__shfl((int)(u32)m1, gl, 64);
            const u64 r1 = wide_half<W>(val, mm1, sl, seg_base);
            r = r1 < r ? r1 : r;
        }
        // lane L owns group L of the tile: sorted in step L/GPI by segment L%GPI
        const int owner_seg = (lane % GPI) * W;
        const u64 rr = ((u64)(u32)__shfl((int)(u32)(r >> 32), owner_seg, 64) << 32) |
                       (u64)(u32)__shfl((int)(u32)r, owner_seg, 64);
        if (lane / GPI == it) res = rr;
    }
    finish_lane<MODE>(res, g, G, commit, lb, rep);
}

// ---------------------------------------------------------------------------
// Reporting changed pairs, pass 2 of 2: one workgroup per tile of the commit
// pass.  A tile without a changed group leaves at once.  The others find their
// offset in the output — the super counts before their run of kSuperTiles tiles
// plus the tile counts before them inside it, one wave reduction — and each wave
// writes the pairs of its spans in group order: (group_base + g, commit[g]) for
// every set ballot bit, positions by popcount prefix.  The last tile also stores
// the total.  Pairs beyond max_pairs are counted, not written.
//   PAIRED  spans of 128 groups, two ballot words (even / odd groups)  — K1, K2
//   else    spans of 64 groups, one ballot word                        — K4r, K4
// ---------------------------------------------------------------------------
constexpr u32 kMaxSpans = 16;  // 2048-group tile of K1/K2

template <bool PAIRED>
__device__ __forceinline__ void pack_tile(u32 tile, const u64* __restrict__ bits, const u32* __restrict__ counts,
                                          const u32* __restrict__ super, u32 ntiles, u32 spans,
                                          const u64* __restrict__ commit, u64 group_base, Pair* __restrict__ out,
                                          u64 max_pairs, u64* __restrict__ total) {
    const u32 n = counts[tile];
    const bool last = tile == ntiles - 1;
    if (n == 0 && !last) return;  // uniform over the workgroup
    __shared__ u64 s_off;
    __shared__ u32 s_cnt[kMaxSpans];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr u32 W = PAIRED ? 2 : 1;
    constexpr u32 NWV = kBlock / 64, kMine = kMaxSpans / NWV;  // spans per wave: wave w owns spans w, w + NWV, ...
    const u64* tb = bits + (u64)tile * spans * W;
    // Everything that does not depend on the tile's offset is asked for first — the ballot words of this
    // wave's spans and the new commit values of their changed groups — so that those loads are in flight
    // while wave 0 adds up the counts before the tile.
    u64 b0[kMine], b1[kMine];
    Vec<u64, 2> val[kMine];
#pragma unroll
    for (u32 k = 0; k < kMine; ++k) {
        const u32 s = wave + k * NWV;
        b0[k] = b1[k] = 0;
        if (s < spans) {
            b0[k] = tb[s * W];
            if (PAIRED) b1[k] = tb[s * W + 1];
        }
    }
#pragma unroll
    for (u32 k = 0; k < kMine; ++k) {
        const u32 s = wave + k * NWV;
        if (PAIRED) {
            if (((b0[k] | b1[k]) >> lane) & 1ull) val[k] = load_vec<u64, 2>(commit + ((u64)tile * spans + s) * 128 + 2 * lane);
        } else {
            if ((b0[k] >> lane) & 1ull) val[k].v[0] = commit[((u64)tile * spans + s) * 64 + lane];
        }
    }
    if (wave == 0) {
        const u32 sb = tile / kSuperTiles;
        u64 sum = 0;
        for (u32 i = lane; i < sb; i += 64) sum += super[i];
        for (u32 i = sb * kSuperTiles + lane; i < tile; i += 64) sum += counts[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const u32 lo = (u32)__shfl_xor((int)(u32)sum, d, 64), hi = (u32)__shfl_xor((int)(u32)(sum >> 32), d, 64);
            sum += ((u64)hi << 32) | lo;
        }
        if (lane == 0) {
            s_off = sum;
            if (last) *total = sum + n;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (u32 k = 0; k < kMine; ++k)
            if (wave + k * NWV < spans) s_cnt[wave + k * NWV] = __popcll(b0[k]) + __popcll(b1[k]);
    }
    if (n == 0) return;
    __syncthreads();
    const u64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (u32 k = 0; k < kMine; ++k) {
        const u32 s = wave + k * NWV;
        if (s >= spans) break;
        u64 pos = s_off + __popcll(b0[k] & lt) + __popcll(b1[k] & lt);
        for (u32 j = 0; j < s; ++j) pos += s_cnt[j];
        if (PAIRED) {
            const u64 g = ((u64)tile * spans + s) * 128 + 2 * lane;
            const bool c0 = (b0[k] >> lane) & 1ull, c1 = (b1[k] >> lane) & 1ull;
            if (c0) {
                if (pos < max_pairs) out[pos] = Pair{group_base + g, val[k].v[0]};
                ++pos;
            }
            if (c1 && pos < max_pairs) out[pos] = Pair{group_base + g + 1, val[k].v[1]};
        } else {
            const u64 g = ((u64)tile * spans + s) * 64 + lane;
            if (((b0[k] >> lane) & 1ull) && pos < max_pairs) out[pos] = Pair{group_base + g, val[k].v[0]};
        }
    }
}

template <bool PAIRED>
__global__ __launch_bounds__(kBlock) void k_pack_changed(const u64* __restrict__ bits,
                                                         const u32* __restrict__ counts,
                                                         const u32* __restrict__ super, u32 ntiles,
                                                         u32 spans, const u64* __restrict__ commit,
                                                         u64 group_base, Pair* __restrict__ out,
                                                         u64 max_pairs, u64* __restrict__ total) {
    pack_tile<PAIRED>(blockIdx.x, bits, counts, super, ntiles, spans, commit, group_base, out, max_pairs, total);
}

// The same for the shards of rafthip_commit_report_many (K1 tiles): one launch, every shard packed into its own buffer.
// The first min(*total, limit) packed pairs once more, into the engine's page-locked arena: a report of few pairs
// reaches a pageable destination without a second copy-and-wait (rafthip.hip commit_report).
__global__ __launch_bounds__(kBlock) void k_copy_head(const Pair* __restrict__ packed, const u64* __restrict__ total,
                                                      u32 limit, Pair* __restrict__ head) {
    const u64 t = *total;
    const u32 m = t < limit ? (u32)t : limit;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) head[i] = packed[i];
}

// Up to two small device buffers into the engine's page-locked arena, by one kernel instead of copy commands; a source
// marked `zero` is cleared behind the copy (the wire status counts: the next call needs no memset).
struct CopyOut {
    u32 n;
    u32 words[2];
    bool zero[2];
    u64* src[2];
    u64* dst[2];
};
__global__ __launch_bounds__(kBlock) void k_copy_out(const CopyOut c) {
    const u32 i = blockIdx.x * kBlock + threadIdx.x;
#pragma unroll
    for (u32 k = 0; k < 2; ++k) {
        if (k < c.n && i < c.words[k]) {
            c.dst[k][i] = c.src[k][i];
            if (c.zero[k]) c.src[k][i] = 0;
        }
    }
}

struct ManyPack {
    u32 n;
    u32 first_tile[kManyMax + 1];
    const u64* bits[kManyMax];
    const u32* counts[kManyMax];
    const u32* super[kManyMax];
    const u64* commit[kManyMax];
    Pair* out[kManyMax];
    u64 max_pairs[kManyMax];
    u64* total[kManyMax];
};

__global__ __launch_bounds__(kBlock) void k_pack_changed_many(const ManyPack sh) {
    u32 k = 0;
    for (u32 i = 1; i < sh.n; ++i) k = blockIdx.x >= sh.first_tile[i] ? i : k;
    pack_tile<true>(blockIdx.x - sh.first_tile[k], sh.bits[k], sh.counts[k], sh.super[k], sh.first_tile[k + 1] - sh.first_tile[k],
                    kTileGroups / 128, sh.commit[k], 0, sh.out[k], sh.max_pairs[k], sh.total[k]);
}

// ---------------------------------------------------------------------------
// C1, after the all-gather: every rank's pairs sit in its own `width`-pair slot of
// recv ([rank][width], only the first counts[rank] are meaningful); pack them into
// one contiguous run in rank order — ranks own ascending group-id ranges, so the
// run is sorted by global group id.  grid = (ceil(width / kBlock), nranks).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_pack_gathered(const Pair* __restrict__ recv,
                                                          const u64* __restrict__ counts, u32 nranks,
                                                          u64 width, Pair* __restrict__ out, u64 max_pairs) {
    const u32 r = blockIdx.y;
    const u64 i = (u64)blockIdx.x * kBlock + threadIdx.x;
    if (r >= nranks) return;
    const u64 mine = counts[r] < width ? counts[r] : width;
    if (i >= mine) return;
    u64 off = 0;
    for (u32 k = 0; k < r; ++k) off += counts[k] < width ? counts[k] : width;
    if (off + i < max_pairs) out[off + i] = recv[(u64)r * width + i];
}

// ---------------------------------------------------------------------------
// K5 — Progress.MaybeUpdate (Match part) for a stream of acks:
// Match = max(Match, index), tracker/progress.go:170-173.  64-bit atomic max is
// order-independent, so duplicates and stale acks inside a batch are exact.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_scatter_acks(u64* __restrict__ match, u64 stride, u64 G,
                                                         u32 S, const Ack* __restrict__ recs, u64 n,
                                                         u64* __restrict__ rejected) {
    const u64 i = (u64)blockIdx.x * kBlock + threadIdx.x;
    bool bad = false;
    if (i < n) {
        const Vec<u64, 2> raw = load_vec<u64, 2>(reinterpret_cast<const u64*>(recs + i));
        const u32 group = (u32)raw.v[0], slot = (u32)(raw.v[0] >> 32);
        const u64 index = raw.v[1];
        if ((u64)group < G && slot < S) {
            atomicMax(match + (u64)slot * stride + group, index);
        } else {
            bad = true;
        }
    }
    const u64 b = __ballot(bad);
    if (b != 0 && (threadIdx.x & 63) == 0) atomicAdd(rejected, (u64)__popcll(b));
}

// ---------------------------------------------------------------------------
// K5d — the same MaybeUpdate for a tick whose acks arrive as a dense plane:
// ack[slot][group] = the index that peer acknowledged this tick, 0 = nothing
// (max(Match, 0) = Match, and an acknowledged index 0 never moves Match either,
// tracker/progress.go:170).  Elementwise max over [slot][group] planes at
// streaming rate, 24 B per cell: beats the scattered atomics of K5 (about 40 ps
// per ack) as soon as more than one cell in ten receives an ack.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_merge_acks(u64* __restrict__ match, u64 stride,
                                                       const u64* __restrict__ acks, u64 ack_stride,
                                                       u64 first, u64 count, u32 S) {
    const u64 g0 = ((u64)blockIdx.x * kBlock + threadIdx.x) * 2;  // pair of groups within [0, count)
    if (g0 >= count) return;
    const bool pair = g0 + 1 < count && ((first | ack_stride) & 1) == 0;  // both 16-byte aligned
    if (pair) {
        constexpr u32 U = 4;  // slots per step: 8 loads of 16 B in flight per lane
        for (u32 s0 = 0; s0 < S; s0 += U) {
            Vec<u64, 2> mv[U], av[U];
#pragma unroll
            for (u32 k = 0; k < U; ++k) {
                const u32 s = s0 + k < S ? s0 + k : S - 1;  // the tail re-reads the last slot (same result)
                mv[k] = load_stream<u64, 2>(match + (u64)s * stride + first + g0);
                av[k] = load_stream<u64, 2>(acks + (u64)s * ack_stride + g0);
            }
#pragma unroll
            for (u32 k = 0; k < U; ++k) {
                const bool c0 = av[k].v[0] > mv[k].v[0], c1 = av[k].v[1] > mv[k].v[1];
                if (__any(c0 | c1) && s0 + k < S) {  // whole-wave stores: full lines, no masked partial writes
                    mv[k].v[0] = c0 ? av[k].v[0] : mv[k].v[0];
                    mv[k].v[1] = c1 ? av[k].v[1] : mv[k].v[1];
                    store_stream<u64, 2>(match + (u64)(s0 + k) * stride + first + g0, mv[k]);
                }
            }
        }
    } else {
        for (u32 s = 0; s < S; ++s) {
            u64* m = match + (u64)s * stride + first + g0;
            const u64* a = acks + (u64)s * ack_stride + g0;
            if (a[0] > m[0]) m[0] = a[0];
            if (g0 + 1 < count && a[1] > m[1]) m[1] = a[1];
        }
    }
}

// ---------------------------------------------------------------------------
// K5v — ProgressTracker.RecordVote for a stream of vote responses:
// "the first vote of an id sticks" (tracker/tracker.go:260-265).  First means
// first in stream order, and two records of one batch may disagree (a duplicate
// from a confused peer), so the order must not depend on which lane runs first:
//   pass 1  every record raises first[cell] to ~i with a 32-bit atomicMax — the
//           largest ~i is the smallest stream index i, whatever the order
//   pass 2  the record that finds its own ~i there writes its vote, if the cell
//           held none before the batch (one writer per cell: plain byte store)
//   pass 3  every record clears first[cell] again
// ---------------------------------------------------------------------------
constexpr uint8_t kMsgVoteResp = 6, kMsgPreVoteResp = 18;  // raftpb/raft.proto:48,60
__device__ __forceinline__ bool vote_record(const Msg* msgs, u64 i, u64 G, u32 S, u32& group, u32& slot, bool& reject) {
    const Vec<u64, 2> raw = load_vec<u64, 2>(reinterpret_cast<const u64*>(msgs + i));
    group = (u32)raw.v[0];
    slot = (u32)(raw.v[0] >> 32) & 0xFFu;
    const u32 type = (u32)(raw.v[0] >> 40) & 0xFFu;
    reject = ((raw.v[0] >> 48) & 0xFFu) != 0;
    return (type == kMsgVoteResp || type == kMsgPreVoteResp) && (u64)group < G && slot < S;
}
template <int PASS>
__global__ __launch_bounds__(kBlock) void k_record_votes(uint8_t* __restrict__ votes, u32* __restrict__ first,
                                                         u64 stride, u64 G, u32 S, const Msg* __restrict__ msgs,
                                                         u32 n, u64* __restrict__ dropped) {
    const u32 i = blockIdx.x * kBlock + threadIdx.x;
    bool bad = false;
    if (i < n) {
        u32 group, slot;
        bool reject;
        if (vote_record(msgs, i, G, S, group, slot, reject)) {
            const u64 cell = (u64)slot * stride + group;
            if (PASS == 1) {
                atomicMax(first + cell, ~i);
            } else if (PASS == 2) {
                if (first[cell] == ~i && votes[cell] == 0) votes[cell] = reject ? 1 : 2;
            } else {
                first[cell] = 0;
            }
        } else {
            bad = PASS == 1;
        }
    }
    if (PASS == 1) {
        const u64 b = __ballot(bad);
        if (dropped != nullptr && b != 0 && (threadIdx.x & 63) == 0) atomicAdd(dropped, (u64)__popcll(b));
    }
}

// ---------------------------------------------------------------------------
// K7 — the leader's per-message bookkeeping for a stream of responses that the
// host has resolved to (group, slot): stepLeader, raft.go:1327-1561.
//   MsgAppResp        RecentActive = true (:1337); if !Reject: MaybeUpdate(Index) (:1477)
//   MsgHeartbeatResp  RecentActive = true (:1528); with the pending ReadIndex
//                     context: readOnly.recvAck, acks[from] = true (read_only.go:68-76)
//   kMsgAppended      local: raise the group's term_last to the leader's new last index
// Every update is a max or an or — commutative and idempotent — so the batch
// equals stepping the messages one by one in any order.  Records with no
// Progress behind them (group >= G, slot >= S) are dropped and counted
// (raft.go:1327-1331); other message types are counted as ignored.
// ---------------------------------------------------------------------------
// The bit is read first: RecentActive / ack bits are mostly set already in
// steady state (they are cleared once per election timeout), and a set bit
// needs no atomic.  A stale read can only cause a redundant or.
__device__ __forceinline__ void or_slot_bit(void* plane, int mbytes, u64 g, u32 slot) {
    if (mbytes == 8) {
        u64* w = reinterpret_cast<u64*>(plane) + g;
        const u64 bit = 1ull << slot;
        if (!(*w & bit)) atomicOr(w, bit);
    } else {
        const u64 byte = g * (u64)mbytes;  // little-endian: bit `slot` of the mask at this byte offset
        u32* w = reinterpret_cast<u32*>(plane) + (byte >> 2);
        const u32 bit = 1u << (((u32)byte & 3u) * 8u + slot);
        if (!(*w & bit)) atomicOr(w, bit);
    }
}

__global__ __launch_bounds__(kBlock) void k_ingest_msgs(u64* __restrict__ match, u64 stride, u64 G, u32 S,
                                                        void* __restrict__ active, void* __restrict__ acks,
                                                        int mbytes, u64* __restrict__ term_last,
                                                        const Msg* __restrict__ msgs, u64 n,
                                                        u64* __restrict__ dropped /* [0] no Progress, [1] ignored type */) {
    const u64 i = (u64)blockIdx.x * kBlock + threadIdx.x;
    bool bad = false, ign = false;
    if (i < n) {
        const Vec<u64, 2> raw = load_vec<u64, 2>(reinterpret_cast<const u64*>(msgs + i));
        const u32 group = (u32)raw.v[0];
        const u32 slot = (u32)(raw.v[0] >> 32) & 0xFFu, type = (u32)(raw.v[0] >> 40) & 0xFFu;
        const bool reject = ((raw.v[0] >> 48) & 0xFFu) != 0;
        const u32 flags = (u32)(raw.v[0] >> 56);
        const u64 index = raw.v[1];
        if (type == kMsgAppended) {
            if ((u64)group < G) {
                if (term_last) atomicMax(term_last + group, index);
            } else {
                bad = true;
            }
        } else if (type != kMsgAppResp && type != kMsgHeartbeatResp) {
            ign = true;
        } else if ((u64)group >= G || slot >= S) {
            bad = true;
        } else {
            or_slot_bit(active, mbytes, group, slot);
            if (type == kMsgAppResp) {
                if (!reject) atomicMax(match + (u64)slot * stride + group, index);
            } else if (flags & kMsgFlagCtx) {
                or_slot_bit(acks, mbytes, group, slot);
            }
        }
    }
    const u64 b = __ballot(bad), c = __ballot(ign);
    if ((threadIdx.x & 63) == 0) {
        if (b) atomicAdd(dropped, (u64)__popcll(b));
        if (c) atomicAdd(dropped + 1, (u64)__popcll(c));
    }
}

// ---------------------------------------------------------------------------
// K6 — the commit index a heartbeat may carry to each peer:
// min(Progress.Match, raftLog.committed), sendHeartbeat raft.go:669-685.
// Elementwise over the [slot][group] planes, 16 B per lane per access.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_heartbeat_commit(const u64* __restrict__ match, u64 stride,
                                                             const u64* __restrict__ commit,
                                                             u64* __restrict__ out, u32 S) {
    const u64 g0 = ((u64)blockIdx.x * kBlock + threadIdx.x) * 2;  // < stride (multiple of 4096)
    const Vec<u64, 2> c = load_stream<u64, 2>(commit + g0);
    for (u32 s = 0; s < S; ++s) {
        Vec<u64, 2> m = load_stream<u64, 2>(match + (u64)s * stride + g0);
        m.v[0] = m.v[0] < c.v[0] ? m.v[0] : c.v[0];
        m.v[1] = m.v[1] < c.v[1] ? m.v[1] : c.v[1];
        store_stream<u64, 2>(out + (u64)s * stride + g0, m);
    }
}

// ---------------------------------------------------------------------------
// raft.committedEntryInCurrentTerm for every group — raft.go:2001-2005:
// zeroTermOnOutOfBounds(term(raftLog.committed)) == r.Term, i.e. the committed
// index lies in the run of entries that carry the leader's term (see kAdvance).
// Gates ReadIndex (raft.go:1312-1317) and releasePendingReadIndexMessages.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_committed_in_term(const u64* __restrict__ commit,
                                                              const u64* __restrict__ term_first,
                                                              const u64* __restrict__ term_last,
                                                              uint8_t* __restrict__ out) {
    const u64 g0 = ((u64)blockIdx.x * kBlock + threadIdx.x) * 2;  // < stride (multiple of 4096)
    const Vec<u64, 2> c = load_stream<u64, 2>(commit + g0), a = load_stream<u64, 2>(term_first + g0),
                      b = load_stream<u64, 2>(term_last + g0);
    Vec<uint8_t, 2> r;
#pragma unroll
    for (int j = 0; j < 2; ++j) r.v[j] = (c.v[j] >= a.v[j] && c.v[j] <= b.v[j]) ? 1 : 0;
    store_vec<uint8_t, 2>(out + g0, r);
}

// ---------------------------------------------------------------------------
// K3 — tallies, S <= 8 (one mask byte per group).  A lane handles 16 groups:
// one 16-byte load per plane, arithmetic done four groups at a time on the
// bytes of a dword (counts never exceed 8, so bytes cannot carry).
// ---------------------------------------------------------------------------
constexpr u32 k01 = 0x01010101u, k80 = 0x80808080u;

__device__ __forceinline__ u32 swar_popc8(u32 x) {  // per-byte popcount
    x = x - ((x >> 1) & 0x55555555u);
    x = (x & 0x33333333u) + ((x >> 2) & 0x33333333u);
    return (x + (x >> 4)) & 0x0F0F0F0Fu;
}
// bit 7 of each byte set iff a >= b (bytes < 128)
__device__ __forceinline__ u32 swar_ge(u32 a, u32 b) { return ((a | k80) - b) & k80; }
// bit 7 of each byte set iff the byte is zero (bytes < 128)
__device__ __forceinline__ u32 swar_is0(u32 a) { return ~((a | k80) - k01) & k80; }

// MajorityConfig.VoteResult on 4 groups: n voters, yes and no counts per byte.
// Returns (won, lost) as bit-7 flags.  quorum/majority.go:178-207.
__device__ __forceinline__ void swar_vote_half(u32 n, u32 yes, u32 no, u32& won, u32& lost) {
    const u32 q = ((n >> 1) & 0x7F7F7F7Fu) + k01;  // n/2+1
    const u32 empty = swar_is0(n);                 // n == 0 -> VoteWon (:179-184)
    won = swar_ge(yes, q) | empty;
    const u32 pend_or_won = swar_ge(n - no, q) | empty;  // yes + missing = n - no
    lost = ~pend_or_won & k80;
}
// JointConfig.VoteResult, quorum/joint.go:61-75: Won iff both won; Lost if
// either lost; else Pending.  Encoded 1 + 2*won + lost per byte.
__device__ __forceinline__ u32 swar_joint(u32 won0, u32 lost0, u32 won1, u32 lost1) {
    const u32 w = ((won0 & won1) >> 7) & k01;
    const u32 l = ((lost0 | lost1) >> 7) & k01;
    return k01 + (w << 1) + l;
}

template <int S>
__global__ __launch_bounds__(kBlock) void k_tally_votes_small(
    const uint8_t* __restrict__ votes, u64 stride, const uint8_t* __restrict__ in_mask,
    const uint8_t* __restrict__ out_mask, const uint8_t* __restrict__ learner_mask, u32 uniform_voters,
    uint8_t* __restrict__ result, uint8_t* __restrict__ granted, uint8_t* __restrict__ rejected) {
    const u64 g0 = ((u64)blockIdx.x * kBlock + threadIdx.x) * 16;
    Vec<u32, 4> cell[S];
#pragma unroll
    for (int s = 0; s < S; ++s) cell[s] = load_stream<u32, 4>(reinterpret_cast<const u32*>(votes + (u64)s * stride + g0));
    Vec<u32, 4> M0, M1, ML;
    if (in_mask) {
        M0 = load_stream<u32, 4>(reinterpret_cast<const u32*>(in_mask + g0));
        M1 = load_stream<u32, 4>(reinterpret_cast<const u32*>(out_mask + g0));
        if (learner_mask) ML = load_stream<u32, 4>(reinterpret_cast<const u32*>(learner_mask + g0));
    }
    const u32 slots = (1u << S) - 1u;
    Vec<u32, 4> res, gr, rj;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        u32 m0, m1, ml;
        if (in_mask) {
            m0 = M0.v[w] & (slots * k01);
            m1 = M1.v[w] & (slots * k01);
            ml = learner_mask ? ML.v[w] : 0u;
        } else {
            const u32 vm = (uniform_voters >= 8 ? 0xFFu : ((1u << uniform_voters) - 1u)) & slots;
            m0 = vm * k01;
            m1 = 0;
            ml = (slots & ~vm) * k01;
        }
        u32 yes0 = 0, no0 = 0, yes1 = 0, no1 = 0, g = 0, r = 0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const u32 c = cell[s].v[w];
            const u32 y = (c >> 1) & k01, n = c & ~(c >> 1) & k01;  // cell 2 = granted, 1 = rejected
            const u32 in0 = (m0 >> s) & k01, in1 = (m1 >> s) & k01, nl = ~(ml >> s) & k01;
            yes0 += y & in0;
            no0 += n & in0;
            yes1 += y & in1;
            no1 += n & in1;
            g += y & nl;  // tracker/tracker.go:274-287
            r += n & nl;
        }
        u32 w0, l0, w1, l1;
        swar_vote_half(swar_popc8(m0), yes0, no0, w0, l0);
        swar_vote_half(swar_popc8(m1), yes1, no1, w1, l1);
        res.v[w] = swar_joint(w0, l0, w1, l1);
        gr.v[w] = g;
        rj.v[w] = r;
    }
    store_stream<u32, 4>(reinterpret_cast<u32*>(result + g0), res);
    if (granted) store_stream<u32, 4>(reinterpret_cast<u32*>(granted + g0), gr);
    if (rejected) store_stream<u32, 4>(reinterpret_cast<u32*>(rejected + g0), rj);
}

// QuorumActive (ACKS=false: output bool, tracker/tracker.go:217-227) and the
// ReadIndex ack quorum (ACKS=true: output VoteResult, raft.go:1552 +
// read_only.go:68-76) from one yes-bitmask per group.
//  - QuorumActive: every non-learner row votes its RecentActive bit, so a voter
//    is never "missing": a half is Won iff yes >= q, else Lost.
//  - Acks: the map only ever holds true, so a half is Won or Pending.
//  - keep != NULL (MsgCheckQuorum, raft.go:1238-1242): after the tally every
//    RecentActive bit except the leader's own (keep = self-slot mask) is cleared
//    "in preparation for the next CheckQuorum".
template <bool ACKS>
__global__ __launch_bounds__(kBlock) void k_tally_bits_small(
    uint8_t* __restrict__ bits, const uint8_t* __restrict__ in_mask,
    const uint8_t* __restrict__ out_mask, const uint8_t* __restrict__ learner_mask, u32 S,
    u32 uniform_voters, uint8_t* __restrict__ out, const uint8_t* __restrict__ keep) {
    const u64 g0 = ((u64)blockIdx.x * kBlock + threadIdx.x) * 16;
    const Vec<u32, 4> B = load_stream<u32, 4>(reinterpret_cast<const u32*>(bits + g0));
    Vec<u32, 4> M0, M1, ML;
    if (in_mask) {
        M0 = load_stream<u32, 4>(reinterpret_cast<const u32*>(in_mask + g0));
        M1 = load_stream<u32, 4>(reinterpret_cast<const u32*>(out_mask + g0));
        if (!ACKS && learner_mask) ML = load_stream<u32, 4>(reinterpret_cast<const u32*>(learner_mask + g0));
    }
    const u32 slots = (S >= 8 ? 0xFFu : ((1u << S) - 1u));
    Vec<u32, 4> res;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        u32 m0, m1, yes = B.v[w];
        if (in_mask) {
            m0 = M0.v[w] & (slots * k01);
            m1 = M1.v[w] & (slots * k01);
            if (!ACKS && learner_mask) yes &= ~ML.v[w];  // learners cast no vote (:220-222)
        } else {
            m0 = ((uniform_voters >= 8 ? 0xFFu : ((1u << uniform_voters) - 1u)) & slots) * k01;
            m1 = 0;
        }
        const u32 n0 = swar_popc8(m0), n1 = swar_popc8(m1);
        const u32 y0 = swar_popc8(yes & m0), y1 = swar_popc8(yes & m1);
        u32 w0, l0, w1, l1;
        // ACKS: no = 0 (never Lost).  QuorumActive: no = n - yes (never Pending).
        swar_vote_half(n0, y0, ACKS ? 0u : n0 - y0, w0, l0);
        swar_vote_half(n1, y1, ACKS ? 0u : n1 - y1, w1, l1);
        if (ACKS) {
            res.v[w] = swar_joint(w0, l0, w1, l1);
        } else {
            res.v[w] = ((w0 & w1) >> 7) & k01;  // == VoteWon
        }
    }
    store_stream<u32, 4>(reinterpret_cast<u32*>(out + g0), res);
    if (keep) {
        const Vec<u32, 4> K = load_stream<u32, 4>(reinterpret_cast<const u32*>(keep + g0));
        Vec<u32, 4> nb;
#pragma unroll
        for (int w = 0; w < 4; ++w) nb.v[w] = B.v[w] & K.v[w];
        store_vec<u32, 4>(reinterpret_cast<u32*>(bits + g0), nb);
    }
}

// ---------------------------------------------------------------------------
// K3, wide slots (9..64): one lane per group, popcount on the 64-bit masks.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int vote_half(u64 m, u64 yes, u64 no) {
    const int n = __popcll(m);
    if (n == 0) return 3;
    const int q = n / 2 + 1;
    const int y = __popcll(yes & m), x = __popcll(no & m);
    if (y >= q) return 3;
    if (n - x >= q) return 1;
    return 2;
}
__device__ __forceinline__ int vote_joint(int r1, int r2) {
    if (r1 == r2) return r1;
    if (r1 == 2 || r2 == 2) return 2;
    return 1;
}

// A lane handles 4*NW consecutive groups: one NW-dword vector per vote plane,
// masks as 4-element vectors, results as one NW-dword vector per output plane.
// Eight slots at a time: bit b of byte j of ay[q] = "slot 8w+b granted in group
// 4q+j", so a slot costs two shift-ors per four groups; the bytes are spread into
// the per-group 64-bit sets once per eight slots.
constexpr int kVoteWideNW = 4;  // 16 groups per lane: 16-B loads per vote plane
template <typename MaskT>
__global__ __launch_bounds__(kBlock) void k_tally_votes_wide(
    const uint8_t* __restrict__ votes, u64 stride, int S, const MaskT* __restrict__ in_mask,
    const MaskT* __restrict__ out_mask, const MaskT* __restrict__ learner_mask, int uniform_voters,
    uint8_t* __restrict__ result, uint8_t* __restrict__ granted, uint8_t* __restrict__ rejected) {
    constexpr int NW = kVoteWideNW;
    const u64 g0 = ((u64)blockIdx.x * kBlock + threadIdx.x) * (4 * NW);  // < stride (multiple of 4096)
    const u64 slots = S >= 64 ? kU64Max : ((1ull << S) - 1ull);
    u64 yes[NW][4], no[NW][4];
#pragma unroll
    for (int q = 0; q < NW; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) yes[q][j] = no[q][j] = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        if (w * 8 >= S) break;  // wave-uniform
        u32 ay[NW], an[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) ay[q] = an[q] = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int s = w * 8 + b;
            if (s < S) {
                const Vec<u32, NW> c = load_stream<u32, NW>(reinterpret_cast<const u32*>(votes + (u64)s * stride + g0));
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    ay[q] |= ((c.v[q] >> 1) & k01) << b;               // cell 2 = granted
                    an[q] |= (c.v[q] & ~(c.v[q] >> 1) & k01) << b;      // cell 1 = rejected
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NW; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                yes[q][j] |= (u64)((ay[q] >> (8 * j)) & 0xFFu) << (8 * w);
                no[q][j] |= (u64)((an[q] >> (8 * j)) & 0xFFu) << (8 * w);
            }
    }
    Vec<u32, NW> res, gr, rj;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        u64 m0[4], m1[4], ml[4];
        if (in_mask) {
            const Vec<MaskT, 4> a = load_vec<MaskT, 4>(in_mask + g0 + 4 * q), b = load_vec<MaskT, 4>(out_mask + g0 + 4 * q);
            Vec<MaskT, 4> c;
            if (learner_mask) c = load_vec<MaskT, 4>(learner_mask + g0 + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m0[j] = (u64)a.v[j] & slots;
                m1[j] = (u64)b.v[j] & slots;
                ml[j] = learner_mask ? (u64)c.v[j] : 0ull;
            }
        } else {
            const u64 vm = (uniform_voters >= 64 ? kU64Max : ((1ull << uniform_voters) - 1ull)) & slots;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m0[j] = vm;
                m1[j] = 0;
                ml[j] = slots & ~vm;
            }
        }
        u32 r = 0, g = 0, x = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r |= (u32)vote_joint(vote_half(m0[j], yes[q][j], no[q][j]), vote_half(m1[j], yes[q][j], no[q][j])) << (8 * j);
            g |= (u32)__popcll(yes[q][j] & ~ml[j]) << (8 * j);  // tracker/tracker.go:274-287
            x |= (u32)__popcll(no[q][j] & ~ml[j]) << (8 * j);
        }
        res.v[q] = r;
        gr.v[q] = g;
        rj.v[q] = x;
    }
    store_vec<u32, NW>(reinterpret_cast<u32*>(result + g0), res);
    if (granted) store_vec<u32, NW>(reinterpret_cast<u32*>(granted + g0), gr);
    if (rejected) store_vec<u32, NW>(reinterpret_cast<u32*>(rejected + g0), rj);
}

template <typename MaskT, bool ACKS>
__global__ __launch_bounds__(kBlock) void k_tally_bits_wide(
    MaskT* __restrict__ bits, int S, const MaskT* __restrict__ in_mask,
    const MaskT* __restrict__ out_mask, const MaskT* __restrict__ learner_mask, int uniform_voters,
    uint8_t* __restrict__ out, const MaskT* __restrict__ keep) {
    const u64 g0 = ((u64)blockIdx.x * kBlock + threadIdx.x) * 4;  // 4 groups per lane; < stride
    const u64 slots = S >= 64 ? kU64Max : ((1ull << S) - 1ull);
    const Vec<MaskT, 4> B = load_vec<MaskT, 4>(bits + g0);
    Vec<MaskT, 4> M0, M1, ML, K;
    const bool lrn = !ACKS && in_mask && learner_mask;
    if (in_mask) {
        M0 = load_vec<MaskT, 4>(in_mask + g0);
        M1 = load_vec<MaskT, 4>(out_mask + g0);
        if (lrn) ML = load_vec<MaskT, 4>(learner_mask + g0);
    }
    if (keep) K = load_vec<MaskT, 4>(keep + g0);
    u32 res = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u64 m0, m1, yes = (u64)B.v[j];
        if (in_mask) {
            m0 = (u64)M0.v[j] & slots;
            m1 = (u64)M1.v[j] & slots;
            if (lrn) yes &= ~(u64)ML.v[j];  // learners cast no vote (tracker.go:220-222)
        } else {
            m0 = (uniform_voters >= 64 ? kU64Max : ((1ull << uniform_voters) - 1ull)) & slots;
            m1 = 0;
        }
        const u64 no = ACKS ? 0ull : ~yes;
        const int r = vote_joint(vote_half(m0, yes, no), vote_half(m1, yes, no));
        res |= (u32)(ACKS ? r : (r == 3)) << (8 * j);
    }
    *reinterpret_cast<u32*>(out + g0) = res;
    if (keep) {
        Vec<MaskT, 4> nb;
#pragma unroll
        for (int j = 0; j < 4; ++j) nb.v[j] = B.v[j] & K.v[j];
        store_vec<MaskT, 4>(bits + g0, nb);
    }
}

}  // namespace rafthip
