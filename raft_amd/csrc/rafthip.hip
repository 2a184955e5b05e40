// rafthip.hip — host side of librafthip.so: the engine object, kernel dispatch
// and the C ABI declared in include/rafthip.h.  No torch, no CPU compute path:
// every result comes from the gfx950 kernels in kernels.h.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <emmintrin.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <initializer_list>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "kernels.h"
#include "progress.h"
#include "scatter_part.h"
#include "progress_part.h"
#include "wire.h"
#include "tick.h"
#include "rafthip.h"
#include "rafthip_internal.h"

using namespace rafthip;

static_assert(sizeof(Pair) == sizeof(rafthip_commit_pair), "pair layout");
static_assert(sizeof(Ack) == sizeof(rafthip_ack), "ack layout");
static_assert(sizeof(Msg) == sizeof(rafthip_msg) && sizeof(Msg) == 16, "msg layout");
static_assert(kWireApplied == RAFTHIP_WIRE_APPLIED && kWireOtherType == RAFTHIP_WIRE_OTHER_TYPE &&
                  kWireNoProgress == RAFTHIP_WIRE_NO_PROGRESS && kWireStaleTerm == RAFTHIP_WIRE_STALE_TERM &&
                  kWireHigherTerm == RAFTHIP_WIRE_HIGHER_TERM && kWireNested == RAFTHIP_WIRE_NESTED &&
                  kWireMalformed == RAFTHIP_WIRE_MALFORMED && kWireBadGroup == RAFTHIP_WIRE_BAD_GROUP &&
                  kWireNeedsLog == RAFTHIP_WIRE_NEEDS_LOG && kWireVote == RAFTHIP_WIRE_VOTE && kEvNeedsHost == RAFTHIP_MSG_NEEDS_HOST &&
                  kWireStatuses == RAFTHIP_WIRE_STATUSES && kMsgFlagHasCtx == RAFTHIP_MSGF_HAS_CTX &&
                  kWireNoSlot == RAFTHIP_WIRE_NO_SLOT,
              "wire constants");
static_assert(kMsgAppResp == RAFTHIP_MSG_APP_RESP && kMsgHeartbeatResp == RAFTHIP_MSG_HEARTBEAT_RESP &&
                  kMsgAppended == RAFTHIP_MSG_APPENDED && kMsgFlagCtx == RAFTHIP_MSGF_READ_CTX,
              "msg constants");

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(_e == hipErrorOutOfMemory ? RAFTHIP_ENOMEM : RAFTHIP_EIO, "%s: %s",   \
                        #expr, hipGetErrorString(_e));                                        \
    } while (0)

u64 round_up(u64 x, u64 a) { return (x + a - 1) / a * a; }

constexpr u64 kMaxRecordsPerLaunch = 1ull << 30;  // scatter / ingest records per kernel launch
constexpr size_t kSmallArenaBytes = 16u << 20;    // rafthip_engine::h_small
constexpr u64 kSmallPairs = 4096;                 // (group, commit) pairs of a report that come back through it
constexpr u64 kSmallBatch = 131072;               // messages per call up to which the wire entry points use it
constexpr u64 kSmallRecords = 16384;              // 16-byte records per call up to which the record entry points do: their
                                                  // kernels read a record more than once, over the link when it lies in the
                                                  // arena (break-even near 30000 records, tools/small_tick_probe.py)

int mask_bytes_for(u32 S) { return S <= 8 ? 1 : S <= 16 ? 2 : S <= 32 ? 4 : 8; }

}  // namespace

// ---- scratch of the allocation-free entry point (rafthip_commit_changed_static) ----
// One copy per GPU the library runs on, held until the process ends.  Layout:
// ballot words, per-tile counts, super counts + total, for planes of up to
// kStaticMaxGroups groups.  Host code never passes the array's name to a kernel
// (that name is a host-side stand-in, not a device address): static_scratch()
// resolves the address once per device and every use goes through it.
constexpr u64 kStaticMaxGroups = 1ull << 24;
constexpr size_t kStaticBitsBytes = kStaticMaxGroups / 8;
constexpr size_t kStaticCountsBytes = kStaticMaxGroups / kBlock * sizeof(u32);
constexpr size_t kStaticSuperBytes = (kStaticMaxGroups / kBlock + kSuperTiles - 1) / kSuperTiles * sizeof(u32) + 8;
constexpr size_t kStaticScratchBytes = kStaticBitsBytes + kStaticCountsBytes + ((kStaticSuperBytes + 255) & ~(size_t)255);
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};
// One holder of a device's scratch array at a time, from the first kernel that writes it to the wait
// that ends the call (rafthip_commit_changed_static): an overlapping call on the same GPU fails with
// RAFTHIP_EBUSY instead of returning pairs built from another call's ballot words.
std::mutex g_scratch_busy[kMaxDevices];

// Address of this device's copy of `scratch` (device must be current).  Resolved
// on the first rafthip_create on a device; later calls only read the table.
int static_scratch(int device, void** out) {
    if (device < 0 || device >= kMaxDevices) return fail(RAFTHIP_EINVAL, "device %d outside the scratch table", device);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return RAFTHIP_OK;
}
}  // namespace

// Roles of the engine's staging buffers (an entry point uses each role for one thing at a time).
enum StageSlot : int {
    kStageBytes = 0,  // serialized message bytes in / heartbeat bytes out
    kStageOffsets,    // message offsets / heartbeat-response masks of a dense tick
    kStageGroups,     // envelope group ids / ReadIndex-context masks of a dense tick
    kStageMsgs,       // decoded records
    kStageStatus,     // per-message status / heartbeat lengths
    kStageAux,        // aux words of events / a dense ack plane
    kStageActions,    // action bytes
    kStagePrev,       // per-event list links of the Progress stepper
    kStageSlots
};

struct rafthip_engine {
    int device = 0;
    u64 G = 0;
    u32 S = 0;
    u64 stride = 0;  // elements per plane row (multiple of kPlaneAlign)
    int mbytes = 1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool wide_bitonic = false;  // RAFTHIP_FLAG_WIDE_BITONIC

    // config
    bool uniform = true;
    u32 uniform_voters = 0;
    void* in_mask = nullptr;
    void* out_mask = nullptr;
    void* learner_mask = nullptr;
    bool have_learners = false;

    // state planes
    u64* match = nullptr;
    u64* commit = nullptr;
    uint8_t* votes = nullptr;
    u32* vote_first = nullptr;  // [S][stride] scratch of rafthip_record_votes, zero between calls
    void* active = nullptr;
    void* acks = nullptr;
    void* self_mask = nullptr;
    u64* term_first = nullptr;  // raftLog entries carrying the leader's term: [term_first, term_last]
    u64* term_last = nullptr;
    u64* hb_commit = nullptr;   // [S][stride], min(Match, committed)

    // outputs
    uint8_t* vote_result = nullptr;
    uint8_t* granted = nullptr;
    uint8_t* rejected = nullptr;
    uint8_t* qactive = nullptr;
    uint8_t* ack_result = nullptr;
    uint8_t* in_term = nullptr;

    // changed-pair reporting (kernels.h: Report, k_pack_changed)
    u64* rep_bits = nullptr;    // ballot words, one bit per group of [0, stride)
    u32* rep_counts = nullptr;  // [stride / kBlock] per-tile counts (sized for the smallest tile)
    u32* rep_super = nullptr;   // two arrays of [ceil(tiles / kSuperTiles)] (used in turn) + the total as the u64 behind them
    int rep_flip = 0;           // which of the two the next reporting pass adds into
    Pair* packed = nullptr;     // landing buffer when the caller's dst is pageable host memory

    // scatter
    Ack* ack_buf = nullptr;  // staging for host-resident ack / msg records (both 16 B)
    u64 ack_cap = 0;
    int scatter_mode = 0;    // 0 auto, 1 always atomics (RAFTHIP_FLAG_SCATTER_ATOMIC), 2 partition whenever it can run
    int step_mode = 0;       // the Progress stepper: 0 auto, 1 always K9's global lists (RAFTHIP_FLAG_STEP_LINKED), 2 K9p whenever it can run
    bool prog_part_attr = false;  // K9p's kernels have had their LDS sizes raised on this device
    Ack* part_a = nullptr;   // K5p (scatter_part.h): records by bucket, then by tile
    Ack* part_b = nullptr;
    u64 part_cap = 0;
    u32* part_words = nullptr;  // tile counts / bases / cursors, bucket cursors, chunk table, flag
    bool part_attr = false;     // the kernels' LDS sizes have been raised on this device
    bool part_dirty = false;    // a partitioned call did not get all its passes onto the stream: its tile counts (zero
                                // between calls by contract) may be left behind; the next call clears the words first
    u64* d_rejected = nullptr;  // [0] acks without Progress, [1] msgs without Progress, [2] msgs of other types
    // Small batches from / to pageable host memory (the wire entry points): inputs are copied by the CPU into this
    // page-locked, device-mapped arena and read from there by the kernels, outputs are written into it by the kernels
    // and copied out by the CPU after the call's one wait — no DMA copies, each of which costs ~10 us whatever its
    // size.  Allocated by the engine's first small batch; bump-allocated per call; a call that uses it waits for the
    // stream before it returns.
    char* h_small = nullptr;
    char* d_small = nullptr;    // the same memory as the device sees it
    size_t small_used = 0;
    bool small_failed = false;  // the arena could not be allocated: never tried again
    u64* h_scalar = nullptr;    // pinned, 4 x u64
    u64* d_scalar = nullptr;    // the same words as kernels address them (a total stored there needs no copy back)

    // wire ingest
    u64* wire_hdr = nullptr;       // [G][H] per group: r.Term, pending ReadIndex ctx, its length, raft id per slot (wire.h)
    bool have_ids = false, have_term = false;
    bool campaigning = false;  // rafthip_upload_campaign named a vote response type for some group
    u64* d_wire_counts = nullptr;  // [RAFTHIP_WIRE_STATUSES]
    bool wire_counts_zero = false;  // the last wire call's copy-out kernel left them at zero: no memset in front of the next
    // growable device staging for host-side inputs and outputs, one buffer per role
    void* stage_buf[kStageSlots] = {};
    size_t stage_cap[kStageSlots] = {};

    // per-peer replication state (rafthip_progress_config)
    bool zero_copy_pending = false;  // a kernel on the stream reads the caller's page-locked memory: wait before returning

    bool prog = false;
    bool heads_dirty = false;  // a link pass ran without its apply pass: list heads may be left in the records
    ProgPlanes pp{};
    u64* first_index = nullptr; // raftLog.firstIndex() per group

    // RawNode.Tick for every group (tick.h)
    u32* tick_el = nullptr;         // electionElapsed
    u32* tick_hb = nullptr;         // heartbeatElapsed
    u32* tick_rt = nullptr;         // randomizedElectionTimeout
    uint8_t* tick_role = nullptr;   // StateType
    uint8_t* tick_flags = nullptr;  // RAFTHIP_TICKF_*
    uint8_t* tick_action = nullptr; // RAFTHIP_TICK_* of the last pass
    u64* d_tick_counts = nullptr;   // [kTickCountCopies][kTickCountStride]
    u64* tick_ev = nullptr;         // rafthip_read_tick_events: the total, then one u32 count / offset per tile of kEvTile groups
    u32 election_timeout = 0, heartbeat_timeout = 0;  // 0: rafthip_tick_config has not been called
    bool check_quorum = false;

    rafthip_stats_t stats{};
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int check_range(const rafthip_engine* e, u64 first, u64 count) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    if (first > e->G || count > e->G - first)
        return fail(RAFTHIP_ERANGE, "group range [%llu, +%llu) outside [0, %llu)", first, count, e->G);
    return RAFTHIP_OK;
}

template <typename T>
int ensure(rafthip_engine* e, T*& p, size_t bytes, bool zero = true) {
    if (p) return RAFTHIP_OK;
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, bytes));
    p = static_cast<T*>(q);  // owned by the engine from here on (freed by rafthip_destroy)
    if (zero) HIP_TRY(hipMemsetAsync(q, 0, bytes, e->stream));
    return RAFTHIP_OK;
}

int ensure_masks(rafthip_engine* e) {
    const size_t b = (size_t)e->stride * e->mbytes;
    int rc;
    if ((rc = ensure(e, e->in_mask, b))) return rc;
    if ((rc = ensure(e, e->out_mask, b))) return rc;
    if ((rc = ensure(e, e->learner_mask, b))) return rc;
    return RAFTHIP_OK;
}

// 2-D copy of `rows` plane segments of `count` elements each.
int copy_rows(rafthip_engine* e, void* dst, u64 dst_stride_b, const void* src, u64 src_stride_b,
              u64 width_b, u64 rows) {
    if (width_b == 0 || rows == 0) return RAFTHIP_OK;
    HIP_TRY(hipMemcpy2DAsync(dst, dst_stride_b, src, src_stride_b, width_b, rows, hipMemcpyDefault,
                             e->stream));
    return RAFTHIP_OK;
}

// ---- kernel dispatch -----------------------------------------------------


// Lane-per-group sorting-network kernels K1 / K2 at <= 8 keys (tile =
// kTileGroups), K2 at 9..16 (tile = kMidTileGroups), the one-group-per-lane wide
// kernels above that (tile = kBlock).
bool small_commit_path(const rafthip_engine* e) {
    return e->S <= 8 || (e->uniform && e->uniform_voters > 0 && e->uniform_voters <= 8);
}
bool mid_commit_path(const rafthip_engine* e) { return !small_commit_path(e) && e->S <= 16 && !e->wide_bitonic; }

template <int MODE>
int launch_commit(rafthip_engine* e, u32* block_groups, Report rep = Report{nullptr, nullptr, nullptr, nullptr, 0},
                  u32 tiles_cap = 0) {
    constexpr bool CHANGED = MODE != kStore;  // needs the report buffers
    const u64 G = e->G;
    hipStream_t st = e->stream;
    const LogBounds lb{e->term_first, e->term_last};
    if (e->uniform && e->uniform_voters == 0) {
        // Both halves empty: MaxUint64 for every group (majority.go:128-132).
        // Expressed through the masked kernel with all-zero masks.
        int rc = ensure_masks(e);
        if (rc) return rc;
    }
    const bool uni = e->uniform && e->uniform_voters > 0;
    if (small_commit_path(e)) {
        // K1 reads only the voter planes, so a uniform config of <= 8 voters takes
        // it whatever S is (e.g. 7 voters + 2 learner rows).
        const u32 blocks = (u32)((G + kTileGroups - 1) / kTileGroups);
        *block_groups = kTileGroups;
        if (CHANGED && blocks > tiles_cap) return fail(RAFTHIP_EINVAL, "internal: report capacity");
        if (uni) {
            switch (e->uniform_voters) {
#define RAFTHIP_CASE_U(V)                                                                          \
    case V:                                                                                        \
        hipLaunchKernelGGL((k_commit_uniform<V, MODE>), dim3(blocks), dim3(kCommitBlock), 0, st,      \
                           e->match, e->stride, G, e->commit, lb, rep);                        \
        break;
                RAFTHIP_CASE_U(1) RAFTHIP_CASE_U(2) RAFTHIP_CASE_U(3) RAFTHIP_CASE_U(4)
                RAFTHIP_CASE_U(5) RAFTHIP_CASE_U(6) RAFTHIP_CASE_U(7) RAFTHIP_CASE_U(8)
#undef RAFTHIP_CASE_U
                default: return fail(RAFTHIP_EINVAL, "internal: uniform voters %u", e->uniform_voters);
            }
        } else {
            const uint8_t* m0 = static_cast<const uint8_t*>(e->in_mask);
            const uint8_t* m1 = static_cast<const uint8_t*>(e->out_mask);
            switch (e->S) {
#define RAFTHIP_CASE_M(SS)                                                                         \
    case SS:                                                                                       \
        hipLaunchKernelGGL((k_commit_masked<SS, uint8_t, MODE, kCommitBlock, kCH>), dim3(blocks),     \
                           dim3(kCommitBlock), 0, st, e->match, e->stride, G, m0, m1, 0u, e->commit,  \
                           lb, rep);                                                               \
        break;
                RAFTHIP_CASE_M(1) RAFTHIP_CASE_M(2) RAFTHIP_CASE_M(3) RAFTHIP_CASE_M(4)
                RAFTHIP_CASE_M(5) RAFTHIP_CASE_M(6) RAFTHIP_CASE_M(7) RAFTHIP_CASE_M(8)
#undef RAFTHIP_CASE_M
                default: return fail(RAFTHIP_EINVAL, "internal: slots %u", e->S);
            }
        }
    } else if (mid_commit_path(e)) {
        const u32 blocks = (u32)((G + kMidTileGroups - 1) / kMidTileGroups);
        *block_groups = kMidTileGroups;
        if (CHANGED && blocks > tiles_cap) return fail(RAFTHIP_EINVAL, "internal: report capacity");
        const uint16_t* m0 = uni ? nullptr : static_cast<const uint16_t*>(e->in_mask);
        const uint16_t* m1 = uni ? nullptr : static_cast<const uint16_t*>(e->out_mask);
        switch (e->S) {
#define RAFTHIP_CASE_MM(SS)                                                                        \
    case SS:                                                                                       \
        hipLaunchKernelGGL((k_commit_masked<SS, uint16_t, MODE, kMidBlock, 1>), dim3(blocks),      \
                           dim3(kMidBlock), 0, st, e->match, e->stride, G, m0, m1,                 \
                           uni ? e->uniform_voters : 0u, e->commit, lb, rep);                      \
        break;
            RAFTHIP_CASE_MM(9) RAFTHIP_CASE_MM(10) RAFTHIP_CASE_MM(11) RAFTHIP_CASE_MM(12)
            RAFTHIP_CASE_MM(13) RAFTHIP_CASE_MM(14) RAFTHIP_CASE_MM(15) RAFTHIP_CASE_MM(16)
#undef RAFTHIP_CASE_MM
            default: return fail(RAFTHIP_EINVAL, "internal: slots %u", e->S);
        }
    } else {
        const u32 blocks = (u32)((G + kBlock - 1) / kBlock);
        *block_groups = kBlock;
        if (CHANGED && blocks > tiles_cap) return fail(RAFTHIP_EINVAL, "internal: report capacity");
        const int uv = uni ? (int)e->uniform_voters : 0;
        if (e->wide_bitonic) {
            const size_t lds = (size_t)(kBlock / 64) * e->S * kTileStride * sizeof(u64);
#define RAFTHIP_WIDE(W, MT)                                                                        \
    hipLaunchKernelGGL((k_commit_wide<W, MT, MODE>), dim3(blocks), dim3(kBlock), lds, st,          \
                       e->match, e->stride, G, (int)e->S,                                          \
                       uni ? nullptr : static_cast<const MT*>(e->in_mask),                         \
                       uni ? nullptr : static_cast<const MT*>(e->out_mask), uv, e->commit, lb,     \
                       rep)
            if (e->S <= 16) {
                RAFTHIP_WIDE(16, uint16_t);
            } else if (e->S <= 32) {
                RAFTHIP_WIDE(32, uint32_t);
            } else {
                RAFTHIP_WIDE(64, uint64_t);
            }
#undef RAFTHIP_WIDE
        } else {
#define RAFTHIP_REGS(N, MT)                                                                        \
    hipLaunchKernelGGL((k_commit_regs<N, MT, MODE>), dim3(blocks), dim3(kBlock), 0, st,            \
                       e->match, e->stride, G, (int)e->S,                                          \
                       uni ? nullptr : static_cast<const MT*>(e->in_mask),                         \
                       uni ? nullptr : static_cast<const MT*>(e->out_mask), uv, e->commit, lb,     \
                       rep)
            if (e->S <= 24) {
                RAFTHIP_REGS(24, uint32_t);
            } else if (e->S <= 32) {
                RAFTHIP_REGS(32, uint32_t);
            } else if (e->S <= 48) {
                RAFTHIP_REGS(48, uint64_t);
            } else {
                RAFTHIP_REGS(64, uint64_t);
            }
#undef RAFTHIP_REGS
        }
    }
    HIP_TRY(hipGetLastError());
    e->stats.commit_launches++;
    e->stats.groups_evaluated += G;
    return RAFTHIP_OK;
}

// The scratch of one reporting pass (any memory the device can address; the engine's
// own buffers, or the static array of rafthip_commit_changed_static).
struct ReportBufs {
    u64* bits;
    u32* counts;
    u32* super;       // zero when the pass starts
    u32* super_next;  // zeroed by the pass for its successor; NULL: `super` is zeroed by a memset in front of the pass
    u64* total;
    u32 tiles_cap;
};

u32 super_words(u32 tiles) { return (tiles + kSuperTiles - 1) / kSuperTiles; }
// bytes of `super` + total for a plane of `stride` groups (smallest tile = kBlock), total 8-byte aligned
size_t super_bytes(u64 stride) { return round_up((size_t)super_words((u32)(stride / kBlock)) * sizeof(u32), 8) + 8; }
// the engine's own block: two such arrays and the total
size_t super_pair_bytes(u64 stride) { return 2 * (super_bytes(stride) - 8) + 8; }

int ensure_report_buffers(rafthip_engine* e, ReportBufs* rb) {
    // Sized for either tiling (a conf change may move an engine between them):
    // tiles partition [0, stride), and kBlock is the smallest tile.
    const u32 tiles = (u32)(e->stride / kBlock);
    int rc;
    if ((rc = ensure(e, e->rep_bits, (size_t)e->stride / 8, false))) return rc;
    if ((rc = ensure(e, e->rep_counts, (size_t)tiles * sizeof(u32), false))) return rc;
    if ((rc = ensure(e, e->rep_super, super_pair_bytes(e->stride), true))) return rc;  // both arrays start at zero
    const size_t one = super_bytes(e->stride) - 8;
    char* base = reinterpret_cast<char*>(e->rep_super);
    rb->bits = e->rep_bits;
    rb->counts = e->rep_counts;
    rb->super = reinterpret_cast<u32*>(base + (e->rep_flip ? one : 0));
    rb->super_next = reinterpret_cast<u32*>(base + (e->rep_flip ? 0 : one));
    rb->total = reinterpret_cast<u64*>(base + 2 * one);
    rb->tiles_cap = tiles;
    return RAFTHIP_OK;
}

// One reporting pass, asynchronous on the engine's stream: zero the super counts,
// run the commit kernel in MODE (ballot words + counts), pack the pairs of the
// changed / advanced groups to d_out (device-addressable, room for max_pairs) with
// group ids rebased by group_base, leave the total in *rb.total.  Allocates nothing.
template <int MODE>
int report_pass(rafthip_engine* e, const ReportBufs& rb, u64 group_base, Pair* d_out, u64 max_pairs, u64* total_dst = nullptr) {
    static_assert(MODE != kStore, "a reporting mode");
    u64* const total = total_dst ? total_dst : rb.total;  // e->d_scalar + k: the host reads it after its wait, no copy
    if (!rb.super_next) HIP_TRY(hipMemsetAsync(rb.super, 0, (size_t)super_words(rb.tiles_cap) * sizeof(u32), e->stream));
    u32 bg = 0;
    int rc = launch_commit<MODE>(e, &bg, Report{rb.bits, rb.counts, rb.super, rb.super_next, super_words(rb.tiles_cap)}, rb.tiles_cap);
    if (rc) {
        // nothing was launched, or something was and may have added into `super`: either way the next pass
        // must not trust the array this one was given
        if (rb.super_next) (void)hipMemsetAsync(rb.super, 0, (size_t)super_words(rb.tiles_cap) * sizeof(u32), e->stream);
        return rc;
    }
    if (rb.super_next) e->rep_flip ^= 1;  // the array this pass zeroed is the next one's
    const u32 tiles = (u32)((e->G + bg - 1) / bg);
    const bool paired = bg != (u32)kBlock;  // K1 / K2 tiles; the one-group-per-lane kernels use kBlock
    const u32 spans = paired ? bg / 128 : bg / 64;
    if (paired) {
        hipLaunchKernelGGL(k_pack_changed<true>, dim3(tiles), dim3(kBlock), 0, e->stream, rb.bits, rb.counts, rb.super,
                           tiles, spans, e->commit, group_base, d_out, max_pairs, total);
    } else {
        hipLaunchKernelGGL(k_pack_changed<false>, dim3(tiles), dim3(kBlock), 0, e->stream, rb.bits, rb.counts, rb.super,
                           tiles, spans, e->commit, group_base, d_out, max_pairs, total);
    }
    HIP_TRY(hipGetLastError());
    return RAFTHIP_OK;
}

static int upload_bits(rafthip_engine* e, void*& plane, uint64_t first, uint64_t count, const void* src) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!src) return fail(RAFTHIP_EINVAL, "src is NULL");
    DeviceGuard guard(e->device);
    const size_t mb = (size_t)e->mbytes;
    if ((rc = ensure(e, plane, (size_t)e->stride * mb))) return rc;
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(plane) + first * mb, src, count * mb, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_uploaded += count * mb;
    return RAFTHIP_OK;
}

static int read_u8(rafthip_engine* e, const uint8_t* plane, const char* what, uint64_t first,
                   uint64_t count, uint8_t* dst) {
    if (!plane) return fail(RAFTHIP_EINVAL, "%s has not been computed yet", what);
    HIP_TRY(hipMemcpyAsync(dst, plane + first, count, hipMemcpyDefault, e->stream));
    e->stats.bytes_downloaded += count;
    return RAFTHIP_OK;
}

template <bool ACKS>
static int tally_bits(rafthip_engine* e, void* bits, uint8_t*& out, const char* what, const void* keep = nullptr) {
    if (!bits) return fail(RAFTHIP_EINVAL, "%s bits have not been uploaded", what);
    int rc;
    if ((rc = ensure(e, out, e->stride))) return rc;
    if (e->uniform && e->uniform_voters == 0 && (rc = ensure_masks(e))) return rc;
    const bool uni = e->uniform && e->uniform_voters > 0;
    hipStream_t st = e->stream;
    if (e->S <= 8) {
        const u32 blocks = (u32)((e->G + (u64)kBlock * 16 - 1) / ((u64)kBlock * 16));
        hipLaunchKernelGGL((k_tally_bits_small<ACKS>), dim3(blocks), dim3(kBlock), 0, st,
                           static_cast<uint8_t*>(bits),
                           uni ? nullptr : static_cast<const uint8_t*>(e->in_mask),
                           uni ? nullptr : static_cast<const uint8_t*>(e->out_mask),
                           uni ? nullptr : static_cast<const uint8_t*>(e->learner_mask), e->S,
                           e->uniform_voters, out, static_cast<const uint8_t*>(keep));
    } else {
        const u32 blocks = (u32)((e->G + (u64)kBlock * 4 - 1) / ((u64)kBlock * 4));
#define RAFTHIP_BW(MT)                                                                              \
    hipLaunchKernelGGL((k_tally_bits_wide<MT, ACKS>), dim3(blocks), dim3(kBlock), 0, st,            \
                       static_cast<MT*>(bits), (int)e->S,                                           \
                       uni ? nullptr : static_cast<const MT*>(e->in_mask),                          \
                       uni ? nullptr : static_cast<const MT*>(e->out_mask),                         \
                       uni ? nullptr : static_cast<const MT*>(e->learner_mask),                     \
                       (int)e->uniform_voters, out, static_cast<const MT*>(keep))
        if (e->mbytes == 2) {
            RAFTHIP_BW(uint16_t);
        } else if (e->mbytes == 4) {
            RAFTHIP_BW(uint32_t);
        } else {
            RAFTHIP_BW(uint64_t);
        }
#undef RAFTHIP_BW
    }
    HIP_TRY(hipGetLastError());
    e->stats.groups_evaluated += e->G;
    return RAFTHIP_OK;
}

static int oneshot_engine(int device, uint64_t G, uint32_t S, const void* in_mask, const void* out_mask,
                          rafthip_engine** out) {
    if (!in_mask) return fail(RAFTHIP_EINVAL, "in_mask is NULL");
    rafthip_config cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.device = device;
    cfg.num_groups = G;
    cfg.num_slots = S;
    rafthip_engine* e = nullptr;
    int rc = rafthip_create(&cfg, &e);
    if (rc) return rc;
    if (out_mask) {
        rc = rafthip_upload_config(e, 0, G, in_mask, out_mask, nullptr);
    } else {
        // non-joint: the outgoing half is the zero MajorityConfig
        rc = ensure_masks(e);
        if (!rc) {
            hipError_t err = hipMemcpyAsync(e->in_mask, in_mask, G * e->mbytes, hipMemcpyDefault, e->stream);
            if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
            if (err != hipSuccess) rc = fail(RAFTHIP_EIO, "mask upload: %s", hipGetErrorString(err));
            e->uniform = false;
        }
    }
    if (rc) {
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        rafthip_destroy(e);
        memcpy(g_err, keep, sizeof(keep));
        return rc;
    }
    *out = e;
    return RAFTHIP_OK;
}

// `bytes` of the engine's small arena (rafthip_engine::h_small), 256-byte aligned: host view, device view; NULL when
// the call has used it up.
char* small_take(rafthip_engine* e, size_t bytes, char** dev) {
    static const bool off = getenv("RAFTHIP_NO_SMALL_ARENA") != nullptr;  // A/B switch for tools/small_tick_probe.py
    const size_t at = (e->small_used + 255) & ~(size_t)255;
    if (off || e->small_failed || at + bytes > kSmallArenaBytes) return nullptr;
    if (!e->h_small) {  // the first small batch of this engine (the caller has made the engine's device current)
        hipError_t err = hipHostMalloc(reinterpret_cast<void**>(&e->h_small), kSmallArenaBytes,
                                       hipHostMallocMapped | hipHostMallocCoherent);
        if (err == hipSuccess) err = hipHostGetDevicePointer(reinterpret_cast<void**>(&e->d_small), e->h_small, 0);
        if (err != hipSuccess) {  // no page-locked memory to be had: the copy commands still work
            (void)hipGetLastError();
            if (e->h_small) (void)hipHostFree(e->h_small);
            e->h_small = nullptr;
            e->small_failed = true;
            return nullptr;
        }
    }
    e->small_used = at + bytes;
    *dev = e->d_small + at;
    return e->h_small + at;
}

// 16-byte records (acks or msgs) -> a device pointer; host memory is staged
// through the engine's record buffer.
static int stage_records(rafthip_engine* e, const void* recs, uint64_t n, const void** d_recs, bool* staged) {
    hipPointerAttribute_t attr;
    *staged = false;
    if (hipPointerGetAttributes(&attr, recs) == hipSuccess) {
        if (attr.type == hipMemoryTypeDevice) {
            *d_recs = recs;
            return RAFTHIP_OK;
        }
        if (attr.type == hipMemoryTypeHost && attr.devicePointer && !getenv("RAFTHIP_NO_ZERO_COPY")) {
            // page-locked host memory (rafthip_host_alloc): the kernel streams the records over the
            // link itself — copy and compute in one pass, no staging buffer.  The caller's buffer is
            // in use until the kernel is done: report it as staged so that the call waits.
            *d_recs = attr.devicePointer;
            *staged = true;
            e->stats.bytes_uploaded += n * sizeof(Ack);
            return RAFTHIP_OK;
        }
    }
    (void)hipGetLastError();  // plain host memory reports an error on some runtimes
    if (n <= kSmallRecords) {  // a small batch: through the page-locked arena, no DMA copy (every caller waits when staged)
        e->small_used = 0;
        char* d = nullptr;
        if (char* h = small_take(e, n * sizeof(Ack), &d)) {
            memcpy(h, recs, n * sizeof(Ack));
            *d_recs = d;
            *staged = true;
            e->stats.bytes_uploaded += n * sizeof(Ack);
            return RAFTHIP_OK;
        }
    }
    if (e->ack_cap < n) {
        if (e->ack_buf) {
            HIP_TRY(hipStreamSynchronize(e->stream));
            HIP_TRY(hipFree(e->ack_buf));
            e->ack_buf = nullptr;
            e->ack_cap = 0;
        }
        void* q = nullptr;
        HIP_TRY(hipMalloc(&q, n * sizeof(Ack)));
        e->ack_buf = static_cast<Ack*>(q);
        e->ack_cap = n;
    }
    HIP_TRY(hipMemcpyAsync(e->ack_buf, recs, n * sizeof(Ack), hipMemcpyHostToDevice, e->stream));
    *d_recs = e->ack_buf;
    *staged = true;
    e->stats.bytes_uploaded += n * sizeof(Ack);
    return RAFTHIP_OK;
}

// dst as memory the pack kernel can write itself: device memory, or page-locked host
// memory through its device alias.  NULL for pageable host memory.
Pair* device_addressable(void* dst) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, dst) == hipSuccess) {
        if (attr.type == hipMemoryTypeDevice) return static_cast<Pair*>(dst);
        if (attr.type == hipMemoryTypeHost && attr.devicePointer) return static_cast<Pair*>(attr.devicePointer);
    }
    (void)hipGetLastError();  // plain host memory reports an error on some runtimes
    return nullptr;
}

// Commit pass that reports pairs (MODE kChanged / kAdvance): the reporting pass,
// then the total and — for a pageable dst — the pairs come back.
template <int MODE>
static int commit_report(rafthip_engine* e, rafthip_commit_pair* dst, uint64_t max_pairs, uint64_t* n_out) {
    if (!e || !n_out) return fail(RAFTHIP_EINVAL, "engine/count pointer is NULL");
    if (max_pairs > 0 && !dst) return fail(RAFTHIP_EINVAL, "dst is NULL with max_pairs > 0");
    DeviceGuard guard(e->device);
    ReportBufs rb;
    int rc = ensure_report_buffers(e, &rb);
    if (rc) return rc;
    if (MODE == kAdvance) {
        if ((rc = ensure(e, e->term_first, (size_t)e->stride * sizeof(u64)))) return rc;
        if ((rc = ensure(e, e->term_last, (size_t)e->stride * sizeof(u64)))) return rc;
    }
    Pair* d_out = max_pairs ? device_addressable(dst) : nullptr;
    const bool landing = max_pairs && !d_out;
    u64 cap = max_pairs;
    if (landing) {  // pageable host memory: pack on the device, copy what fits
        if ((rc = ensure(e, e->packed, (size_t)e->stride * sizeof(Pair), false))) return rc;
        d_out = e->packed;
        cap = max_pairs < e->G ? max_pairs : e->G;
    }
    if ((rc = report_pass<MODE>(e, rb, 0, d_out, cap, e->d_scalar + 1))) return rc;  // the total lands in h_scalar[1]
    const Pair* head = nullptr;  // a short report, in the page-locked arena by the time the wait returns
    if (landing) {
        e->small_used = 0;
        char* d = nullptr;
        const u32 limit = cap < kSmallPairs ? (u32)cap : (u32)kSmallPairs;
        if (char* h = small_take(e, (size_t)limit * sizeof(Pair), &d)) {
            hipLaunchKernelGGL(k_copy_head, dim3((limit + kBlock - 1) / kBlock), dim3(kBlock), 0, e->stream, e->packed,
                               e->d_scalar + 1, limit, reinterpret_cast<Pair*>(d));
            HIP_TRY(hipGetLastError());
            head = reinterpret_cast<const Pair*>(h);
        }
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    const u64 total = e->h_scalar[1];
    *n_out = total;
    const u64 ncopy = total < max_pairs ? total : max_pairs;
    if (landing && ncopy && head && ncopy <= kSmallPairs) {
        memcpy(dst, head, ncopy * sizeof(Pair));
    } else if (landing && ncopy) {
        HIP_TRY(hipMemcpyAsync(dst, e->packed, ncopy * sizeof(Pair), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    e->stats.bytes_downloaded += ncopy * sizeof(Pair);
    if (MODE == kAdvance) {
        e->stats.advanced_reported += total;
    } else {
        e->stats.changed_reported += total;
    }
    return RAFTHIP_OK;
}

bool is_device_ptr(const void* p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeDevice) return true;
    (void)hipGetLastError();  // plain host memory reports an error on some runtimes
    return false;
}

// Growable device staging buffer k (wire bytes / offsets / groups / records / statuses, aux words, action bytes, list links).
int stage_reserve(rafthip_engine* e, StageSlot k, size_t bytes) {
    if (e->stage_cap[k] >= bytes) return RAFTHIP_OK;
    if (e->stage_buf[k]) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        HIP_TRY(hipFree(e->stage_buf[k]));
        e->stage_buf[k] = nullptr;
        e->stage_cap[k] = 0;
    }
    HIP_TRY(hipMalloc(&e->stage_buf[k], bytes));
    e->stage_cap[k] = bytes;
    return RAFTHIP_OK;
}

// Input k as a device-visible pointer: device memory and page-locked host memory as they are, pageable host memory copied in
// (`small`: through the arena when it fits, see rafthip_engine::h_small).
int stage_input(rafthip_engine* e, StageSlot k, const void* src, size_t bytes, const void** dev, bool small = false) {
    if (bytes == 0) {
        *dev = src;
        return RAFTHIP_OK;
    }
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, src) == hipSuccess) {
        if (attr.type == hipMemoryTypeDevice) {
            *dev = src;
            return RAFTHIP_OK;
        }
        if (attr.type == hipMemoryTypeHost && attr.devicePointer && !getenv("RAFTHIP_NO_ZERO_COPY")) {
            *dev = attr.devicePointer;  // page-locked (rafthip_host_alloc): read over the link by the kernel itself
            e->zero_copy_pending = true;
            e->stats.bytes_uploaded += bytes;
            return RAFTHIP_OK;
        }
    }
    (void)hipGetLastError();
    if (small) {
        char* d = nullptr;
        if (char* h = small_take(e, bytes, &d)) {
            memcpy(h, src, bytes);
            e->stats.bytes_uploaded += bytes;
            *dev = d;
            return RAFTHIP_OK;
        }
    }
    int rc = stage_reserve(e, k, bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(e->stage_buf[k], src, bytes, hipMemcpyHostToDevice, e->stream));
    e->stats.bytes_uploaded += bytes;
    *dev = e->stage_buf[k];
    return RAFTHIP_OK;
}

}  // namespace

// ==========================================================================
// C ABI
// ==========================================================================
extern "C" {

uint32_t rafthip_abi_version(void) { return RAFTHIP_ABI_VERSION; }

const char* rafthip_last_error(void) { return g_err; }

int rafthip_device_count(int* count) {
    if (!count) return fail(RAFTHIP_EINVAL, "count is NULL");
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) {
        *count = 0;
        return fail(RAFTHIP_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(err));
    }
    *count = n;
    return RAFTHIP_OK;
}

int rafthip_create(const rafthip_config* cfg, rafthip_engine** out) {
    if (!cfg || !out) return fail(RAFTHIP_EINVAL, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->struct_size < sizeof(rafthip_config))
        return fail(RAFTHIP_EINVAL, "rafthip_config.struct_size %u < %zu", cfg->struct_size,
                    sizeof(rafthip_config));
    if (cfg->num_groups == 0) return fail(RAFTHIP_EINVAL, "num_groups must be >= 1");
    if (cfg->num_groups > 0xFFFFFFFFull)
        return fail(RAFTHIP_EINVAL, "num_groups %llu exceeds the 32-bit group id of rafthip_ack",
                    (u64)cfg->num_groups);
    if (cfg->num_slots == 0 || cfg->num_slots > RAFTHIP_MAX_SLOTS)
        return fail(RAFTHIP_EINVAL, "num_slots %u outside [1, %u]", cfg->num_slots, RAFTHIP_MAX_SLOTS);
    if (cfg->flags & ~(RAFTHIP_FLAG_WIDE_BITONIC | RAFTHIP_FLAG_SCATTER_ATOMIC | RAFTHIP_FLAG_SCATTER_PARTITION |
                       RAFTHIP_FLAG_STEP_LINKED | RAFTHIP_FLAG_STEP_PARTITION))
        return fail(RAFTHIP_EINVAL, "unknown flags 0x%x", cfg->flags);
    if ((cfg->flags & RAFTHIP_FLAG_SCATTER_ATOMIC) && (cfg->flags & RAFTHIP_FLAG_SCATTER_PARTITION))
        return fail(RAFTHIP_EINVAL, "RAFTHIP_FLAG_SCATTER_ATOMIC and RAFTHIP_FLAG_SCATTER_PARTITION exclude each other");
    if ((cfg->flags & RAFTHIP_FLAG_STEP_LINKED) && (cfg->flags & RAFTHIP_FLAG_STEP_PARTITION))
        return fail(RAFTHIP_EINVAL, "RAFTHIP_FLAG_STEP_LINKED and RAFTHIP_FLAG_STEP_PARTITION exclude each other");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(RAFTHIP_ENODEV, "no HIP device available (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(RAFTHIP_ENODEV, "device %d outside [0, %d)", cfg->device, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RAFTHIP_ENODEV, "device %d is %s; librafthip.so carries gfx950 code only",
                    cfg->device, prop.gcnArchName);

    DeviceGuard guard(cfg->device);
    if (!guard.ok) return fail(RAFTHIP_ENODEV, "hipSetDevice(%d) failed", cfg->device);
    rafthip_engine* e = new (std::nothrow) rafthip_engine();
    if (!e) return fail(RAFTHIP_ENOMEM, "host allocation failed");
    e->device = cfg->device;
    e->G = cfg->num_groups;
    e->S = cfg->num_slots;
    e->stride = round_up(e->G, kPlaneAlign);
    e->mbytes = mask_bytes_for(e->S);
    e->wide_bitonic = (cfg->flags & RAFTHIP_FLAG_WIDE_BITONIC) != 0;
    e->scatter_mode = (cfg->flags & RAFTHIP_FLAG_SCATTER_ATOMIC) ? 1 : (cfg->flags & RAFTHIP_FLAG_SCATTER_PARTITION) ? 2 : 0;
    e->step_mode = (cfg->flags & RAFTHIP_FLAG_STEP_LINKED) ? 1 : (cfg->flags & RAFTHIP_FLAG_STEP_PARTITION) ? 2 : 0;
    int rc = RAFTHIP_OK;
    do {
        if (cfg->stream) {
            e->stream = static_cast<hipStream_t>(cfg->stream);
        } else {
            hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
            if (err != hipSuccess) {
                rc = fail(RAFTHIP_EIO, "hipStreamCreate: %s", hipGetErrorString(err));
                break;
            }
            e->own_stream = true;
        }
        void* scratch_ptr = nullptr;  // set-up of rafthip_commit_changed_static: once per device
        if ((rc = static_scratch(e->device, &scratch_ptr))) break;
        if ((rc = ensure(e, e->match, (size_t)e->S * e->stride * sizeof(u64)))) break;
        if ((rc = ensure(e, e->commit, (size_t)e->stride * sizeof(u64)))) break;
        if ((rc = ensure(e, e->d_rejected, 4 * sizeof(u64)))) break;
        hipError_t err = hipHostMalloc(reinterpret_cast<void**>(&e->h_scalar), 4 * sizeof(u64), hipHostMallocDefault);
        if (err == hipSuccess) err = hipHostGetDevicePointer(reinterpret_cast<void**>(&e->d_scalar), e->h_scalar, 0);
        if (err != hipSuccess) {
            rc = fail(RAFTHIP_ENOMEM, "hipHostMalloc: %s", hipGetErrorString(err));
            break;
        }
        err = hipStreamSynchronize(e->stream);
        if (err != hipSuccess) rc = fail(RAFTHIP_EIO, "hipStreamSynchronize: %s", hipGetErrorString(err));
    } while (0);
    if (rc) {
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        rafthip_destroy(e);
        memcpy(g_err, keep, sizeof(keep));
        return rc;
    }
    *out = e;
    return RAFTHIP_OK;
}

int rafthip_destroy(rafthip_engine* e) {
    if (!e) return RAFTHIP_OK;
    DeviceGuard guard(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    void* bufs[] = {e->in_mask, e->out_mask,   e->learner_mask, e->match,    e->commit, e->votes,
                    e->active,  e->acks,       e->vote_result,  e->granted,  e->rejected, e->qactive,
                    e->ack_result, e->rep_bits, e->packed,      e->rep_counts, e->rep_super, e->ack_buf,
                    e->d_rejected, e->self_mask, e->term_first, e->term_last, e->hb_commit, e->in_term,
                    e->wire_hdr, e->d_wire_counts, e->pp.rec, e->pp.ring,
                    e->first_index, e->vote_first, e->part_a, e->part_b, e->part_words,
                    e->tick_el, e->tick_hb, e->tick_rt, e->tick_role, e->tick_flags, e->tick_action, e->d_tick_counts, e->tick_ev};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (void* b : e->stage_buf)
        if (b) (void)hipFree(b);
    if (e->h_scalar) (void)hipHostFree(e->h_scalar);
    if (e->h_small) (void)hipHostFree(e->h_small);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return RAFTHIP_OK;
}

int rafthip_host_alloc(void** ptr, uint64_t bytes) {
    if (!ptr) return fail(RAFTHIP_EINVAL, "ptr is NULL");
    *ptr = nullptr;
    if (bytes == 0) return RAFTHIP_OK;
    HIP_TRY(hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
    return RAFTHIP_OK;
}

int rafthip_host_free(void* ptr) {
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return RAFTHIP_OK;
}

int rafthip_synchronize(rafthip_engine* e) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RAFTHIP_OK;
}

int rafthip_mask_bytes(const rafthip_engine* e) { return e ? e->mbytes : RAFTHIP_EINVAL; }

int rafthip_stats(rafthip_engine* e, rafthip_stats_t* out) {
    if (!e || !out) return fail(RAFTHIP_EINVAL, "engine/out is NULL");
    DeviceGuard guard(e->device);
    u64 host[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(host, e->d_rejected, sizeof(host), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.acks_rejected = host[0];
    e->stats.msgs_dropped = host[1];
    e->stats.msgs_ignored = host[2];
    *out = e->stats;
    return RAFTHIP_OK;
}

// ---- configuration -------------------------------------------------------

int rafthip_set_uniform_config(rafthip_engine* e, uint32_t voters) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    if (voters > e->S) return fail(RAFTHIP_EINVAL, "voters %u > num_slots %u", voters, e->S);
    e->uniform = true;
    e->uniform_voters = voters;
    if (voters == 0 && e->in_mask) {
        DeviceGuard guard(e->device);
        const size_t b = (size_t)e->stride * e->mbytes;
        HIP_TRY(hipMemsetAsync(e->in_mask, 0, b, e->stream));
        HIP_TRY(hipMemsetAsync(e->out_mask, 0, b, e->stream));
        HIP_TRY(hipMemsetAsync(e->learner_mask, 0, b, e->stream));
    }
    return RAFTHIP_OK;
}

int rafthip_upload_config(rafthip_engine* e, uint64_t first, uint64_t count, const void* in_mask,
                          const void* out_mask, const void* learner_mask) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!in_mask || !out_mask) return fail(RAFTHIP_EINVAL, "in_mask/out_mask is NULL");
    DeviceGuard guard(e->device);
    const bool had = e->in_mask != nullptr;
    if ((rc = ensure_masks(e))) return rc;
    const size_t mb = (size_t)e->mbytes;
    if (e->uniform && e->uniform_voters > 0 && (first != 0 || count != e->G)) {
        // Switching from the uniform fast path with a partial upload: the rest
        // of the groups keep the uniform config, so materialise it first.
        const u64 vm = e->uniform_voters >= 64 ? ~0ull : ((1ull << e->uniform_voters) - 1ull);
        const u64 lm = (e->S >= 64 ? ~0ull : ((1ull << e->S) - 1ull)) & ~vm;
        // byte-pattern fill only works for 1-byte masks; wider masks go through a host staging row
        if (mb == 1) {
            HIP_TRY(hipMemsetAsync(e->in_mask, (int)(vm & 0xFF), e->stride, e->stream));
            HIP_TRY(hipMemsetAsync(e->out_mask, 0, e->stride, e->stream));
            HIP_TRY(hipMemsetAsync(e->learner_mask, (int)(lm & 0xFF), e->stride, e->stream));
        } else {
            const size_t bytes = (size_t)e->G * mb;
            unsigned char* tmp = new (std::nothrow) unsigned char[bytes * 2];
            if (!tmp) return fail(RAFTHIP_ENOMEM, "host allocation failed");
            for (u64 g = 0; g < e->G; ++g) {
                memcpy(tmp + g * mb, &vm, mb);
                memcpy(tmp + bytes + g * mb, &lm, mb);
            }
            hipError_t e1 = hipMemcpyAsync(e->in_mask, tmp, bytes, hipMemcpyHostToDevice, e->stream);
            hipError_t e2 = hipMemcpyAsync(e->learner_mask, tmp + bytes, bytes, hipMemcpyHostToDevice, e->stream);
            (void)hipStreamSynchronize(e->stream);
            delete[] tmp;
            HIP_TRY(e1);
            HIP_TRY(e2);
            HIP_TRY(hipMemsetAsync(e->out_mask, 0, (size_t)e->stride * mb, e->stream));
        }
    }
    (void)had;
    char* d0 = static_cast<char*>(e->in_mask) + first * mb;
    char* d1 = static_cast<char*>(e->out_mask) + first * mb;
    char* d2 = static_cast<char*>(e->learner_mask) + first * mb;
    HIP_TRY(hipMemcpyAsync(d0, in_mask, count * mb, hipMemcpyDefault, e->stream));
    HIP_TRY(hipMemcpyAsync(d1, out_mask, count * mb, hipMemcpyDefault, e->stream));
    if (learner_mask) {
        HIP_TRY(hipMemcpyAsync(d2, learner_mask, count * mb, hipMemcpyDefault, e->stream));
        e->have_learners = true;
    } else {
        HIP_TRY(hipMemsetAsync(d2, 0, count * mb, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));  // source buffers may be pageable host memory
    e->uniform = false;
    e->stats.bytes_uploaded += count * mb * (learner_mask ? 3 : 2);
    return RAFTHIP_OK;
}

// ---- state ---------------------------------------------------------------

int rafthip_upload_match(rafthip_engine* e, uint64_t first, uint64_t count, const uint64_t* src,
                         uint64_t src_stride) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!src) return fail(RAFTHIP_EINVAL, "src is NULL");
    if (src_stride < count) return fail(RAFTHIP_EINVAL, "src_stride %llu < count %llu", (u64)src_stride, (u64)count);
    DeviceGuard guard(e->device);
    if ((rc = copy_rows(e, e->match + first, e->stride * 8, src, src_stride * 8, count * 8, e->S))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_uploaded += count * 8 * e->S;
    return RAFTHIP_OK;
}

int rafthip_reset_match(rafthip_engine* e, uint64_t first, uint64_t count) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (count == 0) return RAFTHIP_OK;
    DeviceGuard guard(e->device);
    HIP_TRY(hipMemset2DAsync(e->match + first, e->stride * 8, 0, count * 8, e->S, e->stream));
    return RAFTHIP_OK;
}

int rafthip_read_match(rafthip_engine* e, uint64_t first, uint64_t count, uint64_t* dst,
                       uint64_t dst_stride) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!dst) return fail(RAFTHIP_EINVAL, "dst is NULL");
    if (dst_stride < count) return fail(RAFTHIP_EINVAL, "dst_stride %llu < count %llu", (u64)dst_stride, (u64)count);
    DeviceGuard guard(e->device);
    if ((rc = copy_rows(e, dst, dst_stride * 8, e->match + first, e->stride * 8, count * 8, e->S))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_downloaded += count * 8 * e->S;
    return RAFTHIP_OK;
}

int rafthip_upload_votes(rafthip_engine* e, uint64_t first, uint64_t count, const uint8_t* src,
                         uint64_t src_stride) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!src) return fail(RAFTHIP_EINVAL, "src is NULL");
    if (src_stride < count) return fail(RAFTHIP_EINVAL, "src_stride %llu < count %llu", (u64)src_stride, (u64)count);
    DeviceGuard guard(e->device);
    if ((rc = ensure(e, e->votes, (size_t)e->S * e->stride))) return rc;
    if ((rc = copy_rows(e, e->votes + first, e->stride, src, src_stride, count, e->S))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_uploaded += count * e->S;
    return RAFTHIP_OK;
}

int rafthip_read_votes(rafthip_engine* e, uint64_t first, uint64_t count, uint8_t* dst, uint64_t dst_stride) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!dst) return fail(RAFTHIP_EINVAL, "dst is NULL");
    if (dst_stride < count) return fail(RAFTHIP_EINVAL, "dst_stride %llu < count %llu", (u64)dst_stride, (u64)count);
    DeviceGuard guard(e->device);
    if ((rc = ensure(e, e->votes, (size_t)e->S * e->stride))) return rc;
    if ((rc = copy_rows(e, dst, dst_stride, e->votes + first, e->stride, count, e->S))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_downloaded += count * e->S;
    return RAFTHIP_OK;
}

int rafthip_reset_votes(rafthip_engine* e, uint64_t first, uint64_t count) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    DeviceGuard guard(e->device);
    if ((rc = ensure(e, e->votes, (size_t)e->S * e->stride))) return rc;
    if (count == 0) return RAFTHIP_OK;
    HIP_TRY(hipMemset2DAsync(e->votes + first, e->stride, 0, count, e->S, e->stream));
    return RAFTHIP_OK;
}

int rafthip_upload_active(rafthip_engine* e, uint64_t first, uint64_t count, const void* src) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    return upload_bits(e, e->active, first, count, src);
}

int rafthip_upload_acks(rafthip_engine* e, uint64_t first, uint64_t count, const void* src) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    return upload_bits(e, e->acks, first, count, src);
}

int rafthip_upload_commit(rafthip_engine* e, uint64_t first, uint64_t count, const uint64_t* src) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!src) return fail(RAFTHIP_EINVAL, "src is NULL");
    DeviceGuard guard(e->device);
    HIP_TRY(hipMemcpyAsync(e->commit + first, src, count * 8, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_uploaded += count * 8;
    return RAFTHIP_OK;
}

// ---- compute -------------------------------------------------------------

}  // extern "C"

namespace {
// Tile of K5p for this engine: the largest power of two of groups whose S rows fit 72 KB of LDS.
u32 part_tile_shift(const rafthip_engine* e) {
    u32 shift = 10;
    while (((size_t)e->S << shift) * sizeof(u64) > 72u * 1024u) --shift;
    return shift;
}

// Whether this batch of device-resident records takes the partitioned form (scatter_part.h).  It pays once the
// batch touches a fair share of the cells: its cost is ~5 passes over the records plus one over all of Match,
// K5's is one scattered atomic per record (the break-even is near one ack per six cells).
bool part_wanted(const rafthip_engine* e, u64 n) {
    if (e->scatter_mode == 1 || getenv("RAFTHIP_SCATTER_ATOMIC")) return false;
    const u32 tiles = (u32)(e->stride >> part_tile_shift(e));
    if (round_up(tiles, 1u << kPartLB) > kPartMaxTiles || n > kMaxRecordsPerLaunch) return false;
    if (e->scatter_mode == 2) return true;
    return n >= (1ull << 21) && n * 6 >= (u64)e->S * e->stride;
}

// The scratch the partition passes of K5p and K9p share (one engine, one stream: never at the same time): two
// record buffers of n 16-byte records and one block of words.
int part_scratch(rafthip_engine* e, u64 n, PartBufs* out) {
    int rc;
    if (e->part_cap < n) {
        if (e->part_a || e->part_b) {
            HIP_TRY(hipStreamSynchronize(e->stream));
            if (e->part_a) HIP_TRY(hipFree(e->part_a));
            e->part_a = nullptr;
            if (e->part_b) HIP_TRY(hipFree(e->part_b));
            e->part_b = nullptr;
            e->part_cap = 0;
        }
        if ((rc = ensure(e, e->part_a, (size_t)n * sizeof(Ack), false))) return rc;
        if ((rc = ensure(e, e->part_b, (size_t)n * sizeof(Ack), false))) return rc;
        e->part_cap = n;
    }
    // one block of words: counts (zero between calls), bases, cursors, bucket cursors, chunk table, flag
    const size_t words = (size_t)kPartMaxTiles * 3 + 1 + kPartMaxBuckets * 2 + 1 + 1;
    if ((rc = ensure(e, e->part_words, words * sizeof(u32), true))) return rc;
    PartBufs pb;
    pb.tile_count = e->part_words;
    pb.tile_base = pb.tile_count + kPartMaxTiles;
    pb.tile_cursor = pb.tile_base + kPartMaxTiles + 1;
    pb.bucket_cursor = pb.tile_cursor + kPartMaxTiles;
    pb.chunk_first = pb.bucket_cursor + kPartMaxBuckets;
    pb.flag = pb.chunk_first + kPartMaxBuckets + 1;
    *out = pb;
    if (e->part_dirty) HIP_TRY(hipMemsetAsync(e->part_words, 0, words * sizeof(u32), e->stream));
    e->part_dirty = true;  // until the caller has put every pass on the stream (part_done)
    return RAFTHIP_OK;
}
void part_done(rafthip_engine* e) { e->part_dirty = false; }

int scatter_partitioned(rafthip_engine* e, const Ack* d_recs, u64 n) {
    const u32 shift = part_tile_shift(e);
    const u32 tiles = (u32)(e->stride >> shift);
    const u32 tiles_pad = (u32)round_up(tiles, 1u << kPartLB), nb = tiles_pad >> kPartLB;
    int rc;
    PartBufs pb;
    if ((rc = part_scratch(e, n, &pb))) return rc;
    const size_t lds_count = (size_t)tiles_pad * sizeof(u32), lds_stage = (size_t)kPartChunk * sizeof(Ack),
                 lds_cells = ((size_t)e->S << shift) * sizeof(u64);
    if (!e->part_attr) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_count), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(kPartMaxTiles * sizeof(u32))));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_scan), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(kPartMaxTiles * sizeof(u32))));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_partition<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stage));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_partition<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stage));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_apply), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        e->part_attr = true;
    }
    hipStream_t st = e->stream;
    const u32 count_wgs = 512;
    const u64 slice = (n + count_wgs - 1) / count_wgs;
    hipLaunchKernelGGL(k_part_count, dim3(count_wgs), dim3(kPartCountBlock), lds_count, st, d_recs, n, slice, e->G, e->S, shift, tiles_pad,
                       pb.tile_count, e->d_rejected);
    hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(1024), lds_count, st, pb, tiles_pad, nb);
    const u64 chunks1 = (n + kPartChunk - 1) / kPartChunk, chunks2 = chunks1 + nb;  // pass 2: an upper bound, the table has the count
    const u32 persistent = 256;  // one workgroup per CU (128 KB of LDS each)
    hipLaunchKernelGGL(k_part_partition<1>, dim3((u32)(chunks1 < persistent ? chunks1 : persistent)), dim3(kPartBlock), lds_stage, st,
                       d_recs, n, e->G, e->S, shift, nb, pb, e->part_a);
    hipLaunchKernelGGL(k_part_partition<2>, dim3((u32)(chunks2 < persistent ? chunks2 : persistent)), dim3(kPartBlock), lds_stage, st,
                       static_cast<const Ack*>(e->part_a), n, e->G, e->S, shift, nb, pb, e->part_b);
    hipLaunchKernelGGL(k_part_apply, dim3(tiles), dim3(kPartApplyBlock), lds_cells, st, e->match, e->stride, e->S, shift,
                       static_cast<const Ack*>(e->part_b), pb);
    // a batch the scan found too skewed for tiles: K5's atomics on the original records (decided on the device)
    const u64 flagged_wgs = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_scatter_acks_flagged, dim3((u32)(flagged_wgs < kPartFlaggedGrid ? flagged_wgs : kPartFlaggedGrid)), dim3(kBlock), 0, st, e->match, e->stride, e->G, e->S,
                       d_recs, n, static_cast<const u32*>(pb.flag));
    HIP_TRY(hipGetLastError());
    part_done(e);
    return RAFTHIP_OK;
}
}  // namespace

extern "C" {

int rafthip_scatter_acks(rafthip_engine* e, const rafthip_ack* recs, uint64_t n) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    if (n == 0) return RAFTHIP_OK;
    if (!recs) return fail(RAFTHIP_EINVAL, "recs is NULL");
    DeviceGuard guard(e->device);
    const void* d_recs = nullptr;
    bool staged = false;
    int rc = stage_records(e, recs, n, &d_recs, &staged);
    if (rc) return rc;
    // device-resident records (the caller's, or the engine's staged copy of pageable memory — not page-locked memory,
    // which the kernels read over the link: one pass over it is all the link allows)
    if (is_device_ptr(d_recs) && part_wanted(e, n)) {
        if ((rc = scatter_partitioned(e, static_cast<const Ack*>(d_recs), n))) return rc;
        if (staged) HIP_TRY(hipStreamSynchronize(e->stream));
        e->stats.acks_applied += n;
        return RAFTHIP_OK;
    }
    for (u64 off = 0; off < n; off += kMaxRecordsPerLaunch) {  // a launch stays below 2^32 lanes
        const u64 m = n - off < kMaxRecordsPerLaunch ? n - off : kMaxRecordsPerLaunch;
        const u32 blocks = (u32)((m + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_scatter_acks, dim3(blocks), dim3(kBlock), 0, e->stream, e->match, e->stride, e->G,
                           e->S, static_cast<const Ack*>(d_recs) + off, m, e->d_rejected);
        HIP_TRY(hipGetLastError());
    }
    if (staged) HIP_TRY(hipStreamSynchronize(e->stream));  // caller may reuse its host buffer
    e->stats.acks_applied += n;
    return RAFTHIP_OK;
}

int rafthip_merge_acks(rafthip_engine* e, uint64_t first, uint64_t count, const uint64_t* acks,
                       uint64_t src_stride) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!acks) return fail(RAFTHIP_EINVAL, "acks is NULL");
    if (src_stride < count) return fail(RAFTHIP_EINVAL, "src_stride %llu < count %llu", (u64)src_stride, (u64)count);
    if (count == 0) return RAFTHIP_OK;
    DeviceGuard guard(e->device);
    const u64* d_acks = reinterpret_cast<const u64*>(acks);
    u64 d_stride = src_stride;
    bool staged = false;
    if (!is_device_ptr(acks)) {  // host plane: staged through the engine's growable buffer, packed to pitch = count
        d_stride = (count + 1) & ~1ull;
        if ((rc = stage_reserve(e, kStageAux, (size_t)d_stride * e->S * sizeof(u64)))) return rc;
        if ((rc = copy_rows(e, e->stage_buf[kStageAux], d_stride * 8, acks, src_stride * 8, count * 8, e->S))) return rc;
        d_acks = static_cast<const u64*>(e->stage_buf[kStageAux]);
        e->stats.bytes_uploaded += count * 8 * e->S;
        staged = true;
    }
    const u32 blocks = (u32)(((count + 1) / 2 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_merge_acks, dim3(blocks), dim3(kBlock), 0, e->stream, e->match, e->stride, d_acks, d_stride,
                       (u64)first, (u64)count, e->S);
    HIP_TRY(hipGetLastError());
    if (staged) HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.acks_applied += count * e->S;
    return RAFTHIP_OK;
}

int rafthip_record_votes(rafthip_engine* e, const rafthip_msg* msgs, uint64_t n) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    if (n == 0) return RAFTHIP_OK;
    if (!msgs) return fail(RAFTHIP_EINVAL, "msgs is NULL");
    if (n > (1ull << 31)) return fail(RAFTHIP_EINVAL, "more than 2^31 vote records in one batch: split it");
    DeviceGuard guard(e->device);
    int rc;
    if ((rc = ensure(e, e->votes, (size_t)e->S * e->stride))) return rc;
    if ((rc = ensure(e, e->vote_first, (size_t)e->S * e->stride * sizeof(u32)))) return rc;
    const void* d_recs = nullptr;
    bool staged = false;
    if ((rc = stage_records(e, msgs, n, &d_recs, &staged))) return rc;
    const u32 blocks = (u32)((n + kBlock - 1) / kBlock);
    const Msg* m = static_cast<const Msg*>(d_recs);
    hipLaunchKernelGGL(k_record_votes<1>, dim3(blocks), dim3(kBlock), 0, e->stream, e->votes, e->vote_first, e->stride,
                       e->G, e->S, m, (u32)n, e->d_rejected + 1);
    hipLaunchKernelGGL(k_record_votes<2>, dim3(blocks), dim3(kBlock), 0, e->stream, e->votes, e->vote_first, e->stride,
                       e->G, e->S, m, (u32)n, e->d_rejected + 1);
    hipLaunchKernelGGL(k_record_votes<3>, dim3(blocks), dim3(kBlock), 0, e->stream, e->votes, e->vote_first, e->stride,
                       e->G, e->S, m, (u32)n, e->d_rejected + 1);
    HIP_TRY(hipGetLastError());
    if (staged) HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.msgs_ingested += n;
    return RAFTHIP_OK;
}

int rafthip_ingest_msgs(rafthip_engine* e, const rafthip_msg* msgs, uint64_t n) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    if (n == 0) return RAFTHIP_OK;
    if (!msgs) return fail(RAFTHIP_EINVAL, "msgs is NULL");
    DeviceGuard guard(e->device);
    int rc;
    const size_t mb = (size_t)e->stride * e->mbytes;
    if ((rc = ensure(e, e->active, mb))) return rc;
    if ((rc = ensure(e, e->acks, mb))) return rc;
    if ((rc = ensure(e, e->term_last, (size_t)e->stride * sizeof(u64)))) return rc;
    if ((rc = ensure(e, e->term_first, (size_t)e->stride * sizeof(u64)))) return rc;
    const void* d_recs = nullptr;
    bool staged = false;
    if ((rc = stage_records(e, msgs, n, &d_recs, &staged))) return rc;
    for (u64 off = 0; off < n; off += kMaxRecordsPerLaunch) {
        const u64 m = n - off < kMaxRecordsPerLaunch ? n - off : kMaxRecordsPerLaunch;
        const u32 blocks = (u32)((m + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_ingest_msgs, dim3(blocks), dim3(kBlock), 0, e->stream, e->match, e->stride, e->G,
                           e->S, e->active, e->acks, e->mbytes, e->term_last,
                           static_cast<const Msg*>(d_recs) + off, m, e->d_rejected + 1);
        HIP_TRY(hipGetLastError());
    }
    if (staged) HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.msgs_ingested += n;
    return RAFTHIP_OK;
}

// ---- wire ingest -----------------------------------------------------------

namespace {

int ensure_wire_hdr(rafthip_engine* e) {
    return ensure(e, e->wire_hdr, (size_t)e->G * wire_hdr_words(e->S) * sizeof(u64));
}

// `bytes` of host or device memory as a device pointer valid until *tmp is freed.
int device_view(rafthip_engine* e, const void* src, size_t bytes, void** tmp, const void** dev) {
    *tmp = nullptr;
    if (is_device_ptr(src)) {
        *dev = src;
        return RAFTHIP_OK;
    }
    HIP_TRY(hipMalloc(tmp, bytes));
    hipError_t err = hipMemcpyAsync(*tmp, src, bytes, hipMemcpyHostToDevice, e->stream);
    if (err != hipSuccess) {
        (void)hipFree(*tmp);
        *tmp = nullptr;
        HIP_TRY(err);
    }
    *dev = *tmp;
    return RAFTHIP_OK;
}

int finish_and_free(rafthip_engine* e, void* a, void* b) {
    const hipError_t err = hipGetLastError();
    const hipError_t err2 = hipStreamSynchronize(e->stream);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    HIP_TRY(err);
    HIP_TRY(err2);
    return RAFTHIP_OK;
}

}  // namespace

int rafthip_upload_ids(rafthip_engine* e, uint64_t first, uint64_t count, const uint64_t* ids,
                       uint64_t src_stride) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!ids) return fail(RAFTHIP_EINVAL, "ids is NULL");
    if (src_stride < count) return fail(RAFTHIP_EINVAL, "src_stride %llu < count %llu", (u64)src_stride, (u64)count);
    DeviceGuard guard(e->device);
    if ((rc = ensure_wire_hdr(e))) return rc;
    e->have_ids = true;
    if (count == 0) return RAFTHIP_OK;
    const u32 H = wire_hdr_words(e->S);
    void* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t)count * e->S * sizeof(u64)));
    hipError_t err = hipMemcpy2DAsync(tmp, count * 8, ids, src_stride * 8, count * 8, e->S, hipMemcpyDefault, e->stream);
    if (err != hipSuccess) {
        (void)hipFree(tmp);
        HIP_TRY(err);
    }
    hipLaunchKernelGGL(k_hdr_set_ids, dim3((u32)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream,
                       static_cast<const u64*>(tmp), (u64)count, e->S, H, e->wire_hdr + first * H);
    if ((rc = finish_and_free(e, tmp, nullptr))) return rc;
    e->stats.bytes_uploaded += count * 8 * e->S;
    return RAFTHIP_OK;
}

int rafthip_upload_term(rafthip_engine* e, uint64_t first, uint64_t count, const uint64_t* term) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!term) return fail(RAFTHIP_EINVAL, "term is NULL");
    DeviceGuard guard(e->device);
    if ((rc = ensure_wire_hdr(e))) return rc;
    e->have_term = true;
    if (count == 0) return RAFTHIP_OK;
    const u32 H = wire_hdr_words(e->S);
    void* tmp = nullptr;
    const void* d_term = nullptr;
    if ((rc = device_view(e, term, count * 8, &tmp, &d_term))) return rc;
    hipLaunchKernelGGL(k_hdr_set_field, dim3((u32)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream,
                       static_cast<const u64*>(d_term), static_cast<const uint8_t*>(nullptr), (u64)count, H, kHdrTerm,
                       0u, 0u, 0xFFu, e->wire_hdr + first * H);
    if ((rc = finish_and_free(e, tmp, nullptr))) return rc;
    e->stats.bytes_uploaded += count * 8;
    return RAFTHIP_OK;
}

int rafthip_upload_read_ctx(rafthip_engine* e, uint64_t first, uint64_t count, const uint64_t* ctx,
                            const uint8_t* ctx_len) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!ctx || !ctx_len) return fail(RAFTHIP_EINVAL, "ctx/ctx_len is NULL");
    DeviceGuard guard(e->device);
    if (!is_device_ptr(ctx_len)) {  // a row holds 8 context bytes; device-resident lengths are clamped by the kernel
        for (u64 g = 0; g < count; ++g)
            if (ctx_len[g] > 8)
                return fail(RAFTHIP_EINVAL, "ctx_len[%llu] = %u: a pending ReadIndex context of more than 8 bytes is not supported",
                            (u64)g, (unsigned)ctx_len[g]);
    }
    if ((rc = ensure_wire_hdr(e))) return rc;
    if (count == 0) return RAFTHIP_OK;
    const u32 H = wire_hdr_words(e->S);
    void *tmp = nullptr, *tmp8 = nullptr;
    const void *d_ctx = nullptr, *d_len = nullptr;
    if ((rc = device_view(e, ctx, count * 8, &tmp, &d_ctx))) return rc;
    if ((rc = device_view(e, ctx_len, count, &tmp8, &d_len))) {
        if (tmp) (void)hipFree(tmp);
        return rc;
    }
    hipLaunchKernelGGL(k_hdr_set_field, dim3((u32)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream,
                       static_cast<const u64*>(d_ctx), static_cast<const uint8_t*>(d_len), (u64)count, H, kHdrCtx,
                       kHdrCtxLen, 0u, 8u, e->wire_hdr + first * H);
    if ((rc = finish_and_free(e, tmp, tmp8))) return rc;
    e->stats.bytes_uploaded += count * 9;
    return RAFTHIP_OK;
}

int rafthip_upload_campaign(rafthip_engine* e, uint64_t first, uint64_t count, const uint8_t* resp_type) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    DeviceGuard guard(e->device);
    if ((rc = ensure_wire_hdr(e))) return rc;
    if (!resp_type) {  // nobody campaigns any more: the wire path skips the vote passes again
        if (first != 0 || count != e->G) return fail(RAFTHIP_EINVAL, "resp_type NULL clears every group: pass the whole range");
        e->campaigning = false;
    }
    if (count == 0) return RAFTHIP_OK;
    const u32 H = wire_hdr_words(e->S);
    void* tmp = nullptr;
    const void* d_type = nullptr;
    if ((rc = ensure(e, e->votes, (size_t)e->S * e->stride))) return rc;
    if ((rc = ensure(e, e->vote_first, (size_t)e->S * e->stride * sizeof(u32)))) return rc;
    if (resp_type) {
        if ((rc = device_view(e, resp_type, count, &tmp, &d_type))) return rc;
    } else {
        HIP_TRY(hipMalloc(&tmp, count));
        if (hipError_t err = hipMemsetAsync(tmp, 0, count, e->stream)) {
            (void)hipFree(tmp);
            return fail(RAFTHIP_EIO, "hipMemsetAsync: %s", hipGetErrorString(err));
        }
        d_type = tmp;
    }
    hipLaunchKernelGGL(k_hdr_set_field, dim3((u32)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream,
                       static_cast<const u64*>(nullptr), static_cast<const uint8_t*>(d_type), (u64)count, H, 0u,
                       kHdrCtxLen, kHdrCampaignShift, 0xFFu, e->wire_hdr + first * H);
    if ((rc = finish_and_free(e, tmp, nullptr))) return rc;
    if (resp_type) e->campaigning = true;
    e->stats.bytes_uploaded += resp_type ? count : 0;
    return RAFTHIP_OK;
}

static int progress_passes(rafthip_engine* e, const Msg* d_msgs, const u64* d_aux, uint64_t n, uint8_t* d_act,
                           bool linked, bool part);
static bool prog_part_wanted(const rafthip_engine* e, u64 n);
// Before a link pass: if an earlier call failed between its link and apply passes, zero every list head.
static int clear_stale_heads(rafthip_engine* e) {
    if (!e->heads_dirty || !e->pp.rec) return RAFTHIP_OK;
    const u64 cells = (u64)e->S * e->stride;
    hipLaunchKernelGGL(k_progress_clear_heads, dim3((u32)((cells + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream,
                       e->pp.rec, cells);
    HIP_TRY(hipGetLastError());
    e->heads_dirty = false;
    return RAFTHIP_OK;
}

// Decode + classify n serialized messages; actions == NULL: step them as K7 does
// (rafthip_ingest_wire); else hand the decoded records to the Progress stepper
// (rafthip_step_wire).
static int wire_run(rafthip_engine* e, const uint8_t* bytes, const uint64_t* offsets, const uint32_t* groups,
                    uint64_t n, rafthip_msg* out_msgs, uint8_t* out_status, uint64_t* status_counts, bool to_progress,
                    uint8_t* actions) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    if (to_progress && !e->prog) return fail(RAFTHIP_EINVAL, "rafthip_progress_config must come first");
    if ((to_progress || e->campaigning) && n > (1ull << 31))
        return fail(RAFTHIP_EINVAL, "more than 2^31 events in one batch: split it");
    if (status_counts) memset(status_counts, 0, RAFTHIP_WIRE_STATUSES * sizeof(uint64_t));
    if (n == 0) return RAFTHIP_OK;
    if (!offsets || !groups) return fail(RAFTHIP_EINVAL, "offsets/groups is NULL");
    if (!e->have_ids || !e->have_term)
        return fail(RAFTHIP_EINVAL, "rafthip_upload_ids and rafthip_upload_term must come first");
    DeviceGuard guard(e->device);
    int rc;
    const size_t mb = (size_t)e->stride * e->mbytes;
    if ((rc = ensure(e, e->active, mb))) return rc;
    if ((rc = ensure(e, e->acks, mb))) return rc;
    if ((rc = ensure(e, e->d_wire_counts, RAFTHIP_WIRE_STATUSES * sizeof(u64)))) return rc;

    // a small batch from / to pageable host memory goes through the page-locked arena, not through DMA copies
    const bool small = n <= kSmallBatch;
    e->small_used = 0;
    struct { void* dst; const void* src; size_t bytes; } after[5];  // arena -> caller, once the stream has drained
    int n_after = 0;

    // total byte length = offsets[n] (the kernel bounds every message by it)
    const void* d_off = nullptr;
    u64 nbytes = 0;
    if (is_device_ptr(offsets)) {
        HIP_TRY(hipMemcpyAsync(e->h_scalar + 2, offsets + n, sizeof(u64), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        nbytes = e->h_scalar[2];
        d_off = offsets;
    } else {
        nbytes = offsets[n];
        if ((rc = stage_input(e, kStageOffsets, offsets, (size_t)(n + 1) * sizeof(u64), &d_off, small))) return rc;
    }
    if (nbytes > 0 && !bytes) return fail(RAFTHIP_EINVAL, "bytes is NULL");
    const void *d_bytes = nullptr, *d_groups = nullptr;
    if ((rc = stage_input(e, kStageBytes, bytes, (size_t)nbytes, &d_bytes, small))) return rc;
    if ((rc = stage_input(e, kStageGroups, groups, (size_t)n * sizeof(u32), &d_groups, small))) return rc;
    Msg* d_msgs = nullptr;
    uint8_t* d_status = nullptr;
    bool msgs_host = out_msgs && !is_device_ptr(out_msgs);
    bool status_host = out_status && !is_device_ptr(out_status);
    // an output the kernels only write can live in the arena; the messages are read back by the passes that follow the
    // decode (Progress, votes), so those calls keep them in device memory
    auto small_output = [&](void* dst, size_t bytes) -> char* {
        char* d = nullptr;
        char* h = small ? small_take(e, bytes, &d) : nullptr;
        if (!h) return nullptr;
        after[n_after].dst = dst;
        after[n_after].src = h;
        after[n_after].bytes = bytes;
        ++n_after;
        e->stats.bytes_downloaded += bytes;
        return d;
    };
    if (msgs_host && !to_progress && !e->campaigning) {
        if (char* d = small_output(out_msgs, (size_t)n * sizeof(Msg))) {
            d_msgs = reinterpret_cast<Msg*>(d);
            msgs_host = false;
        }
    }
    if (d_msgs) {
    } else if (out_msgs && !msgs_host) {
        d_msgs = reinterpret_cast<Msg*>(out_msgs);
    } else if (out_msgs || to_progress || e->campaigning) {
        if ((rc = stage_reserve(e, kStageMsgs, (size_t)n * sizeof(Msg)))) return rc;
        d_msgs = static_cast<Msg*>(e->stage_buf[kStageMsgs]);
    }
    u64* d_aux = nullptr;
    if (to_progress) {
        if ((rc = stage_reserve(e, kStageAux, (size_t)n * sizeof(u64)))) return rc;
        d_aux = static_cast<u64*>(e->stage_buf[kStageAux]);
    }
    if (status_host) {
        if (char* d = small_output(out_status, (size_t)n)) {
            d_status = reinterpret_cast<uint8_t*>(d);
            status_host = false;
        }
    }
    if (out_status && !d_status) {
        if (status_host) {
            if ((rc = stage_reserve(e, kStageStatus, (size_t)n))) return rc;
            d_status = static_cast<uint8_t*>(e->stage_buf[kStageStatus]);
        } else {
            d_status = out_status;
        }
    }
    bool act_host = to_progress && (!actions || !is_device_ptr(actions));
    uint8_t* d_act = actions;
    u32* d_prev = nullptr;
    if (to_progress) {  // the decode kernel also links each event into its Progress's list (k_progress_link's work)
        if (act_host && actions) {
            if (char* d = small_output(actions, (size_t)n)) {
                d_act = reinterpret_cast<uint8_t*>(d);
                act_host = false;
            }
        }
        if (act_host) {
            if ((rc = stage_reserve(e, kStageActions, (size_t)n))) return rc;
            d_act = static_cast<uint8_t*>(e->stage_buf[kStageActions]);
        }
        if ((rc = stage_reserve(e, kStagePrev, (size_t)n * sizeof(u32)))) return rc;
        d_prev = static_cast<u32*>(e->stage_buf[kStagePrev]);
    }
    if (!e->wire_counts_zero)
        HIP_TRY(hipMemsetAsync(e->d_wire_counts, 0, RAFTHIP_WIRE_STATUSES * sizeof(u64), e->stream));
    e->wire_counts_zero = false;  // until this call's copy-out has zeroed them again
    const bool part = to_progress && prog_part_wanted(e, n);  // K9p: the decode kernel leaves the linking to the tiles
    if (to_progress && !part) {
        if ((rc = clear_stale_heads(e))) return rc;
        e->heads_dirty = true;  // the decode kernel links; cleared once the apply pass is on the stream
    }
    for (u64 off = 0; off < n; off += kMaxRecordsPerLaunch) {
        const u64 m = n - off < kMaxRecordsPerLaunch ? n - off : kMaxRecordsPerLaunch;
        const u32 blocks = (u32)((m + kWireBlock - 1) / kWireBlock);
        hipLaunchKernelGGL(k_ingest_wire, dim3(blocks), dim3(kWireBlock), 0, e->stream,
                           static_cast<const uint8_t*>(d_bytes), nbytes, static_cast<const u64*>(d_off) + off,
                           static_cast<const u32*>(d_groups) + off, m, e->G, e->S, static_cast<const u64*>(e->wire_hdr),
                           wire_hdr_words(e->S), e->match, e->stride, e->active, e->acks, e->mbytes,
                           d_msgs ? d_msgs + off : nullptr, d_status ? d_status + off : nullptr, e->d_wire_counts,
                           d_aux ? d_aux + off : nullptr, part ? nullptr : e->pp.rec, d_prev, d_act, (u32)off);
        HIP_TRY(hipGetLastError());
    }
    if (to_progress && (rc = progress_passes(e, d_msgs, d_aux, n, d_act, !part, part))) return rc;
    if (e->campaigning) {
        // RecordVote for the vote responses the decode let through (status VOTE; every other vote
        // response left the kernel without a slot), first vote per peer in stream order
        const u32 blocks = (u32)((n + kBlock - 1) / kBlock);
        u64* const no_count = nullptr;
        hipLaunchKernelGGL(k_record_votes<1>, dim3(blocks), dim3(kBlock), 0, e->stream, e->votes, e->vote_first,
                           e->stride, e->G, e->S, d_msgs, (u32)n, no_count);
        hipLaunchKernelGGL(k_record_votes<2>, dim3(blocks), dim3(kBlock), 0, e->stream, e->votes, e->vote_first,
                           e->stride, e->G, e->S, d_msgs, (u32)n, no_count);
        hipLaunchKernelGGL(k_record_votes<3>, dim3(blocks), dim3(kBlock), 0, e->stream, e->votes, e->vote_first,
                           e->stride, e->G, e->S, d_msgs, (u32)n, no_count);
        HIP_TRY(hipGetLastError());
    }
    bool wait = d_bytes != bytes || d_off != offsets || d_groups != groups || e->zero_copy_pending || n_after;  // caller may reuse its host buffers
    e->zero_copy_pending = false;
    // what is left in device memory comes back through the arena too when it fits: a copy into page-locked memory is
    // one asynchronous transfer, a copy into pageable memory is staged and waited for by the runtime piece by piece
    // (one kernel for both pieces, which also leaves the status counts at zero for the next call: a launch costs a
    // third of what a copy command does, and the memset in front of the next call goes as well)
    CopyOut out2{};
    auto download = [&](void* dst, const void* src, size_t bytes, bool zero_src) -> hipError_t {
        char* d = nullptr;
        char* h = small ? small_take(e, bytes, &d) : nullptr;
        wait = true;
        if (h && n_after < (int)(sizeof(after) / sizeof(after[0]))) {
            after[n_after].dst = dst;
            after[n_after].src = h;
            after[n_after].bytes = bytes;
            ++n_after;
            if (bytes % 8 == 0 && out2.n < 2) {
                out2.src[out2.n] = static_cast<u64*>(const_cast<void*>(src));
                out2.dst[out2.n] = reinterpret_cast<u64*>(d);
                out2.words[out2.n] = (u32)(bytes / 8);
                out2.zero[out2.n] = zero_src;
                ++out2.n;
                return hipSuccess;
            }
            dst = h;
        }
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream);
    };
    if (msgs_host) {
        HIP_TRY(download(out_msgs, d_msgs, (size_t)n * sizeof(Msg), false));
        e->stats.bytes_downloaded += n * sizeof(Msg);
    }
    if (status_host) {
        HIP_TRY(hipMemcpyAsync(out_status, d_status, (size_t)n, hipMemcpyDeviceToHost, e->stream));
        e->stats.bytes_downloaded += n;
        wait = true;
    }
    if (status_counts) HIP_TRY(download(status_counts, e->d_wire_counts, RAFTHIP_WIRE_STATUSES * sizeof(u64), true));
    if (to_progress && actions && act_host) {
        HIP_TRY(hipMemcpyAsync(actions, d_act, (size_t)n, hipMemcpyDeviceToHost, e->stream));
        e->stats.bytes_downloaded += n;
        wait = true;
    }
    if (out2.n) {
        const u32 words = out2.words[0] > out2.words[1] ? out2.words[0] : out2.words[1];
        hipLaunchKernelGGL(k_copy_out, dim3((words + kBlock - 1) / kBlock), dim3(kBlock), 0, e->stream, out2);
        HIP_TRY(hipGetLastError());
        e->wire_counts_zero = (out2.zero[0] && out2.n > 0) || (out2.zero[1] && out2.n > 1);
    }
    if (wait) HIP_TRY(hipStreamSynchronize(e->stream));
    for (int i = 0; i < n_after; ++i) memcpy(after[i].dst, after[i].src, after[i].bytes);
    e->stats.msgs_ingested += n;
    return RAFTHIP_OK;
}

int rafthip_ingest_wire(rafthip_engine* e, const uint8_t* bytes, const uint64_t* offsets, const uint32_t* groups,
                        uint64_t n, rafthip_msg* out_msgs, uint8_t* out_status, uint64_t* status_counts) {
    return wire_run(e, bytes, offsets, groups, n, out_msgs, out_status, status_counts, false, nullptr);
}

int rafthip_step_wire(rafthip_engine* e, const uint8_t* bytes, const uint64_t* offsets, const uint32_t* groups,
                      uint64_t n, rafthip_msg* out_msgs, uint8_t* out_status, uint64_t* status_counts,
                      uint8_t* actions) {
    return wire_run(e, bytes, offsets, groups, n, out_msgs, out_status, status_counts, true, actions);
}

// ---- per-peer replication state (tracker.Progress, tracker.Inflights) --------

static_assert(kStepLimit == RAFTHIP_STEP_LIMIT && kActSendAppend == RAFTHIP_ACT_SEND_APPEND &&
                  kActUpdated == RAFTHIP_ACT_UPDATED && kActWasPaused == RAFTHIP_ACT_WAS_PAUSED &&
                  kActCaughtUp == RAFTHIP_ACT_CAUGHT_UP && kActError == RAFTHIP_ACT_ERROR &&
                  kActDeferred == RAFTHIP_ACT_DEFERRED && kActDropped == RAFTHIP_ACT_DROPPED &&
                  kEvUnreachable == RAFTHIP_MSG_UNREACHABLE && kEvSnapStatus == RAFTHIP_MSG_SNAP_STATUS &&
                  kEvSentApp == RAFTHIP_MSG_SENT_APP && kEvSentSnap == RAFTHIP_MSG_SENT_SNAP &&
                  kStateProbe == RAFTHIP_STATE_PROBE && kStateReplicate == RAFTHIP_STATE_REPLICATE &&
                  kStateSnapshot == RAFTHIP_STATE_SNAPSHOT && kEvNeedsHostRec == RAFTHIP_MSG_NEEDS_HOST,
              "progress constants");

static int need_progress(const rafthip_engine* e) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    if (!e->prog) return fail(RAFTHIP_EINVAL, "rafthip_progress_config must come first");
    return RAFTHIP_OK;
}

int rafthip_progress_config(rafthip_engine* e, uint32_t max_inflight, uint64_t max_inflight_bytes) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    if (max_inflight > kMaxInflight) return fail(RAFTHIP_EINVAL, "max_inflight %u > %u", max_inflight, kMaxInflight);
    if (e->prog) {
        if (e->pp.cap == max_inflight && e->pp.max_bytes == max_inflight_bytes) return RAFTHIP_OK;
        return fail(RAFTHIP_EINVAL, "the Inflights limits of an engine are fixed once configured");
    }
    DeviceGuard guard(e->device);
    const size_t cells = (size_t)e->S * e->stride;
    const size_t ring = (size_t)(max_inflight ? max_inflight : 1) * cells * sizeof(RingEntry);
    int rc;
    if ((rc = ensure(e, e->pp.rec, cells * sizeof(ProgRec)))) return rc;
    if ((rc = ensure(e, e->pp.ring, ring))) return rc;
    if ((rc = ensure(e, e->first_index, (size_t)e->stride * sizeof(u64)))) return rc;
    if ((rc = ensure(e, e->term_last, (size_t)e->stride * sizeof(u64)))) return rc;
    if ((rc = ensure(e, e->term_first, (size_t)e->stride * sizeof(u64)))) return rc;
    e->pp.match = e->match;
    e->pp.stride = e->stride;
    e->pp.cap = max_inflight;
    e->pp.ring_len = max_inflight ? max_inflight : 1;
    e->pp.max_bytes = max_inflight_bytes;
    e->prog = true;
    return RAFTHIP_OK;
}

namespace {

// Transient [S][count] planes on the device for the SoA <-> record conversions.
struct TempPlanes {
    char* base = nullptr;
    ~TempPlanes() {
        if (base) (void)hipFree(base);
    }
    int alloc(size_t bytes) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&base), bytes));
        return RAFTHIP_OK;
    }
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Groups per round of rafthip_upload_inflights / rafthip_read_inflights: the rings pass through two transient planes
// of `rows` x groups words on the device, kept below kRingTempBytes together (a window of 256 entries over 9 slots is
// 36 KB per group).
constexpr size_t kRingTempBytes = 512u << 20;
u64 ring_temp_groups(u64 rows, u64 count) {
    const u64 fit = kRingTempBytes / (rows * 16);
    return fit < 1 ? 1 : fit < count ? fit : count;
}

}  // namespace

int rafthip_upload_progress(rafthip_engine* e, uint64_t first, uint64_t count, const uint8_t* state,
                            const uint64_t* next, const uint64_t* pending, uint64_t src_stride) {
    int rc = need_progress(e);
    if (rc || (rc = check_range(e, first, count))) return rc;
    if (!state || !next) return fail(RAFTHIP_EINVAL, "state/next is NULL");
    if (src_stride < count) return fail(RAFTHIP_EINVAL, "src_stride %llu < count %llu", (u64)src_stride, (u64)count);
    if (count == 0) return RAFTHIP_OK;
    DeviceGuard guard(e->device);
    const size_t cells = (size_t)e->S * count;
    const size_t o_next = align256(cells), o_pend = o_next + cells * 8;
    TempPlanes t;
    if ((rc = t.alloc(o_pend + cells * 8))) return rc;
    uint8_t* t_state = reinterpret_cast<uint8_t*>(t.base);
    u64* t_next = reinterpret_cast<u64*>(t.base + o_next);
    u64* t_pend = reinterpret_cast<u64*>(t.base + o_pend);
    if ((rc = copy_rows(e, t_state, count, state, src_stride, count, e->S))) return rc;
    if ((rc = copy_rows(e, t_next, count * 8, next, src_stride * 8, count * 8, e->S))) return rc;
    if (pending && (rc = copy_rows(e, t_pend, count * 8, pending, src_stride * 8, count * 8, e->S))) return rc;
    hipLaunchKernelGGL(k_progress_pack, dim3((u32)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream,
                       e->pp.rec, e->stride, (u64)first, (u64)count, e->S, static_cast<const uint8_t*>(t_state),
                       static_cast<const u64*>(t_next), pending ? static_cast<const u64*>(t_pend) : nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_uploaded += cells * (pending ? 17 : 9);
    return RAFTHIP_OK;
}

int rafthip_init_progress_row(rafthip_engine* e, uint64_t group, uint32_t slot, uint8_t state, uint64_t next,
                              uint64_t pending) {
    int rc = need_progress(e);
    if (rc || (rc = check_range(e, group, 1))) return rc;
    if (slot >= e->S) return fail(RAFTHIP_EINVAL, "slot %u outside [0, %u)", slot, e->S);
    if ((state & 3u) > RAFTHIP_STATE_SNAPSHOT) return fail(RAFTHIP_EINVAL, "state %u is not a tracker.StateType", state & 3u);
    DeviceGuard guard(e->device);
    hipLaunchKernelGGL(k_progress_init_row, dim3(1), dim3(64), 0, e->stream, e->pp.rec, e->stride, (u64)group, slot,
                       (u32)state, (u64)next, (u64)pending);
    HIP_TRY(hipGetLastError());
    return RAFTHIP_OK;
}

int rafthip_read_progress(rafthip_engine* e, uint64_t first, uint64_t count, uint8_t* state, uint64_t* next,
                          uint64_t* pending, uint32_t* infl_meta, uint64_t* infl_bytes, uint64_t dst_stride) {
    int rc = need_progress(e);
    if (rc || (rc = check_range(e, first, count))) return rc;
    if (dst_stride < count) return fail(RAFTHIP_EINVAL, "dst_stride %llu < count %llu", (u64)dst_stride, (u64)count);
    if (count == 0) return RAFTHIP_OK;
    DeviceGuard guard(e->device);
    const size_t cells = (size_t)e->S * count;
    const size_t o_next = align256(cells), o_pend = o_next + cells * 8, o_bytes = o_pend + cells * 8,
                 o_meta = o_bytes + cells * 8;
    TempPlanes t;
    if ((rc = t.alloc(o_meta + cells * 4))) return rc;
    uint8_t* t_state = reinterpret_cast<uint8_t*>(t.base);
    u64* t_next = reinterpret_cast<u64*>(t.base + o_next);
    u64* t_pend = reinterpret_cast<u64*>(t.base + o_pend);
    u64* t_bytes = reinterpret_cast<u64*>(t.base + o_bytes);
    u32* t_meta = reinterpret_cast<u32*>(t.base + o_meta);
    hipLaunchKernelGGL(k_progress_unpack, dim3((u32)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream,
                       static_cast<const ProgRec*>(e->pp.rec), e->stride, (u64)first, (u64)count, e->S, t_state,
                       t_next, t_pend, t_meta, t_bytes);
    HIP_TRY(hipGetLastError());
    if (state && (rc = copy_rows(e, state, dst_stride, t_state, count, count, e->S))) return rc;
    if (next && (rc = copy_rows(e, next, dst_stride * 8, t_next, count * 8, count * 8, e->S))) return rc;
    if (pending && (rc = copy_rows(e, pending, dst_stride * 8, t_pend, count * 8, count * 8, e->S))) return rc;
    if (infl_meta && (rc = copy_rows(e, infl_meta, dst_stride * 4, t_meta, count * 4, count * 4, e->S))) return rc;
    if (infl_bytes && (rc = copy_rows(e, infl_bytes, dst_stride * 8, t_bytes, count * 8, count * 8, e->S))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_downloaded += cells * ((state ? 1 : 0) + (next ? 8 : 0) + (pending ? 8 : 0) +
                                          (infl_meta ? 4 : 0) + (infl_bytes ? 8 : 0));
    return RAFTHIP_OK;
}

int rafthip_upload_inflights(rafthip_engine* e, uint64_t first, uint64_t count, const uint32_t* infl_meta,
                             const uint64_t* infl_bytes, const uint64_t* ring_index, const uint64_t* ring_bytes,
                             uint64_t src_stride) {
    int rc = need_progress(e);
    if (rc || (rc = check_range(e, first, count))) return rc;
    if (!infl_meta || !infl_bytes || !ring_index || !ring_bytes) return fail(RAFTHIP_EINVAL, "a source is NULL");
    if (src_stride < count) return fail(RAFTHIP_EINVAL, "src_stride %llu < count %llu", (u64)src_stride, (u64)count);
    if (count == 0) return RAFTHIP_OK;
    DeviceGuard guard(e->device);
    const u64 st = e->stride, rows = (u64)(e->pp.cap ? e->pp.cap : 1) * e->S;
    const size_t cells = (size_t)e->S * count;
    TempPlanes t;
    if ((rc = t.alloc(cells * 12))) return rc;
    u64* t_bytes = reinterpret_cast<u64*>(t.base);
    u32* t_meta = reinterpret_cast<u32*>(t.base + cells * 8);
    if ((rc = copy_rows(e, t_meta, count * 4, infl_meta, src_stride * 4, count * 4, e->S))) return rc;
    if ((rc = copy_rows(e, t_bytes, count * 8, infl_bytes, src_stride * 8, count * 8, e->S))) return rc;
    hipLaunchKernelGGL(k_progress_set_inflights, dim3((u32)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       e->stream, e->pp.rec, st, (u64)first, (u64)count, e->S, static_cast<const u32*>(t_meta),
                       static_cast<const u64*>(t_bytes));
    HIP_TRY(hipGetLastError());
    // the rings: the ABI's [entry][slot][group] planes -> one contiguous ring per Progress, through transient planes
    // on the device, a range of groups at a time so that the transient stays below kRingTempBytes
    const u64 step = ring_temp_groups(rows, count);
    TempPlanes tr;
    const size_t plane_b = (size_t)rows * step * 8;
    if ((rc = tr.alloc(2 * plane_b))) return rc;
    u64* t_ri = reinterpret_cast<u64*>(tr.base);
    u64* t_rb = reinterpret_cast<u64*>(tr.base + plane_b);
    for (u64 g0 = 0; g0 < count; g0 += step) {
        const u64 m = count - g0 < step ? count - g0 : step;
        if ((rc = copy_rows(e, t_ri, m * 8, ring_index + g0, src_stride * 8, m * 8, rows))) return rc;
        if ((rc = copy_rows(e, t_rb, m * 8, ring_bytes + g0, src_stride * 8, m * 8, rows))) return rc;
        hipLaunchKernelGGL(k_ring_pack, dim3((u32)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream, e->pp.ring, st,
                           (u64)(first + g0), m, e->S, e->pp.ring_len, static_cast<const u64*>(t_ri),
                           static_cast<const u64*>(t_rb), m);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_uploaded += count * (e->S * 12 + rows * 16);
    return RAFTHIP_OK;
}

int rafthip_read_inflights(rafthip_engine* e, uint64_t first, uint64_t count, uint64_t* ring_index,
                           uint64_t* ring_bytes, uint64_t dst_stride) {
    int rc = need_progress(e);
    if (rc || (rc = check_range(e, first, count))) return rc;
    if (dst_stride < count) return fail(RAFTHIP_EINVAL, "dst_stride %llu < count %llu", (u64)dst_stride, (u64)count);
    DeviceGuard guard(e->device);
    if (count == 0 || (!ring_index && !ring_bytes)) return RAFTHIP_OK;
    const u64 st = e->stride, rows = (u64)e->pp.ring_len * e->S;
    const u64 step = ring_temp_groups(rows, count);
    TempPlanes tr;
    const size_t plane_b = (size_t)rows * step * 8;
    if ((rc = tr.alloc(2 * plane_b))) return rc;
    u64* t_ri = ring_index ? reinterpret_cast<u64*>(tr.base) : nullptr;
    u64* t_rb = ring_bytes ? reinterpret_cast<u64*>(tr.base + plane_b) : nullptr;
    for (u64 g0 = 0; g0 < count; g0 += step) {
        const u64 m = count - g0 < step ? count - g0 : step;
        hipLaunchKernelGGL(k_ring_unpack, dim3((u32)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream,
                           static_cast<const RingEntry*>(e->pp.ring), st, (u64)(first + g0), m, e->S, e->pp.ring_len, t_ri,
                           t_rb, m);
        HIP_TRY(hipGetLastError());
        if (ring_index && (rc = copy_rows(e, ring_index + g0, dst_stride * 8, t_ri, m * 8, m * 8, rows))) return rc;
        if (ring_bytes && (rc = copy_rows(e, ring_bytes + g0, dst_stride * 8, t_rb, m * 8, m * 8, rows))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_downloaded += count * rows * ((ring_index ? 8 : 0) + (ring_bytes ? 8 : 0));
    return RAFTHIP_OK;
}

int rafthip_upload_first_index(rafthip_engine* e, uint64_t first, uint64_t count, const uint64_t* first_index) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!first_index) return fail(RAFTHIP_EINVAL, "first_index is NULL");
    DeviceGuard guard(e->device);
    if ((rc = ensure(e, e->first_index, (size_t)e->stride * sizeof(u64)))) return rc;
    HIP_TRY(hipMemcpyAsync(e->first_index + first, first_index, count * 8, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_uploaded += count * 8;
    return RAFTHIP_OK;
}

// Tile of K9p for this engine: the largest power of two of groups (at most 1024) whose S cell counters fit 64 KB of LDS.
static u32 prog_part_tile_shift(const rafthip_engine* e) {
    u32 shift = 10;  // 256- and 512-group tiles measured: the same within 2 % (1460 / 1429 / 1466 us per 2^24 events)
    while (((size_t)e->S << shift) * sizeof(u32) > 64u * 1024u) --shift;
    return shift;
}

// Whether a batch of n events takes the partitioned stepper (progress_part.h).  Measured (tools/step_sweep.py,
// profiles/r03_c_step_sweep.txt): its passes have a floor of 45-70 us, and K9's random lines are cheap while the
// Progress state (40 B per cell) stays in the 256 MiB Infinity Cache — at 2^18 groups x 9 K9 is never slower, at 2^20
// and 2^22 groups x 9 K9p wins from 2^22 events on (1.15x ... 1.55x at 2^24).
static bool prog_part_wanted(const rafthip_engine* e, u64 n) {
    if (e->step_mode == 1 || getenv("RAFTHIP_STEP_LINKED")) return false;
    const u32 tiles = (u32)(e->stride >> prog_part_tile_shift(e));
    if (round_up(tiles, 1u << kPartLB) > kPartMaxTiles || n > kMaxRecordsPerLaunch) return false;
    if (e->step_mode == 2) return true;
    return n >= (1ull << 22) && (u64)e->S * e->stride * (sizeof(ProgRec) + sizeof(u64)) > (256ull << 20);
}

// K9p: count, scan, two partition passes, per-tile apply; K9's link and apply kernels behind the device-side
// skew flag.  The caller has reserved kStagePrev for n words.
static int progress_partitioned(rafthip_engine* e, const Msg* d_msgs, const u64* d_aux, u64 n, uint8_t* d_act, u32* prev) {
    const u32 shift = prog_part_tile_shift(e);
    const u32 tiles = (u32)(e->stride >> shift);
    const u32 tiles_pad = (u32)round_up(tiles, 1u << kPartLB), nb = tiles_pad >> kPartLB;
    int rc;
    PartBufs pb;
    if ((rc = part_scratch(e, n, &pb))) return rc;
    const size_t lds_count = (size_t)tiles_pad * sizeof(u32), lds_stage2 = (size_t)kPartChunk * sizeof(Ack),
                 lds_stage1 = lds_stage2 + kPartChunk, lds_cells = ((size_t)e->S << shift) * sizeof(u32);
    if (!e->prog_part_attr) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_prog_part_count), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(kPartMaxTiles * sizeof(u32))));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_scan), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(kPartMaxTiles * sizeof(u32))));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_partition<1, kPartEvents>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stage1));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_part_partition<2, kPartEvents>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stage2));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_prog_part_apply), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    64 * 1024));
        e->prog_part_attr = true;
    }
    if ((rc = clear_stale_heads(e))) return rc;  // the fallback links through the records' head words
    hipStream_t st = e->stream;
    const u32 count_wgs = 512;
    const u64 slice = (n + count_wgs - 1) / count_wgs;
    hipLaunchKernelGGL(k_prog_part_count, dim3(count_wgs), dim3(kPartCountBlock), lds_count, st, d_msgs, n, slice, e->G, e->S, shift,
                       tiles_pad, pb.tile_count, e->term_last, d_act);
    hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(1024), lds_count, st, pb, tiles_pad, nb);
    const u64 chunks1 = (n + kPartChunk - 1) / kPartChunk, chunks2 = chunks1 + nb;
    const u32 persistent = 256;
    hipLaunchKernelGGL((k_part_partition<1, kPartEvents>), dim3((u32)(chunks1 < persistent ? chunks1 : persistent)), dim3(kPartBlock),
                       lds_stage1, st, reinterpret_cast<const Ack*>(d_msgs), n, e->G, e->S, shift, nb, pb, e->part_a);
    hipLaunchKernelGGL((k_part_partition<2, kPartEvents>), dim3((u32)(chunks2 < persistent ? chunks2 : persistent)), dim3(kPartBlock),
                       lds_stage2, st, static_cast<const Ack*>(e->part_a), n, e->G, e->S, shift, nb, pb, e->part_b);
    e->pp.match = e->match;
    hipLaunchKernelGGL(k_prog_part_apply, dim3(tiles), dim3(kProgPartBlock), lds_cells, st, static_cast<const Ack*>(e->part_b), e->part_a,
                       d_aux, e->S, shift, e->pp, pb, e->active, e->acks, e->mbytes, static_cast<const u64*>(e->term_last),
                       static_cast<const u64*>(e->first_index), d_act);
    HIP_TRY(hipGetLastError());
    // a batch the scan found piled onto one tile: K9's two passes over the original records (decided on the device;
    // the usual case is two grids that leave at once)
    const u64 want = (n + kBlock - 1) / kBlock;
    const u32 blocks = (u32)(want < kProgFlaggedGrid ? want : kProgFlaggedGrid);
    e->heads_dirty = true;
    hipLaunchKernelGGL(k_progress_link_flagged, dim3(blocks), dim3(kBlock), 0, st, d_msgs, (u32)n, e->G, e->S, e->stride, e->pp.rec,
                       prev, e->term_last, d_act, static_cast<const u32*>(pb.flag));
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_progress_apply_flagged, dim3(blocks), dim3(kBlock), 0, st, d_msgs, d_aux, (u32)n, e->G, e->S, e->pp,
                       static_cast<const u32*>(prev), e->active, e->acks, e->mbytes, static_cast<const u64*>(e->term_last),
                       static_cast<const u64*>(e->first_index), d_act, static_cast<const u32*>(pb.flag));
    HIP_TRY(hipGetLastError());
    e->heads_dirty = false;
    part_done(e);
    return RAFTHIP_OK;
}

// The Progress stepper over n device-resident records (d_act: device, n bytes).  `linked`: the decode kernel has
// already linked the events into K9's lists; `part`: take K9p (never together with `linked`).
static int progress_passes(rafthip_engine* e, const Msg* d_msgs, const u64* d_aux, uint64_t n, uint8_t* d_act,
                           bool linked, bool part) {
    int rc;
    const size_t mb = (size_t)e->stride * e->mbytes;
    if ((rc = ensure(e, e->active, mb))) return rc;
    if ((rc = ensure(e, e->acks, mb))) return rc;
    if ((rc = stage_reserve(e, kStagePrev, (size_t)n * sizeof(u32)))) return rc;
    u32* prev = static_cast<u32*>(e->stage_buf[kStagePrev]);
    if (part && !linked) return progress_partitioned(e, d_msgs, d_aux, n, d_act, prev);
    const u32 blocks = (u32)((n + kBlock - 1) / kBlock);
    if (!linked) {
        int rc2 = clear_stale_heads(e);
        if (rc2) return rc2;
        e->heads_dirty = true;  // until the apply pass is on the stream
        hipLaunchKernelGGL(k_progress_link, dim3(blocks), dim3(kBlock), 0, e->stream, d_msgs, (u32)n, e->G, e->S,
                           e->stride, e->pp.rec, prev, e->term_last, d_act);
        HIP_TRY(hipGetLastError());
    }
    e->pp.match = e->match;
    hipLaunchKernelGGL(k_progress_apply, dim3(blocks), dim3(kBlock), 0, e->stream, d_msgs, d_aux, (u32)n, e->G, e->S,
                       e->pp, static_cast<const u32*>(prev), e->active, e->acks, e->mbytes,
                       static_cast<const u64*>(e->term_last), static_cast<const u64*>(e->first_index), d_act);
    HIP_TRY(hipGetLastError());
    e->heads_dirty = false;
    return RAFTHIP_OK;
}

int rafthip_step_dense(rafthip_engine* e, uint64_t first, uint64_t count, const uint64_t* acks, uint64_t ack_stride,
                       const void* hb_mask, const void* hb_ctx_mask, uint8_t* actions, uint64_t act_stride) {
    int rc = need_progress(e);
    if (rc || (rc = check_range(e, first, count))) return rc;
    if (acks && ack_stride < count)
        return fail(RAFTHIP_EINVAL, "ack_stride %llu < count %llu", (u64)ack_stride, (u64)count);
    if (actions && act_stride < count)
        return fail(RAFTHIP_EINVAL, "act_stride %llu < count %llu", (u64)act_stride, (u64)count);
    if (hb_ctx_mask && !hb_mask) return fail(RAFTHIP_EINVAL, "hb_ctx_mask without hb_mask");
    if (count == 0 || (!acks && !hb_mask)) return RAFTHIP_OK;
    DeviceGuard guard(e->device);
    const size_t mb = (size_t)e->mbytes, plane_b = (size_t)e->stride * mb;
    if ((rc = ensure(e, e->active, plane_b))) return rc;
    if ((rc = ensure(e, e->acks, plane_b))) return rc;
    // host inputs are staged; device (or page-locked) ones are read in place
    const void *d_acks = nullptr, *d_hb = nullptr, *d_ctx = nullptr;
    u64 d_ack_stride = ack_stride;
    if (acks) {
        if (is_device_ptr(acks)) {
            d_acks = acks;
        } else {
            d_ack_stride = count;
            if ((rc = stage_reserve(e, kStageAux, (size_t)count * e->S * sizeof(u64)))) return rc;
            if ((rc = copy_rows(e, e->stage_buf[kStageAux], count * 8, acks, ack_stride * 8, count * 8, e->S))) return rc;
            d_acks = e->stage_buf[kStageAux];
            e->stats.bytes_uploaded += count * 8 * e->S;
        }
    }
    if (hb_mask && (rc = stage_input(e, kStageOffsets, hb_mask, (size_t)count * mb, &d_hb))) return rc;
    if (hb_ctx_mask && (rc = stage_input(e, kStageGroups, hb_ctx_mask, (size_t)count * mb, &d_ctx))) return rc;
    const bool act_host = actions && !is_device_ptr(actions);
    uint8_t* d_act = actions;
    u64 d_act_stride = act_stride;
    if (act_host) {
        d_act_stride = count;
        if ((rc = stage_reserve(e, kStageActions, (size_t)count * e->S))) return rc;
        d_act = static_cast<uint8_t*>(e->stage_buf[kStageActions]);
    }
    e->pp.match = e->match;
    const u32 blocks = (u32)((count + kBlock - 1) / kBlock);
#define RAFTHIP_DENSE(MT)                                                                                          \
    hipLaunchKernelGGL((k_progress_dense<MT>), dim3(blocks), dim3(kBlock), 0, e->stream, e->pp, (u64)first,         \
                       (u64)count, e->S, static_cast<const u64*>(d_acks), d_ack_stride,                              \
                       static_cast<const MT*>(d_hb), static_cast<const MT*>(d_ctx), static_cast<MT*>(e->active),     \
                       static_cast<MT*>(e->acks), static_cast<const u64*>(e->term_last),                             \
                       static_cast<const u64*>(e->first_index), d_act, d_act_stride)
    if (e->mbytes == 1) {
        RAFTHIP_DENSE(uint8_t);
    } else if (e->mbytes == 2) {
        RAFTHIP_DENSE(uint16_t);
    } else if (e->mbytes == 4) {
        RAFTHIP_DENSE(uint32_t);
    } else {
        RAFTHIP_DENSE(uint64_t);
    }
#undef RAFTHIP_DENSE
    HIP_TRY(hipGetLastError());
    bool wait = d_acks != acks || (hb_mask && d_hb != hb_mask) || (hb_ctx_mask && d_ctx != hb_ctx_mask) ||
                e->zero_copy_pending;
    e->zero_copy_pending = false;
    if (act_host) {
        if ((rc = copy_rows(e, actions, act_stride, d_act, count, count, e->S))) return rc;
        e->stats.bytes_downloaded += count * e->S;
        wait = true;
    }
    if (wait) HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.msgs_ingested += count * e->S;
    return RAFTHIP_OK;
}

static_assert(kSendNone == RAFTHIP_SEND_NONE && kSendAppend == RAFTHIP_SEND_APPEND && kSendEmpty == RAFTHIP_SEND_EMPTY &&
                  kSendSnapshot == RAFTHIP_SEND_SNAPSHOT,
              "send plan constants");

int rafthip_plan_appends(rafthip_engine* e, uint64_t first, uint64_t count, int send_if_empty, uint8_t* plan,
                         uint64_t* next, uint64_t stride) {
    int rc = need_progress(e);
    if (rc || (rc = check_range(e, first, count))) return rc;
    if (!plan) return fail(RAFTHIP_EINVAL, "plan is NULL");
    if (stride < count) return fail(RAFTHIP_EINVAL, "stride %llu < count %llu", (u64)stride, (u64)count);
    if (count == 0) return RAFTHIP_OK;
    DeviceGuard guard(e->device);
    const size_t plane_b = (size_t)e->stride * e->mbytes;
    if ((rc = ensure(e, e->active, plane_b))) return rc;
    if ((rc = ensure(e, e->self_mask, plane_b))) return rc;
    const bool every_slot = e->uniform && e->uniform_voters > 0;
    if (!every_slot && (rc = ensure_masks(e))) return rc;
    const bool plan_host = !is_device_ptr(plan), next_host = next && !is_device_ptr(next);
    uint8_t* d_plan = plan;
    u64* d_next = reinterpret_cast<u64*>(next);
    u64 d_stride = stride;
    if (plan_host || next_host) {  // staged outputs are packed at row pitch `count`
        if (!plan_host || (next && !next_host))
            return fail(RAFTHIP_EINVAL, "plan and next must both be host or both be device memory");
        d_stride = count;
        if ((rc = stage_reserve(e, kStageActions, (size_t)count * e->S))) return rc;
        d_plan = static_cast<uint8_t*>(e->stage_buf[kStageActions]);
        if (next) {
            if ((rc = stage_reserve(e, kStageAux, (size_t)count * e->S * sizeof(u64)))) return rc;
            d_next = static_cast<u64*>(e->stage_buf[kStageAux]);
        }
    }
    e->pp.match = e->match;
    const u32 blocks = (u32)((count + kBlock - 1) / kBlock);
#define RAFTHIP_PLAN(MT)                                                                                            \
    hipLaunchKernelGGL((k_plan_appends<MT>), dim3(blocks), dim3(kBlock), 0, e->stream, e->pp, (u64)first, (u64)count, \
                       e->S, every_slot ? nullptr : static_cast<const MT*>(e->in_mask),                              \
                       static_cast<const MT*>(e->out_mask), static_cast<const MT*>(e->learner_mask),                  \
                       static_cast<const MT*>(e->self_mask), static_cast<const MT*>(e->active),                       \
                       static_cast<const u64*>(e->first_index), static_cast<const u64*>(e->term_last),                \
                       send_if_empty != 0, d_plan, d_next, d_stride)
    if (e->mbytes == 1) {
        RAFTHIP_PLAN(uint8_t);
    } else if (e->mbytes == 2) {
        RAFTHIP_PLAN(uint16_t);
    } else if (e->mbytes == 4) {
        RAFTHIP_PLAN(uint32_t);
    } else {
        RAFTHIP_PLAN(uint64_t);
    }
#undef RAFTHIP_PLAN
    HIP_TRY(hipGetLastError());
    e->stats.groups_evaluated += count;
    if (plan_host) {
        if ((rc = copy_rows(e, plan, stride, d_plan, count, count, e->S))) return rc;
        e->stats.bytes_downloaded += count * e->S;
        if (next) {
            if ((rc = copy_rows(e, next, stride * 8, d_next, count * 8, count * 8, e->S))) return rc;
            e->stats.bytes_downloaded += count * 8 * e->S;
        }
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return RAFTHIP_OK;
}

int rafthip_step_progress(rafthip_engine* e, const rafthip_msg* msgs, const uint64_t* aux, uint64_t n,
                          uint8_t* actions) {
    int rc = need_progress(e);
    if (rc) return rc;
    if (n == 0) return RAFTHIP_OK;
    if (!msgs) return fail(RAFTHIP_EINVAL, "msgs is NULL");
    if (n > (1ull << 31)) return fail(RAFTHIP_EINVAL, "more than 2^31 events in one batch: split it");
    DeviceGuard guard(e->device);
    const size_t mb = (size_t)e->stride * e->mbytes;
    if ((rc = ensure(e, e->active, mb))) return rc;
    if ((rc = ensure(e, e->acks, mb))) return rc;
    const void* d_recs = nullptr;
    bool staged = false;
    e->small_used = 0;
    if ((rc = stage_records(e, msgs, n, &d_recs, &staged))) return rc;
    const void* d_aux = nullptr;
    const bool small = staged && !is_device_ptr(d_recs) && n <= kSmallRecords;  // stage_records took the arena
    if (aux && (rc = stage_input(e, kStageAux, aux, (size_t)n * sizeof(u64), &d_aux, small))) return rc;
    bool act_host = !actions || !is_device_ptr(actions);
    uint8_t* d_act = actions;
    const char* h_act = nullptr;
    if (act_host && actions && small) {
        char* d = nullptr;
        if ((h_act = small_take(e, (size_t)n, &d))) {
            d_act = reinterpret_cast<uint8_t*>(d);
            act_host = false;
        }
    }
    if (act_host) {
        if ((rc = stage_reserve(e, kStageActions, (size_t)n))) return rc;
        d_act = static_cast<uint8_t*>(e->stage_buf[kStageActions]);
    }
    if ((rc = progress_passes(e, static_cast<const Msg*>(d_recs), static_cast<const u64*>(d_aux), n, d_act, false,
                              is_device_ptr(d_recs) && prog_part_wanted(e, n))))
        return rc;
    bool wait = staged || (aux && d_aux != aux) || e->zero_copy_pending;
    e->zero_copy_pending = false;
    if (actions && act_host) {
        HIP_TRY(hipMemcpyAsync(actions, d_act, (size_t)n, hipMemcpyDeviceToHost, e->stream));
        e->stats.bytes_downloaded += n;
        wait = true;
    }
    if (wait) HIP_TRY(hipStreamSynchronize(e->stream));
    if (h_act) {
        memcpy(actions, h_act, (size_t)n);
        e->stats.bytes_downloaded += n;
    }
    e->stats.msgs_ingested += n;
    return RAFTHIP_OK;
}

int rafthip_commit(rafthip_engine* e) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    DeviceGuard guard(e->device);
    u32 bg = 0;
    return launch_commit<kStore>(e, &bg);
}

}  // extern "C"

namespace {
// How one launch can serve a set of engines (the shards of one GPU):
//   kManyUniform  every engine has the same uniform config of 1..8 voters: K1m
//   kManyMasked   every engine has the same S <= 8 and takes the lane-per-group kernels; some carry per-group
//                 masks (a conf change in flight), the others a uniform config: K2m
//   kManySingle   anything else: engine by engine
enum ManyKind { kManySingle = 0, kManyUniform, kManyMasked };
ManyKind many_kind(rafthip_engine* const* engines, u32 n) {
    const rafthip_engine* e0 = engines[0];
    bool uni = e0->uniform_voters >= 1 && e0->uniform_voters <= 8, masked = e0->S <= 8;
    for (u32 i = 0; i < n; ++i) {
        const rafthip_engine* e = engines[i];
        if (e->device != e0->device) return kManySingle;
        if (!e->uniform || e->uniform_voters != e0->uniform_voters) uni = false;
        if (e->S != e0->S) masked = false;
        if (!e->uniform && !e->in_mask) masked = false;
    }
    return uni ? kManyUniform : masked ? kManyMasked : kManySingle;
}
void fill_many_masks(rafthip_engine* const* engines, u32 n, ManyMasks* mk) {
    for (u32 k = 0; k < n; ++k) {
        const rafthip_engine* e = engines[k];
        mk->in_mask[k] = e->uniform ? nullptr : static_cast<const uint8_t*>(e->in_mask);
        mk->out_mask[k] = e->uniform ? nullptr : static_cast<const uint8_t*>(e->out_mask);
        mk->uniform_voters[k] = (uint8_t)(e->uniform ? e->uniform_voters : 0);
    }
}
}  // namespace

extern "C" {

int rafthip_commit_many(rafthip_engine* const* engines, uint32_t n) {
    if (!engines || n == 0) return fail(RAFTHIP_EINVAL, "engines is NULL or n is 0");
    for (u32 i = 0; i < n; ++i)
        if (!engines[i]) return fail(RAFTHIP_EINVAL, "engines[%u] is NULL", i);
    rafthip_engine* e0 = engines[0];
    // One launch needs: one GPU and either the K1 shape (a uniform config of 1..8 voters, the same in every
    // shard) or the K2 shape (the same S <= 8 in every shard, masks or not).  Anything else is the per-engine
    // pass, engine by engine — same results.
    const ManyKind kind = many_kind(engines, n);
    bool one_stream = true;
    for (u32 i = 0; i < n; ++i)
        if (engines[i]->stream != e0->stream) one_stream = false;
    if (kind == kManySingle) {
        for (u32 i = 0; i < n; ++i) {
            int rc = rafthip_commit(engines[i]);
            if (rc) return rc;
        }
        return RAFTHIP_OK;
    }
    DeviceGuard guard(e0->device);
    hipStream_t st = e0->stream;
    hipEvent_t ev = nullptr;
    if (!one_stream) {
        // the launch runs on engines[0]'s stream: it waits for what the other engines' streams hold,
        // and they wait for it
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (u32 i = 1; i < n; ++i) {
            if (engines[i]->stream == st) continue;
            hipError_t err = hipStreamQuery(engines[i]->stream);  // an idle stream has nothing to wait for
            if (err == hipSuccess) continue;
            if (err != hipErrorNotReady) {
                (void)hipEventDestroy(ev);
                return fail(RAFTHIP_EIO, "hipStreamQuery: %s", hipGetErrorString(err));
            }
            (void)hipGetLastError();
            err = hipEventRecord(ev, engines[i]->stream);
            if (err == hipSuccess) err = hipStreamWaitEvent(st, ev, 0);
            if (err != hipSuccess) {
                (void)hipEventDestroy(ev);
                return fail(RAFTHIP_EIO, "stream join: %s", hipGetErrorString(err));
            }
        }
    }
    int rc = RAFTHIP_OK;
    for (u32 lo = 0; lo < n && !rc; lo += kManyMax) {
        ManyShards sh{};
        sh.n = n - lo < (u32)kManyMax ? n - lo : (u32)kManyMax;
        u32 tiles = 0;
        for (u32 k = 0; k < sh.n; ++k) {
            rafthip_engine* e = engines[lo + k];
            sh.first_tile[k] = tiles;
            sh.match[k] = e->match;
            sh.commit[k] = e->commit;
            sh.stride[k] = e->stride;
            tiles += (u32)((e->G + kTileGroups - 1) / kTileGroups);
            e->stats.commit_launches++;
            e->stats.groups_evaluated += e->G;
        }
        sh.first_tile[sh.n] = tiles;
        if (kind == kManyUniform) {
            switch (e0->uniform_voters) {
#define RAFTHIP_CASE_MANY(V)                                                                             \
    case V:                                                                                             \
        hipLaunchKernelGGL((k_commit_uniform_many<V>), dim3(tiles), dim3(kCommitBlock), 0, st, sh);      \
        break;
                RAFTHIP_CASE_MANY(1) RAFTHIP_CASE_MANY(2) RAFTHIP_CASE_MANY(3) RAFTHIP_CASE_MANY(4)
                RAFTHIP_CASE_MANY(5) RAFTHIP_CASE_MANY(6) RAFTHIP_CASE_MANY(7) RAFTHIP_CASE_MANY(8)
#undef RAFTHIP_CASE_MANY
            }
        } else {
            ManyMasks mk{};
            fill_many_masks(engines + lo, sh.n, &mk);
            switch (e0->S) {
#define RAFTHIP_CASE_MANYM(SS)                                                                           \
    case SS:                                                                                            \
        hipLaunchKernelGGL((k_commit_masked_many<SS>), dim3(tiles), dim3(kCommitBlock), 0, st, sh, mk);  \
        break;
                RAFTHIP_CASE_MANYM(1) RAFTHIP_CASE_MANYM(2) RAFTHIP_CASE_MANYM(3) RAFTHIP_CASE_MANYM(4)
                RAFTHIP_CASE_MANYM(5) RAFTHIP_CASE_MANYM(6) RAFTHIP_CASE_MANYM(7) RAFTHIP_CASE_MANYM(8)
#undef RAFTHIP_CASE_MANYM
            }
        }
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) rc = fail(RAFTHIP_EIO, "k_commit_uniform_many: %s", hipGetErrorString(err));
    }
    if (ev) {
        hipError_t err = hipEventRecord(ev, st);
        for (u32 i = 1; i < n && err == hipSuccess; ++i)
            if (engines[i]->stream != st) err = hipStreamWaitEvent(engines[i]->stream, ev, 0);
        (void)hipEventDestroy(ev);
        if (err != hipSuccess && !rc) rc = fail(RAFTHIP_EIO, "stream join: %s", hipGetErrorString(err));
    }
    return rc;
}

}  // extern "C"

// rafthip_commit_changed / rafthip_commit_advance over several engines.  One GPU, the K1 shape and the same
// voter count in every shard: one commit launch, one pack launch and one wait for all of them; otherwise engine by
// engine — same results.
template <int MODE>
static int commit_report_many(rafthip_engine* const* engines, u32 n, rafthip_commit_pair* const* dst, const uint64_t* max_pairs,
                              uint64_t* n_out) {
    rafthip_engine* e0 = engines[0];
    for (u32 i = 0; i < n; ++i)
        for (u32 k = 0; k < i; ++k)
            if (engines[k] == engines[i]) return fail(RAFTHIP_EINVAL, "engines[%u] and engines[%u] are the same engine", k, i);
    const ManyKind kind = n <= (u32)kManyMax ? many_kind(engines, n) : kManySingle;
    if (kind == kManySingle) {
        for (u32 i = 0; i < n; ++i) {
            int rc = commit_report<MODE>(engines[i], dst ? dst[i] : nullptr, max_pairs ? max_pairs[i] : 0, &n_out[i]);
            if (rc) return rc;
        }
        return RAFTHIP_OK;
    }
    DeviceGuard guard(e0->device);
    hipStream_t st = e0->stream;
    ManyReport sh{};
    ManyPack pk{};
    ReportBufs rb[kManyMax];
    Pair* d_out[kManyMax];
    bool landing[kManyMax];
    u64 cap[kManyMax];
    sh.n = pk.n = n;
    u32 tiles = 0;
    int rc;
    for (u32 i = 0; i < n; ++i) {
        rafthip_engine* e = engines[i];
        const u64 mp = (dst && dst[i] && max_pairs) ? max_pairs[i] : 0;
        if ((rc = ensure_report_buffers(e, &rb[i]))) return rc;
        if (MODE == kAdvance) {
            if ((rc = ensure(e, e->term_first, (size_t)e->stride * sizeof(u64)))) return rc;
            if ((rc = ensure(e, e->term_last, (size_t)e->stride * sizeof(u64)))) return rc;
        }
        d_out[i] = mp ? device_addressable(dst[i]) : nullptr;
        landing[i] = mp && !d_out[i];
        cap[i] = mp;
        if (landing[i]) {
            if ((rc = ensure(e, e->packed, (size_t)e->stride * sizeof(Pair), false))) return rc;
            d_out[i] = e->packed;
            cap[i] = mp < e->G ? mp : e->G;
        }
    }
    // Everything runs on engines[0]'s stream.  Another engine's stream is joined to it only if it still holds work
    // (an upload's memset, a scatter): an idle stream costs one query.  No join afterwards: the call waits for the
    // GPU before it returns, so nothing the caller issues later can overtake it.
    hipEvent_t ev = nullptr;
    for (u32 i = 1; i < n; ++i) {
        if (engines[i]->stream == st) continue;
        hipError_t err = hipStreamQuery(engines[i]->stream);
        if (err == hipSuccess) continue;
        if (err != hipErrorNotReady) return fail(RAFTHIP_EIO, "hipStreamQuery: %s", hipGetErrorString(err));
        (void)hipGetLastError();
        if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        err = hipEventRecord(ev, engines[i]->stream);
        if (err == hipSuccess) err = hipStreamWaitEvent(st, ev, 0);
        if (err != hipSuccess) {
            (void)hipEventDestroy(ev);
            return fail(RAFTHIP_EIO, "stream join: %s", hipGetErrorString(err));
        }
    }
    if (ev) (void)hipEventDestroy(ev);  // released once the recorded work is done
    for (u32 i = 0; i < n; ++i) {
        rafthip_engine* e = engines[i];
        const u32 t = (u32)((e->G + kTileGroups - 1) / kTileGroups);
        sh.first_tile[i] = pk.first_tile[i] = tiles;
        tiles += t;
        sh.match[i] = e->match;
        sh.commit[i] = e->commit;
        sh.stride[i] = e->stride;
        sh.groups[i] = e->G;
        sh.term_start[i] = e->term_first;
        sh.last_index[i] = e->term_last;
        sh.bits[i] = rb[i].bits;
        sh.counts[i] = rb[i].counts;
        sh.super[i] = rb[i].super;
        sh.super_next[i] = rb[i].super_next;
        sh.super_words[i] = super_words(rb[i].tiles_cap);
        pk.bits[i] = rb[i].bits;
        pk.counts[i] = rb[i].counts;
        pk.super[i] = rb[i].super;
        pk.commit[i] = e->commit;
        pk.out[i] = d_out[i];
        pk.max_pairs[i] = d_out[i] ? cap[i] : 0;
        pk.total[i] = e->d_scalar + 1;  // lands in h_scalar[1]
        e->stats.commit_launches++;
        e->stats.groups_evaluated += e->G;
    }
    sh.first_tile[n] = pk.first_tile[n] = tiles;
    if (kind == kManyUniform) {
        switch (e0->uniform_voters) {
#define RAFTHIP_CASE_MR(V)                                                                                     \
    case V:                                                                                                   \
        hipLaunchKernelGGL((k_commit_uniform_many_report<V, MODE>), dim3(tiles), dim3(kCommitBlock), 0, st, sh); \
        break;
            RAFTHIP_CASE_MR(1) RAFTHIP_CASE_MR(2) RAFTHIP_CASE_MR(3) RAFTHIP_CASE_MR(4)
            RAFTHIP_CASE_MR(5) RAFTHIP_CASE_MR(6) RAFTHIP_CASE_MR(7) RAFTHIP_CASE_MR(8)
#undef RAFTHIP_CASE_MR
        }
    } else {
        ManyMasks mk{};
        fill_many_masks(engines, n, &mk);
        switch (e0->S) {
#define RAFTHIP_CASE_MRM(SS)                                                                                        \
    case SS:                                                                                                       \
        hipLaunchKernelGGL((k_commit_masked_many_report<SS, MODE>), dim3(tiles), dim3(kCommitBlock), 0, st, sh, mk); \
        break;
            RAFTHIP_CASE_MRM(1) RAFTHIP_CASE_MRM(2) RAFTHIP_CASE_MRM(3) RAFTHIP_CASE_MRM(4)
            RAFTHIP_CASE_MRM(5) RAFTHIP_CASE_MRM(6) RAFTHIP_CASE_MRM(7) RAFTHIP_CASE_MRM(8)
#undef RAFTHIP_CASE_MRM
        }
    }
    hipLaunchKernelGGL(k_pack_changed_many, dim3(tiles), dim3(kBlock), 0, st, pk);
    hipError_t err = hipGetLastError();
    for (u32 i = 0; i < n; ++i) engines[i]->rep_flip ^= 1;  // the arrays this pass zeroed are the next one's
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (err != hipSuccess) return fail(RAFTHIP_EIO, "commit_report_many: %s", hipGetErrorString(err));
    bool copies = false;
    for (u32 i = 0; i < n; ++i) {
        rafthip_engine* e = engines[i];
        const u64 total = e->h_scalar[1];
        n_out[i] = total;
        const u64 mp = (dst && dst[i] && max_pairs) ? max_pairs[i] : 0;
        const u64 ncopy = total < mp ? total : mp;
        if (landing[i] && ncopy) {
            HIP_TRY(hipMemcpyAsync(dst[i], e->packed, ncopy * sizeof(Pair), hipMemcpyDeviceToHost, st));
            copies = true;
        }
        e->stats.bytes_downloaded += ncopy * sizeof(Pair);
        if (MODE == kAdvance) {
            e->stats.advanced_reported += total;
        } else {
            e->stats.changed_reported += total;
        }
    }
    if (copies) HIP_TRY(hipStreamSynchronize(st));
    return RAFTHIP_OK;
}

extern "C" {

int rafthip_commit_report_many(rafthip_engine* const* engines, uint32_t n, int advance, rafthip_commit_pair* const* dst,
                               const uint64_t* max_pairs, uint64_t* n_out) {
    if (!engines || n == 0 || !n_out) return fail(RAFTHIP_EINVAL, "engines/n_out is NULL or n is 0");
    for (u32 i = 0; i < n; ++i) {
        if (!engines[i]) return fail(RAFTHIP_EINVAL, "engines[%u] is NULL", i);
        if (max_pairs && max_pairs[i] > 0 && (!dst || !dst[i])) return fail(RAFTHIP_EINVAL, "dst[%u] is NULL with max_pairs > 0", i);
    }
    return advance ? commit_report_many<kAdvance>(engines, n, dst, max_pairs, n_out)
                   : commit_report_many<kChanged>(engines, n, dst, max_pairs, n_out);
}

int rafthip_commit_changed(rafthip_engine* e, rafthip_commit_pair* dst, uint64_t max_pairs,
                           uint64_t* n_changed) {
    return commit_report<kChanged>(e, dst, max_pairs, n_changed);
}

int rafthip_commit_advance(rafthip_engine* e, rafthip_commit_pair* dst, uint64_t max_pairs,
                           uint64_t* n_advanced) {
    return commit_report<kAdvance>(e, dst, max_pairs, n_advanced);
}

int rafthip_commit_changed_static(rafthip_engine* e, rafthip_commit_pair* dst, uint64_t max_pairs,
                                  uint64_t* n_changed) {
    if (!e || !n_changed) return fail(RAFTHIP_EINVAL, "engine/count pointer is NULL");
    if (max_pairs > 0 && !dst) return fail(RAFTHIP_EINVAL, "dst is NULL with max_pairs > 0");
    if (e->stride > kStaticMaxGroups)
        return fail(RAFTHIP_EINVAL, "rafthip_commit_changed_static serves engines of at most %llu groups (this one: %llu); "
                                    "use rafthip_commit_changed", (u64)kStaticMaxGroups, (u64)e->G);
    if (e->uniform && e->uniform_voters == 0 && !e->in_mask)
        return fail(RAFTHIP_EINVAL, "rafthip_commit_changed_static: set a config first (the empty default needs mask planes, "
                                    "which this entry point may not allocate)");
    DeviceGuard guard(e->device);
    Pair* d_out = nullptr;
    if (max_pairs) {
        d_out = device_addressable(dst);
        if (!d_out)
            return fail(RAFTHIP_EINVAL, "rafthip_commit_changed_static: dst must be device memory or page-locked host memory "
                                        "(rafthip_host_alloc); a pageable buffer would need a landing buffer");
    }
    void* base = nullptr;
    int rc = static_scratch(e->device, &base);  // resolved by rafthip_create: a table read
    if (rc) return rc;
    std::unique_lock<std::mutex> busy(g_scratch_busy[e->device], std::try_to_lock);
    if (!busy.owns_lock())
        return fail(RAFTHIP_EBUSY, "rafthip_commit_changed_static: another call is using device %d's scratch array "
                                   "(calls on one GPU share it: serialise them)", e->device);
    unsigned char* p = static_cast<unsigned char*>(base);
    ReportBufs rb;
    rb.bits = reinterpret_cast<u64*>(p);
    rb.counts = reinterpret_cast<u32*>(p + kStaticBitsBytes);
    rb.super = reinterpret_cast<u32*>(p + kStaticBitsBytes + kStaticCountsBytes);
    rb.super_next = nullptr;  // the array is shared by every engine on this GPU: zeroed in front of each pass
    rb.total = reinterpret_cast<u64*>(p + kStaticBitsBytes + kStaticCountsBytes + kStaticSuperBytes - 8);
    rb.tiles_cap = (u32)(e->stride / kBlock);
    if ((rc = report_pass<kChanged>(e, rb, 0, d_out, max_pairs, e->d_scalar + 1))) return rc;  // the total lands in h_scalar[1]
    HIP_TRY(hipStreamSynchronize(e->stream));
    *n_changed = e->h_scalar[1];
    e->stats.changed_reported += *n_changed;
    return RAFTHIP_OK;
}

int rafthip_upload_term_range(rafthip_engine* e, uint64_t first, uint64_t count, const uint64_t* term_first,
                              const uint64_t* term_last) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!term_first || !term_last) return fail(RAFTHIP_EINVAL, "term_first/term_last is NULL");
    DeviceGuard guard(e->device);
    if ((rc = ensure(e, e->term_first, (size_t)e->stride * sizeof(u64)))) return rc;
    if ((rc = ensure(e, e->term_last, (size_t)e->stride * sizeof(u64)))) return rc;
    HIP_TRY(hipMemcpyAsync(e->term_first + first, term_first, count * 8, hipMemcpyDefault, e->stream));
    HIP_TRY(hipMemcpyAsync(e->term_last + first, term_last, count * 8, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_uploaded += count * 16;
    return RAFTHIP_OK;
}

int rafthip_read_term_range(rafthip_engine* e, uint64_t first, uint64_t count, uint64_t* term_first,
                            uint64_t* term_last) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!term_first && !term_last) return fail(RAFTHIP_EINVAL, "term_first and term_last are both NULL");
    if (!e->term_first || !e->term_last) return fail(RAFTHIP_EINVAL, "no term range has been uploaded yet");
    DeviceGuard guard(e->device);
    if (term_first) HIP_TRY(hipMemcpyAsync(term_first, e->term_first + first, count * 8, hipMemcpyDefault, e->stream));
    if (term_last) HIP_TRY(hipMemcpyAsync(term_last, e->term_last + first, count * 8, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_downloaded += count * 8 * ((term_first ? 1 : 0) + (term_last ? 1 : 0));
    return RAFTHIP_OK;
}

int rafthip_committed_in_term(rafthip_engine* e) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    DeviceGuard guard(e->device);
    int rc;
    if ((rc = ensure(e, e->term_first, (size_t)e->stride * sizeof(u64)))) return rc;
    if ((rc = ensure(e, e->term_last, (size_t)e->stride * sizeof(u64)))) return rc;
    if ((rc = ensure(e, e->in_term, e->stride))) return rc;
    const u32 blocks = (u32)(e->stride / ((u64)kBlock * 2));
    hipLaunchKernelGGL(k_committed_in_term, dim3(blocks), dim3(kBlock), 0, e->stream, e->commit, e->term_first,
                       e->term_last, e->in_term);
    HIP_TRY(hipGetLastError());
    e->stats.groups_evaluated += e->G;
    return RAFTHIP_OK;
}

int rafthip_read_committed_in_term(rafthip_engine* e, uint64_t first, uint64_t count, uint8_t* dst) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!dst) return fail(RAFTHIP_EINVAL, "dst is NULL");
    DeviceGuard guard(e->device);
    if ((rc = read_u8(e, e->in_term, "committed-in-term", first, count, dst))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RAFTHIP_OK;
}

int rafthip_heartbeat_commit(rafthip_engine* e, uint64_t first, uint64_t count, uint64_t* dst,
                             uint64_t dst_stride) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (dst && dst_stride < count)
        return fail(RAFTHIP_EINVAL, "dst_stride %llu < count %llu", (u64)dst_stride, (u64)count);
    DeviceGuard guard(e->device);
    if ((rc = ensure(e, e->hb_commit, (size_t)e->S * e->stride * sizeof(u64)))) return rc;
    const u32 blocks = (u32)(e->stride / ((u64)kBlock * 2));  // stride is a multiple of 4096
    hipLaunchKernelGGL(k_heartbeat_commit, dim3(blocks), dim3(kBlock), 0, e->stream, e->match, e->stride,
                       e->commit, e->hb_commit, e->S);
    HIP_TRY(hipGetLastError());
    e->stats.groups_evaluated += e->G;
    if (!dst) return RAFTHIP_OK;  // result stays on the device: rafthip_plane(RAFTHIP_PLANE_HEARTBEAT)
    if ((rc = copy_rows(e, dst, dst_stride * 8, e->hb_commit + first, e->stride * 8, count * 8, e->S))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_downloaded += count * 8 * e->S;
    return RAFTHIP_OK;
}

static_assert(kHeartbeatStride == RAFTHIP_HEARTBEAT_STRIDE, "heartbeat stride");
static_assert(kStaticMaxGroups == RAFTHIP_STATIC_MAX_GROUPS, "static scratch limit");

static int emit_heartbeats(rafthip_engine* e, uint64_t first, uint64_t count, uint8_t* out_bytes, uint8_t* out_len, bool due);

int rafthip_emit_heartbeats(rafthip_engine* e, uint64_t first, uint64_t count, uint8_t* out_bytes, uint8_t* out_len) {
    return emit_heartbeats(e, first, count, out_bytes, out_len, false);
}

int rafthip_emit_heartbeats_due(rafthip_engine* e, uint64_t first, uint64_t count, uint8_t* out_bytes, uint8_t* out_len) {
    return emit_heartbeats(e, first, count, out_bytes, out_len, true);
}

static int emit_heartbeats(rafthip_engine* e, uint64_t first, uint64_t count, uint8_t* out_bytes, uint8_t* out_len, bool due) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (due && !e->tick_action) return fail(RAFTHIP_EINVAL, "rafthip_tick has not run on this engine");
    if (!out_bytes || !out_len) return fail(RAFTHIP_EINVAL, "out_bytes/out_len is NULL");
    if (!e->have_ids || !e->have_term)
        return fail(RAFTHIP_EINVAL, "rafthip_upload_ids and rafthip_upload_term must come first");
    if (count == 0) return RAFTHIP_OK;
    DeviceGuard guard(e->device);
    if ((rc = ensure(e, e->self_mask, (size_t)e->stride * e->mbytes))) return rc;
    const size_t cells = (size_t)e->S * count;
    const bool bytes_host = !is_device_ptr(out_bytes), len_host = !is_device_ptr(out_len);
    uint8_t *d_bytes = out_bytes, *d_len = out_len;
    if (bytes_host) {
        if ((rc = stage_reserve(e, kStageBytes, cells * kHeartbeatStride))) return rc;
        d_bytes = static_cast<uint8_t*>(e->stage_buf[kStageBytes]);
    } else if (reinterpret_cast<uintptr_t>(out_bytes) % 8 != 0) {
        return fail(RAFTHIP_EINVAL, "out_bytes on the device must be 8-byte aligned");
    }
    if (len_host) {
        if ((rc = stage_reserve(e, kStageStatus, cells))) return rc;
        d_len = static_cast<uint8_t*>(e->stage_buf[kStageStatus]);
    }
    const u32 blocks = (u32)((count + kBlock - 1) / kBlock);
    const u32 H = wire_hdr_words(e->S);
#define RAFTHIP_EMIT(MT)                                                                                        \
    hipLaunchKernelGGL((k_emit_heartbeats<MT>), dim3(blocks), dim3(kBlock), 0, e->stream, e->match, e->stride,   \
                       e->commit, static_cast<const u64*>(e->wire_hdr), H, static_cast<const MT*>(e->self_mask), \
                       (u64)first, (u64)count, e->S, d_bytes, d_len, due ? e->tick_action : nullptr)
    if (e->mbytes == 1) {
        RAFTHIP_EMIT(uint8_t);
    } else if (e->mbytes == 2) {
        RAFTHIP_EMIT(uint16_t);
    } else if (e->mbytes == 4) {
        RAFTHIP_EMIT(uint32_t);
    } else {
        RAFTHIP_EMIT(uint64_t);
    }
#undef RAFTHIP_EMIT
    HIP_TRY(hipGetLastError());
    e->stats.groups_evaluated += count;
    if (bytes_host) {
        HIP_TRY(hipMemcpyAsync(out_bytes, d_bytes, cells * kHeartbeatStride, hipMemcpyDeviceToHost, e->stream));
        e->stats.bytes_downloaded += cells * kHeartbeatStride;
    }
    if (len_host) {
        HIP_TRY(hipMemcpyAsync(out_len, d_len, cells, hipMemcpyDeviceToHost, e->stream));
        e->stats.bytes_downloaded += cells;
    }
    if (bytes_host || len_host) HIP_TRY(hipStreamSynchronize(e->stream));
    return RAFTHIP_OK;
}

int rafthip_read_commit(rafthip_engine* e, uint64_t first, uint64_t count, uint64_t* dst) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!dst) return fail(RAFTHIP_EINVAL, "dst is NULL");
    DeviceGuard guard(e->device);
    HIP_TRY(hipMemcpyAsync(dst, e->commit + first, count * 8, hipMemcpyDefault, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_downloaded += count * 8;
    return RAFTHIP_OK;
}

// The vote tally kernels over e's planes (votes, masks, vote_result set; granted / rejected may be NULL).
static int launch_tally_votes(rafthip_engine* e) {
    const bool uni = e->uniform && e->uniform_voters > 0;
    hipStream_t st = e->stream;
    // A uniform config of <= 8 voters only needs its voter planes (the other rows
    // are learners: no vote, no count), whatever S is: the 8-slot kernel at S := V.
    const bool small = e->S <= 8 || (uni && e->uniform_voters <= 8);
    if (small) {
        const u32 blocks = (u32)((e->G + (u64)kBlock * 16 - 1) / ((u64)kBlock * 16));
        const uint8_t* m0 = uni ? nullptr : static_cast<const uint8_t*>(e->in_mask);
        const uint8_t* m1 = uni ? nullptr : static_cast<const uint8_t*>(e->out_mask);
        const uint8_t* ml = uni ? nullptr : static_cast<const uint8_t*>(e->learner_mask);
        switch (e->S <= 8 ? e->S : e->uniform_voters) {
#define RAFTHIP_CASE_T(SS)                                                                          \
    case SS:                                                                                        \
        hipLaunchKernelGGL((k_tally_votes_small<SS>), dim3(blocks), dim3(kBlock), 0, st, e->votes,   \
                           e->stride, m0, m1, ml, e->uniform_voters, e->vote_result, e->granted,    \
                           e->rejected);                                                            \
        break;
            RAFTHIP_CASE_T(1) RAFTHIP_CASE_T(2) RAFTHIP_CASE_T(3) RAFTHIP_CASE_T(4)
            RAFTHIP_CASE_T(5) RAFTHIP_CASE_T(6) RAFTHIP_CASE_T(7) RAFTHIP_CASE_T(8)
#undef RAFTHIP_CASE_T
        }
    } else {
        const u64 per = (u64)kBlock * 4 * kVoteWideNW;  // 4096 groups per workgroup = one plane granule
        const u32 blocks = (u32)((e->G + per - 1) / per);
#define RAFTHIP_TW(MT)                                                                               \
    hipLaunchKernelGGL((k_tally_votes_wide<MT>), dim3(blocks), dim3(kBlock), 0, st, e->votes, e->stride, \
                       (int)e->S, uni ? nullptr : static_cast<const MT*>(e->in_mask),                \
                       uni ? nullptr : static_cast<const MT*>(e->out_mask),                          \
                       uni ? nullptr : static_cast<const MT*>(e->learner_mask),                      \
                       (int)e->uniform_voters, e->vote_result, e->granted, e->rejected)
        if (e->mbytes == 2) {
            RAFTHIP_TW(uint16_t);
        } else if (e->mbytes == 4) {
            RAFTHIP_TW(uint32_t);
        } else {
            RAFTHIP_TW(uint64_t);
        }
#undef RAFTHIP_TW
    }
    HIP_TRY(hipGetLastError());
    e->stats.groups_evaluated += e->G;
    return RAFTHIP_OK;
}

int rafthip_tally_votes(rafthip_engine* e) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    DeviceGuard guard(e->device);
    int rc;
    if ((rc = ensure(e, e->votes, (size_t)e->S * e->stride))) return rc;
    if ((rc = ensure(e, e->vote_result, e->stride))) return rc;
    if ((rc = ensure(e, e->granted, e->stride))) return rc;
    if ((rc = ensure(e, e->rejected, e->stride))) return rc;
    if (e->uniform && e->uniform_voters == 0 && (rc = ensure_masks(e))) return rc;
    return launch_tally_votes(e);
}

int rafthip_read_vote_result(rafthip_engine* e, uint64_t first, uint64_t count, uint8_t* result,
                             uint8_t* granted, uint8_t* rejected) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!result) return fail(RAFTHIP_EINVAL, "result is NULL");
    DeviceGuard guard(e->device);
    if ((rc = read_u8(e, e->vote_result, "vote result", first, count, result))) return rc;
    if (granted && (rc = read_u8(e, e->granted, "granted", first, count, granted))) return rc;
    if (rejected && (rc = read_u8(e, e->rejected, "rejected", first, count, rejected))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RAFTHIP_OK;
}

int rafthip_quorum_active(rafthip_engine* e) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    DeviceGuard guard(e->device);
    return tally_bits<false>(e, e->active, e->qactive, "RecentActive");
}

int rafthip_upload_self(rafthip_engine* e, uint64_t first, uint64_t count, const void* src) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    return upload_bits(e, e->self_mask, first, count, src);
}

int rafthip_check_quorum(rafthip_engine* e) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    DeviceGuard guard(e->device);
    int rc;
    // groups whose self mask was never uploaded keep no bit: every RecentActive is cleared
    if ((rc = ensure(e, e->self_mask, (size_t)e->stride * e->mbytes))) return rc;
    return tally_bits<false>(e, e->active, e->qactive, "RecentActive", e->self_mask);
}

// ---- RawNode.Tick for every group (tick.h) ---------------------------------

static int ensure_tick(rafthip_engine* e) {
    int rc;
    if ((rc = ensure(e, e->tick_el, (size_t)e->stride * 4))) return rc;
    if ((rc = ensure(e, e->tick_hb, (size_t)e->stride * 4))) return rc;
    if ((rc = ensure(e, e->tick_rt, (size_t)e->stride * 4))) return rc;
    if ((rc = ensure(e, e->tick_role, (size_t)e->stride))) return rc;
    if ((rc = ensure(e, e->tick_flags, (size_t)e->stride))) return rc;
    if ((rc = ensure(e, e->tick_action, (size_t)e->stride))) return rc;
    return ensure(e, e->d_tick_counts, sizeof(u64) * kTickCountCopies * kTickCountStride);
}

int rafthip_tick_config(rafthip_engine* e, uint32_t election_tick, uint32_t heartbeat_tick, int check_quorum) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    // Config.validate, raft.go:296-302
    if (heartbeat_tick == 0) return fail(RAFTHIP_EINVAL, "heartbeat tick must be greater than 0");
    if (election_tick <= heartbeat_tick) return fail(RAFTHIP_EINVAL, "election tick must be greater than heartbeat tick");
    if (election_tick > 0x7FFFFFFFu) return fail(RAFTHIP_EINVAL, "election tick %u does not fit the reference's int", election_tick);
    e->election_timeout = election_tick;
    e->heartbeat_timeout = heartbeat_tick;
    e->check_quorum = check_quorum != 0;
    return RAFTHIP_OK;
}

int rafthip_upload_tick_state(rafthip_engine* e, uint64_t first, uint64_t count, const uint32_t* election_elapsed,
                              const uint32_t* heartbeat_elapsed, const uint32_t* randomized_timeout,
                              const uint8_t* role, const uint8_t* flags) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    DeviceGuard guard(e->device);
    if ((rc = ensure_tick(e))) return rc;
    const struct {
        void* dst;
        const void* src;
        size_t width;
    } rows[] = {{e->tick_el, election_elapsed, 4}, {e->tick_hb, heartbeat_elapsed, 4}, {e->tick_rt, randomized_timeout, 4},
                {e->tick_role, role, 1},           {e->tick_flags, flags, 1}};
    for (const auto& r : rows) {
        if (!r.src || count == 0) continue;
        HIP_TRY(hipMemcpyAsync(static_cast<char*>(r.dst) + first * r.width, r.src, count * r.width, hipMemcpyDefault, e->stream));
        e->stats.bytes_uploaded += count * r.width;
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RAFTHIP_OK;
}

int rafthip_read_tick_state(rafthip_engine* e, uint64_t first, uint64_t count, uint32_t* election_elapsed,
                            uint32_t* heartbeat_elapsed, uint32_t* randomized_timeout, uint8_t* role, uint8_t* flags) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    DeviceGuard guard(e->device);
    if ((rc = ensure_tick(e))) return rc;
    const struct {
        const void* src;
        void* dst;
        size_t width;
    } rows[] = {{e->tick_el, election_elapsed, 4}, {e->tick_hb, heartbeat_elapsed, 4}, {e->tick_rt, randomized_timeout, 4},
                {e->tick_role, role, 1},           {e->tick_flags, flags, 1}};
    for (const auto& r : rows) {
        if (!r.dst || count == 0) continue;
        HIP_TRY(hipMemcpyAsync(r.dst, static_cast<const char*>(r.src) + first * r.width, count * r.width, hipMemcpyDefault, e->stream));
        e->stats.bytes_downloaded += count * r.width;
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RAFTHIP_OK;
}

int rafthip_read_tick_actions(rafthip_engine* e, uint64_t first, uint64_t count, uint8_t* dst) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!dst) return fail(RAFTHIP_EINVAL, "dst is NULL");
    DeviceGuard guard(e->device);
    if ((rc = read_u8(e, e->tick_action, "tick actions", first, count, dst))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RAFTHIP_OK;
}

// What a tick of engine e needs before its kernel can run (the caller has made e's device current).
static int tick_prepare(rafthip_engine* e) {
    if (e->election_timeout == 0) return fail(RAFTHIP_EINVAL, "rafthip_tick_config has not been called");
    int rc;
    if ((rc = ensure_tick(e))) return rc;
    if (e->check_quorum) {
        if (!e->active) return fail(RAFTHIP_EINVAL, "RecentActive bits have not been uploaded");
        // groups whose self mask was never uploaded keep no bit, as in rafthip_check_quorum
        if ((rc = ensure(e, e->self_mask, (size_t)e->stride * e->mbytes))) return rc;
        if (e->uniform && e->uniform_voters == 0 && (rc = ensure_masks(e))) return rc;
    }
    return RAFTHIP_OK;
}

// The per-action counts of engine e's last counting tick, summed over the counter sets (after the stream has drained).
static void tick_sum_counts(const u64* sets, uint64_t* counts) {
    for (int b = 0; b < kTickActions; ++b) {
        u64 sum = 0;
        for (int c = 0; c < kTickCountCopies; ++c) sum += sets[c * kTickCountStride + b];
        counts[b] = sum;
    }
}

int rafthip_tick_many(rafthip_engine* const* engines, uint32_t n, uint64_t* counts) {
    if (!engines || n == 0) return fail(RAFTHIP_EINVAL, "engines is NULL or n is 0");
    for (u32 i = 0; i < n; ++i)
        if (!engines[i]) return fail(RAFTHIP_EINVAL, "engines[%u] is NULL", i);
    for (u32 i = 1; i < n; ++i)
        for (u32 j = 0; j < i; ++j)
            if (engines[i] == engines[j]) return fail(RAFTHIP_EINVAL, "engines[%u] and engines[%u] are the same engine", j, i);
    rafthip_engine* e0 = engines[0];
    // one launch needs one GPU and one mask width; anything else is the per-engine pass, engine by engine
    bool same = true;
    for (u32 i = 1; i < n; ++i)
        if (engines[i]->device != e0->device || engines[i]->mbytes != e0->mbytes) same = false;
    if (!same || n == 1) {
        for (u32 i = 0; i < n; ++i) {
            int rc = rafthip_tick(engines[i], counts ? counts + (size_t)i * RAFTHIP_TICK_ACTIONS : nullptr);
            if (rc) return rc;
        }
        return RAFTHIP_OK;
    }
    DeviceGuard guard(e0->device);
    int rc;
    for (u32 i = 0; i < n; ++i)
        if ((rc = tick_prepare(engines[i]))) return rc;
    hipStream_t st = e0->stream;
    hipEvent_t ev = nullptr;
    bool one_stream = true;
    for (u32 i = 1; i < n; ++i)
        if (engines[i]->stream != st) one_stream = false;
    if (!one_stream) {  // as rafthip_commit_many: the launch waits for what the other engines' streams hold, and they for it
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (u32 i = 1; i < n; ++i) {
            if (engines[i]->stream == st) continue;
            hipError_t err = hipStreamQuery(engines[i]->stream);
            if (err == hipSuccess) continue;
            if (err != hipErrorNotReady) {
                (void)hipEventDestroy(ev);
                return fail(RAFTHIP_EIO, "hipStreamQuery: %s", hipGetErrorString(err));
            }
            (void)hipGetLastError();
            err = hipEventRecord(ev, engines[i]->stream);
            if (err == hipSuccess) err = hipStreamWaitEvent(st, ev, 0);
            if (err != hipSuccess) {
                (void)hipEventDestroy(ev);
                return fail(RAFTHIP_EIO, "stream join: %s", hipGetErrorString(err));
            }
        }
    }
    rc = RAFTHIP_OK;
    for (u32 lo = 0; lo < n && !rc; lo += kTickManyMax) {
        TickShards sh{};
        sh.n = n - lo < (u32)kTickManyMax ? n - lo : (u32)kTickManyMax;
        u32 blocks = 0;
        for (u32 k = 0; k < sh.n && !rc; ++k) {
            rafthip_engine* e = engines[lo + k];
            const bool uni = e->uniform && e->uniform_voters > 0;
            sh.first_block[k] = blocks;
            sh.planes[k] = TickPlanes{e->tick_el, e->tick_hb, e->tick_rt, e->tick_role, e->tick_flags, e->tick_action};
            sh.groups[k] = e->G;
            sh.active[k] = e->active;
            sh.in_mask[k] = uni ? nullptr : e->in_mask;
            sh.out_mask[k] = uni ? nullptr : e->out_mask;
            sh.learner_mask[k] = uni ? nullptr : e->learner_mask;
            sh.self_mask[k] = e->self_mask;
            sh.counts[k] = counts ? e->d_tick_counts : nullptr;
            sh.election_timeout[k] = e->election_timeout;
            sh.heartbeat_timeout[k] = e->heartbeat_timeout;
            sh.check_quorum[k] = e->check_quorum ? 1 : 0;
            sh.slots[k] = (uint8_t)e->S;
            sh.uniform_voters[k] = (uint8_t)e->uniform_voters;
            blocks += (u32)((e->G + (u64)kBlock * 4 - 1) / ((u64)kBlock * 4));
            e->stats.groups_evaluated += e->G;
            if (counts) {
                hipError_t err = hipMemsetAsync(e->d_tick_counts, 0, sizeof(u64) * kTickCountCopies * kTickCountStride, st);
                if (err != hipSuccess) rc = fail(RAFTHIP_EIO, "hipMemsetAsync: %s", hipGetErrorString(err));
            }
        }
        sh.first_block[sh.n] = blocks;
        if (rc) break;
        if (e0->mbytes == 1) {
            hipLaunchKernelGGL((k_tick_many<uint8_t>), dim3(blocks), dim3(kBlock), 0, st, sh);
        } else if (e0->mbytes == 2) {
            hipLaunchKernelGGL((k_tick_many<uint16_t>), dim3(blocks), dim3(kBlock), 0, st, sh);
        } else if (e0->mbytes == 4) {
            hipLaunchKernelGGL((k_tick_many<uint32_t>), dim3(blocks), dim3(kBlock), 0, st, sh);
        } else {
            hipLaunchKernelGGL((k_tick_many<uint64_t>), dim3(blocks), dim3(kBlock), 0, st, sh);
        }
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) rc = fail(RAFTHIP_EIO, "k_tick_many: %s", hipGetErrorString(err));
    }
    if (ev) {
        hipError_t err = hipEventRecord(ev, st);
        for (u32 i = 1; i < n && err == hipSuccess; ++i)
            if (engines[i]->stream != st) err = hipStreamWaitEvent(engines[i]->stream, ev, 0);
        (void)hipEventDestroy(ev);
        if (err != hipSuccess && !rc) rc = fail(RAFTHIP_EIO, "stream join: %s", hipGetErrorString(err));
    }
    if (rc || !counts) return rc;
    // the counter sets of every shard in one wait
    std::vector<u64> sets((size_t)n * kTickCountCopies * kTickCountStride);
    for (u32 i = 0; i < n; ++i)
        HIP_TRY(hipMemcpyAsync(sets.data() + (size_t)i * kTickCountCopies * kTickCountStride, engines[i]->d_tick_counts,
                               sizeof(u64) * kTickCountCopies * kTickCountStride, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (u32 i = 0; i < n; ++i)
        tick_sum_counts(sets.data() + (size_t)i * kTickCountCopies * kTickCountStride, counts + (size_t)i * RAFTHIP_TICK_ACTIONS);
    return RAFTHIP_OK;
}

int rafthip_tick(rafthip_engine* e, uint64_t* counts) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    DeviceGuard guard(e->device);
    int rc;
    if ((rc = tick_prepare(e))) return rc;
    const bool uni = e->uniform && e->uniform_voters > 0;
    hipStream_t st = e->stream;
    static_assert(kTickActions == RAFTHIP_TICK_ACTIONS && kTickActions <= kTickCountStride, "rafthip.h and tick.h disagree");
    if (counts) HIP_TRY(hipMemsetAsync(e->d_tick_counts, 0, sizeof(u64) * kTickCountCopies * kTickCountStride, st));
    const TickPlanes tp{e->tick_el, e->tick_hb, e->tick_rt, e->tick_role, e->tick_flags, e->tick_action};
    const u32 blocks = (u32)((e->G + (u64)kBlock * 4 - 1) / ((u64)kBlock * 4));
#define RAFTHIP_TK(MT)                                                                                             \
    hipLaunchKernelGGL((k_tick<MT>), dim3(blocks), dim3(kBlock), 0, st, tp, e->G, (int)e->S, e->election_timeout,    \
                       e->heartbeat_timeout, e->check_quorum ? 1u : 0u, static_cast<MT*>(e->active),                 \
                       uni ? nullptr : static_cast<const MT*>(e->in_mask),                                           \
                       uni ? nullptr : static_cast<const MT*>(e->out_mask),                                          \
                       uni ? nullptr : static_cast<const MT*>(e->learner_mask), (int)e->uniform_voters,              \
                       static_cast<const MT*>(e->self_mask), counts ? e->d_tick_counts : nullptr)
    if (blocks) {
        if (e->mbytes == 1) {
            RAFTHIP_TK(uint8_t);
        } else if (e->mbytes == 2) {
            RAFTHIP_TK(uint16_t);
        } else if (e->mbytes == 4) {
            RAFTHIP_TK(uint32_t);
        } else {
            RAFTHIP_TK(uint64_t);
        }
    }
#undef RAFTHIP_TK
    HIP_TRY(hipGetLastError());
    e->stats.groups_evaluated += e->G;
    if (counts) {
        u64 sets[kTickCountCopies * kTickCountStride];
        HIP_TRY(hipMemcpyAsync(sets, e->d_tick_counts, sizeof(sets), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        tick_sum_counts(sets, counts);
    }
    return RAFTHIP_OK;
}

int rafthip_tick_heard_from_leader(rafthip_engine* e, const uint32_t* groups, uint64_t n, uint64_t* rejected) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    if (n && !groups) return fail(RAFTHIP_EINVAL, "groups is NULL");
    if (n > (1ull << 32)) return fail(RAFTHIP_EINVAL, "at most 2^32 group ids per call");
    DeviceGuard guard(e->device);
    int rc;
    if ((rc = ensure_tick(e))) return rc;
    if (rejected) *rejected = 0;
    if (n == 0) return RAFTHIP_OK;
    const void* d_groups = nullptr;
    if ((rc = stage_input(e, kStageGroups, groups, (size_t)n * sizeof(u32), &d_groups))) return rc;
    // the out-of-range count shares the engine's counter words (the tick's own sets are cleared by every counting tick)
    u64* d_bad = e->d_tick_counts + (kTickCountCopies - 1) * kTickCountStride + kTickActions;
    HIP_TRY(hipMemsetAsync(d_bad, 0, sizeof(u64), e->stream));
    hipLaunchKernelGGL(k_tick_heard, dim3((u32)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, e->stream, e->tick_el, e->G,
                       static_cast<const u32*>(d_groups), (u64)n, d_bad);
    HIP_TRY(hipGetLastError());
    u64 bad = 0;
    if (rejected) HIP_TRY(hipMemcpyAsync(&bad, d_bad, sizeof(u64), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));  // the kernel may have read the caller's own (page-locked) memory
    e->zero_copy_pending = false;
    if (rejected) *rejected = bad;
    return RAFTHIP_OK;
}

int rafthip_read_tick_events(rafthip_engine* e, uint32_t mask, rafthip_tick_event* dst, uint64_t max_events,
                             uint64_t* n_events) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    if (!n_events) return fail(RAFTHIP_EINVAL, "n_events is NULL");
    if (max_events && !dst) return fail(RAFTHIP_EINVAL, "dst is NULL");
    if (!e->tick_action) return fail(RAFTHIP_EINVAL, "rafthip_tick has not run on this engine");
    if (e->G > 0xFFFFFFFFull) return fail(RAFTHIP_EINVAL, "a tick event holds a 32-bit group id");
    static_assert(sizeof(TickEvent) == sizeof(rafthip_tick_event), "rafthip.h and tick.h disagree");
    DeviceGuard guard(e->device);
    const u32 tiles = (u32)((e->G + kEvTile - 1) / kEvTile);
    int rc;
    if ((rc = ensure(e, e->tick_ev, sizeof(u64) + (size_t)tiles * sizeof(u32)))) return rc;
    u32* tile_counts = reinterpret_cast<u32*>(e->tick_ev + 1);
    hipStream_t st = e->stream;
    hipLaunchKernelGGL(k_tick_event_count, dim3(tiles), dim3(kBlock), 0, st, e->tick_action, e->G, mask & 0xFFu, tile_counts);
    hipLaunchKernelGGL(k_tick_event_scan, dim3(1), dim3(kBlock), 0, st, tile_counts, tiles, e->tick_ev);
    HIP_TRY(hipGetLastError());
    u64 total = 0;
    HIP_TRY(hipMemcpyAsync(&total, e->tick_ev, sizeof(u64), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *n_events = total;
    const u64 take = total < max_events ? total : max_events;
    if (take == 0) return RAFTHIP_OK;
    // device memory and page-locked host memory are written by the kernel itself; pageable memory through a staging buffer
    TickEvent* out = nullptr;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, dst) == hipSuccess) {
        if (attr.type == hipMemoryTypeDevice) out = reinterpret_cast<TickEvent*>(dst);
        else if (attr.type == hipMemoryTypeHost && attr.devicePointer) out = static_cast<TickEvent*>(attr.devicePointer);
    }
    (void)hipGetLastError();
    const bool staged = out == nullptr;
    if (staged) {
        if ((rc = stage_reserve(e, kStageAux, (size_t)take * sizeof(TickEvent)))) return rc;
        out = static_cast<TickEvent*>(e->stage_buf[kStageAux]);
    }
    hipLaunchKernelGGL(k_tick_event_pack, dim3(tiles), dim3(kBlock), 0, st, e->tick_action, e->G, mask & 0xFFu, tile_counts,
                       (u64)0, out, take);
    HIP_TRY(hipGetLastError());
    if (staged) HIP_TRY(hipMemcpyAsync(dst, out, (size_t)take * sizeof(TickEvent), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    e->stats.bytes_downloaded += take * sizeof(TickEvent);
    return RAFTHIP_OK;
}

static int read_bits(rafthip_engine* e, const void* plane, const char* what, uint64_t first, uint64_t count,
                     void* dst) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!dst) return fail(RAFTHIP_EINVAL, "dst is NULL");
    if (!plane) return fail(RAFTHIP_EINVAL, "%s bits do not exist yet", what);
    DeviceGuard guard(e->device);
    const size_t mb = (size_t)e->mbytes;
    HIP_TRY(hipMemcpyAsync(dst, static_cast<const char*>(plane) + first * mb, count * mb, hipMemcpyDefault,
                           e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->stats.bytes_downloaded += count * mb;
    return RAFTHIP_OK;
}

int rafthip_read_active(rafthip_engine* e, uint64_t first, uint64_t count, void* dst) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    return read_bits(e, e->active, "RecentActive", first, count, dst);
}

int rafthip_read_acks(rafthip_engine* e, uint64_t first, uint64_t count, void* dst) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    return read_bits(e, e->acks, "ReadIndex ack", first, count, dst);
}

int rafthip_read_quorum_active(rafthip_engine* e, uint64_t first, uint64_t count, uint8_t* dst) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!dst) return fail(RAFTHIP_EINVAL, "dst is NULL");
    DeviceGuard guard(e->device);
    if ((rc = read_u8(e, e->qactive, "quorum-active", first, count, dst))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RAFTHIP_OK;
}

int rafthip_ack_quorum(rafthip_engine* e) {
    if (!e) return fail(RAFTHIP_EINVAL, "engine is NULL");
    DeviceGuard guard(e->device);
    return tally_bits<true>(e, e->acks, e->ack_result, "ReadIndex ack");
}

int rafthip_read_ack_quorum(rafthip_engine* e, uint64_t first, uint64_t count, uint8_t* dst) {
    int rc = check_range(e, first, count);
    if (rc) return rc;
    if (!dst) return fail(RAFTHIP_EINVAL, "dst is NULL");
    DeviceGuard guard(e->device);
    if ((rc = read_u8(e, e->ack_result, "ack-quorum", first, count, dst))) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    return RAFTHIP_OK;
}

int rafthip_plane(rafthip_engine* e, int which, void** ptr, uint64_t* stride) {
    if (!e || !ptr) return fail(RAFTHIP_EINVAL, "engine/ptr is NULL");
    DeviceGuard guard(e->device);
    int rc;
    switch (which) {
        case RAFTHIP_PLANE_MATCH: *ptr = e->match; break;
        case RAFTHIP_PLANE_COMMIT: *ptr = e->commit; break;
        case RAFTHIP_PLANE_VOTES:
            if ((rc = ensure(e, e->votes, (size_t)e->S * e->stride))) return rc;
            *ptr = e->votes;
            break;
        case RAFTHIP_PLANE_HEARTBEAT:
            if ((rc = ensure(e, e->hb_commit, (size_t)e->S * e->stride * sizeof(u64)))) return rc;
            *ptr = e->hb_commit;
            break;
        default: return fail(RAFTHIP_EINVAL, "unknown plane %d", which);
    }
    if (stride) *stride = e->stride;
    return RAFTHIP_OK;
}

// ---- C1: the exchange step over RCCL ------------------------------------------
}  // extern "C"

struct rafthip_comm;
// The loopback transport (rafthip_internal.h): n engines on ONE GPU stand for n ranks, and an
// all-gather is n x n device copies ordered with events.  Everything around the gather — the
// reporting pass with group_base, the speculative width and its redo, k_pack_gathered, clipping
// at cap, the landing buffers, the dense form — is the code the RCCL transport runs.
struct LoopGroup {
    int n = 0;
    int refs = 0;
    rafthip_comm* member[64] = {};
    hipEvent_t ready[64] = {};  // rank r's send buffer is complete on its stream
    hipEvent_t done[64] = {};   // rank i has copied every rank's contribution
};

struct rafthip_comm {
    rafthip_engine* e = nullptr;
    ncclComm_t nccl = nullptr;
    LoopGroup* loop = nullptr;  // set instead of nccl for a loopback communicator
    int rank = 0, nranks = 1;
    u64 group_base = 0;
    u64 cap = 0;             // pairs per rank, equal on every rank
    Pair* d_send = nullptr;  // [cap] this rank's pairs, global group ids
    u64* d_counts = nullptr; // [nranks]
    Pair* d_recv = nullptr;  // [nranks][width <= cap]
    Pair* d_out = nullptr;   // [nranks * cap] landing buffer for a pageable dst
    u64* h_counts = nullptr; // pinned [nranks]
    u64 last_max = 0;        // largest per-rank count of the previous exchange
    bool have_history = false;
    u64 width = 0;           // dense form: elements per rank = the largest plane stride of any rank
    u64* d_plane_send = nullptr;  // [width] only when this rank's own plane is shorter than width
    u64* d_plane_recv = nullptr;  // [nranks * width] landing buffer of the dense form for a pageable dst (first use)
};

namespace {

#define NCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t _r = (expr);                                                                   \
        if (_r != ncclSuccess) return fail(RAFTHIP_EIO, "%s: %s", #expr, ncclGetErrorString(_r));   \
    } while (0)

constexpr u64 kGatherGranule = 256;  // payload widths are multiples of this many pairs

void comm_free(rafthip_comm* c) {
    if (!c) return;
    if (c->e) {
        DeviceGuard guard(c->e->device);
        if (c->e->stream) (void)hipStreamSynchronize(c->e->stream);
        if (c->nccl) (void)ncclCommDestroy(c->nccl);
        if (c->loop) {
            LoopGroup* g = c->loop;
            if (c->rank >= 0 && c->rank < g->n) g->member[c->rank] = nullptr;
            if (--g->refs == 0) {
                for (int r = 0; r < g->n; ++r) {
                    if (g->ready[r]) (void)hipEventDestroy(g->ready[r]);
                    if (g->done[r]) (void)hipEventDestroy(g->done[r]);
                }
                delete g;
            }
        }
        for (void* b : {(void*)c->d_send, (void*)c->d_counts, (void*)c->d_recv, (void*)c->d_out, (void*)c->d_plane_send,
                        (void*)c->d_plane_recv})
            if (b) (void)hipFree(b);
        if (c->h_counts) (void)hipHostFree(c->h_counts);
    }
    delete c;
}

// The small buffers of a communicator whose nccl handle, rank and nranks are set, and everything
// the engine allocates lazily for a reporting pass, so that an exchange allocates nothing.
int comm_finish_create(rafthip_comm* c) {
    rafthip_engine* e = c->e;
    int rc;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_counts), (size_t)c->nranks * 2 * sizeof(u64)));  // 2 words per rank at create
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_counts), (size_t)c->nranks * 2 * sizeof(u64), hipHostMallocDefault));
    ReportBufs rb;
    if ((rc = ensure_report_buffers(e, &rb))) return rc;  // so that the exchange itself allocates nothing
    if ((rc = ensure(e, e->term_first, (size_t)e->stride * sizeof(u64)))) return rc;
    if ((rc = ensure(e, e->term_last, (size_t)e->stride * sizeof(u64)))) return rc;
    return RAFTHIP_OK;
}

int comm_alloc_payload(rafthip_comm* c) {
    const size_t one = (size_t)c->cap * sizeof(Pair);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_send), one));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_recv), one * c->nranks));
    return RAFTHIP_OK;
}

// Enqueue the packing of the gathered slots into `out` (device-addressable) on c's stream.
int enqueue_pack_gathered(rafthip_comm* c, u64 width, Pair* out, u64 max_pairs) {
    if (width == 0 || max_pairs == 0) return RAFTHIP_OK;
    const dim3 grid((u32)((width + kBlock - 1) / kBlock), (u32)c->nranks);
    hipLaunchKernelGGL(k_pack_gathered, grid, dim3(kBlock), 0, c->e->stream, static_cast<const Pair*>(c->d_recv),
                       static_cast<const u64*>(c->d_counts), (u32)c->nranks, width, out, max_pairs);
    HIP_TRY(hipGetLastError());
    return RAFTHIP_OK;
}

// ---- the one indirection under every collective of this file ---------------------------------
// An all-gather of `count` uint64 words per rank on each comms[i] (the ranks this thread drives),
// enqueued on the engines' streams.  RCCL: one ncclAllGather per rank, inside one RCCL group when
// there are several; the group is always closed, and the first error is what comes back (a failed
// exchange leaves the communicator unusable: destroy it and create a new one — the peers of a rank
// that failed here are still inside the collective).  Loopback: device copies ordered with events.
struct GatherArgs {
    const void* send;
    void* recv;
    size_t count;
};

int loop_all_gather(rafthip_comm** comms, int n, const GatherArgs* a) {
    LoopGroup* g = comms[0]->loop;
    if (n != g->n) return fail(RAFTHIP_EINVAL, "loopback exchange: %d of %d ranks in the call (all of them run in one thread)", n, g->n);
    for (int i = 0; i < n; ++i) {
        if (comms[i]->loop != g || comms[i]->rank != i) return fail(RAFTHIP_EINVAL, "loopback exchange: comms[%d] is not rank %d of the group", i, i);
        if (a[i].count != a[0].count) return fail(RAFTHIP_EINVAL, "loopback exchange: ranks disagree on the element count");
    }
    const size_t bytes = a[0].count * sizeof(u64);
    if (bytes == 0) return RAFTHIP_OK;
    DeviceGuard guard(comms[0]->e->device);
    for (int r = 0; r < n; ++r) HIP_TRY(hipEventRecord(g->ready[r], comms[r]->e->stream));
    for (int i = 0; i < n; ++i) {
        hipStream_t st = comms[i]->e->stream;
        for (int r = 0; r < n; ++r) {
            if (r != i) HIP_TRY(hipStreamWaitEvent(st, g->ready[r], 0));
            void* d = static_cast<char*>(a[i].recv) + (size_t)r * bytes;
            if (d != a[r].send) HIP_TRY(hipMemcpyAsync(d, a[r].send, bytes, hipMemcpyDefault, st));  // in place: own slot already there
        }
        HIP_TRY(hipEventRecord(g->done[i], st));
    }
    // like the collective it stands for: a rank's send buffer is free again only when every rank has read it
    for (int r = 0; r < n; ++r)
        for (int i = 0; i < n; ++i)
            if (i != r) HIP_TRY(hipStreamWaitEvent(comms[r]->e->stream, g->done[i], 0));
    return RAFTHIP_OK;
}

int all_gather_many(rafthip_comm** comms, int n, const GatherArgs* a) {
    if (comms[0]->loop) return loop_all_gather(comms, n, a);
    int rc = RAFTHIP_OK;
    const bool grouped = n > 1;
    if (grouped) NCCL_TRY(ncclGroupStart());
    for (int i = 0; i < n; ++i) {
        rafthip_comm* c = comms[i];
        DeviceGuard guard(c->e->device);
        const ncclResult_t r = ncclAllGather(a[i].send, a[i].recv, a[i].count, ncclUint64, c->nccl, c->e->stream);
        if (r != ncclSuccess && !rc) rc = fail(RAFTHIP_EIO, "ncclAllGather (rank %d): %s", c->rank, ncclGetErrorString(r));
    }
    if (grouped) {
        const ncclResult_t r = ncclGroupEnd();  // always closed, also after a failed call inside it
        if (r != ncclSuccess && !rc) rc = fail(RAFTHIP_EIO, "ncclGroupEnd: %s", ncclGetErrorString(r));
    }
    return rc;
}

template <int MODE>
int exchange_many(rafthip_comm** comms, int n, rafthip_commit_pair** dst, const uint64_t* max_pairs,
                  uint64_t** counts_out, uint64_t* n_total) {
    GatherArgs ga[kMaxDevices];
    // 1. every engine's commit pass: pairs with global ids into its send buffer, count in *rb.total
    ReportBufs rb[kMaxDevices];
    for (int i = 0; i < n; ++i) {
        rafthip_comm* c = comms[i];
        DeviceGuard guard(c->e->device);
        int rc = ensure_report_buffers(c->e, &rb[i]);  // allocated at create: a lookup
        if (rc) return rc;
        if ((rc = report_pass<MODE>(c->e, rb[i], c->group_base, c->d_send, c->cap))) return rc;
    }
    // 2. counts to every rank
    for (int i = 0; i < n; ++i) ga[i] = GatherArgs{rb[i].total, comms[i]->d_counts, 1};
    {
        int rc = all_gather_many(comms, n, ga);
        if (rc) return rc;
    }
    for (int i = 0; i < n; ++i) {
        rafthip_comm* c = comms[i];
        DeviceGuard guard(c->e->device);
        HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, (size_t)c->nranks * sizeof(u64), hipMemcpyDeviceToHost, c->e->stream));
    }
    // 3. payload width: from the previous exchange when there was one (no host wait between the two
    //    collectives; checked afterwards), else from the counts themselves
    const rafthip_comm* c0 = comms[0];
    u64 width = 0;
    bool speculative = c0->have_history;
    auto width_for = [&](u64 largest) {
        u64 w = round_up(largest, kGatherGranule);
        return w > c0->cap ? c0->cap : w;
    };
    auto largest_count = [&]() {
        u64 m = 0;
        for (int r = 0; r < c0->nranks; ++r) m = c0->h_counts[r] > m ? c0->h_counts[r] : m;
        return m < c0->cap ? m : c0->cap;  // a rank sends at most cap pairs
    };
    if (speculative) {
        width = width_for(c0->last_max + c0->last_max / 2 + 1);  // half again as much as last time
    } else {
        for (int i = 0; i < n; ++i) {
            DeviceGuard guard(comms[i]->e->device);
            HIP_TRY(hipStreamSynchronize(comms[i]->e->stream));
        }
        width = width_for(largest_count());
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (width) {
            for (int i = 0; i < n; ++i) ga[i] = GatherArgs{comms[i]->d_send, comms[i]->d_recv, (size_t)width * 2};
            int rc = all_gather_many(comms, n, ga);
            if (rc) return rc;
        }
        for (int i = 0; i < n; ++i) {
            rafthip_comm* c = comms[i];
            DeviceGuard guard(c->e->device);
            const u64 mp = (dst && dst[i] && max_pairs) ? max_pairs[i] : 0;
            Pair* direct = mp ? device_addressable(dst[i]) : nullptr;  // else: through the landing buffer
            const u64 room = direct ? mp : (mp < (u64)c->nranks * c->cap ? mp : (u64)c->nranks * c->cap);
            int rc = enqueue_pack_gathered(c, width, direct ? direct : c->d_out, room);
            if (rc) return rc;
        }
        for (int i = 0; i < n; ++i) {
            DeviceGuard guard(comms[i]->e->device);
            HIP_TRY(hipStreamSynchronize(comms[i]->e->stream));
        }
        const u64 need = width_for(largest_count());
        if (need <= width) break;  // every rank's pairs fitted the slots that travelled
        width = need;              // the guess was too small (same verdict on every rank): once more, exact
    }
    // 4. hand back
    for (int i = 0; i < n; ++i) {
        rafthip_comm* c = comms[i];
        DeviceGuard guard(c->e->device);
        u64 total = 0, delivered = 0;
        for (int r = 0; r < c->nranks; ++r) {
            total += c->h_counts[r];
            delivered += c->h_counts[r] < c->cap ? c->h_counts[r] : c->cap;
            if (counts_out && counts_out[i]) counts_out[i][r] = c->h_counts[r];
        }
        const u64 mp = (dst && dst[i] && max_pairs) ? max_pairs[i] : 0;
        if (mp && !device_addressable(dst[i])) {
            const u64 ncopy = delivered < mp ? delivered : mp;
            if (ncopy) {
                HIP_TRY(hipMemcpyAsync(dst[i], c->d_out, ncopy * sizeof(Pair), hipMemcpyDeviceToHost, c->e->stream));
                HIP_TRY(hipStreamSynchronize(c->e->stream));
                c->e->stats.bytes_downloaded += ncopy * sizeof(Pair);
            }
        }
        if (n_total) n_total[i] = total;
        const u64 mine = c->h_counts[c->rank];
        if (MODE == kAdvance) {
            c->e->stats.advanced_reported += mine;
        } else {
            c->e->stats.changed_reported += mine;
        }
        c->last_max = largest_count();
        c->have_history = true;
    }
    return RAFTHIP_OK;
}

// The dense form: every engine's commit pass, then one all-gather of the commit planes — [rank][width]
// — into dst[i] (device-addressable: straight into it; pageable: through a landing buffer).
int exchange_dense(rafthip_comm** comms, int n, int advance, uint64_t** dst, const uint64_t* dst_elems, uint64_t* width_out) {
    bool query = true;  // every dst NULL: report the width, do nothing else (a local call, no collective)
    for (int i = 0; i < n; ++i) query = query && (!dst || !dst[i]);
    if (query) {
        if (!width_out) return fail(RAFTHIP_EINVAL, "dense exchange: dst and width are both NULL");
        for (int i = 0; i < n; ++i) width_out[i] = comms[i]->width;
        return RAFTHIP_OK;
    }
    for (int i = 0; i < n; ++i) {
        rafthip_comm* c = comms[i];
        const u64 need = (u64)c->nranks * c->width;
        if (!dst || !dst[i] || !dst_elems || dst_elems[i] < need)
            return fail(RAFTHIP_EINVAL, "dense exchange: dst[%d] needs room for %llu elements (nranks x width)", i, need);
    }
    u64* target[kMaxDevices];
    for (int i = 0; i < n; ++i) {
        rafthip_comm* c = comms[i];
        rafthip_engine* e = c->e;
        DeviceGuard guard(e->device);
        int rc;
        if (advance) {  // the term-guarded pass; its pair report goes nowhere
            ReportBufs rb;
            if ((rc = ensure_report_buffers(e, &rb))) return rc;
            if ((rc = report_pass<kAdvance>(e, rb, 0, nullptr, 0))) return rc;
        } else if ((rc = rafthip_commit(e))) {
            return rc;
        }
        if (c->d_plane_send)  // this rank's plane is shorter than the width every rank sends
            HIP_TRY(hipMemcpyAsync(c->d_plane_send, e->commit, (size_t)e->stride * sizeof(u64), hipMemcpyDeviceToDevice, e->stream));
        u64* direct = nullptr;
        {
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, dst[i]) == hipSuccess) {
                if (attr.type == hipMemoryTypeDevice) direct = reinterpret_cast<u64*>(dst[i]);
                else if (attr.type == hipMemoryTypeHost && attr.devicePointer) direct = static_cast<u64*>(attr.devicePointer);
            }
            (void)hipGetLastError();
        }
        if (!direct) {
            if ((rc = ensure(e, c->d_plane_recv, (size_t)c->nranks * c->width * sizeof(u64), false))) return rc;
            direct = c->d_plane_recv;
        }
        target[i] = direct;
    }
    GatherArgs ga[kMaxDevices];
    for (int i = 0; i < n; ++i) {
        const rafthip_comm* c = comms[i];
        ga[i] = GatherArgs{c->d_plane_send ? c->d_plane_send : c->e->commit, target[i], (size_t)c->width};
    }
    {
        int rc = all_gather_many(comms, n, ga);
        if (rc) return rc;
    }
    for (int i = 0; i < n; ++i) {
        rafthip_comm* c = comms[i];
        DeviceGuard guard(c->e->device);
        const size_t bytes = (size_t)c->nranks * c->width * sizeof(u64);
        if (target[i] == c->d_plane_recv) {
            HIP_TRY(hipMemcpyAsync(dst[i], c->d_plane_recv, bytes, hipMemcpyDeviceToHost, c->e->stream));
            c->e->stats.bytes_downloaded += bytes;
        }
        HIP_TRY(hipStreamSynchronize(c->e->stream));
        if (width_out) width_out[i] = c->width;
    }
    return RAFTHIP_OK;
}

int check_comms(rafthip_comm** comms, int n) {
    if (!comms || n < 1 || n > kMaxDevices) return fail(RAFTHIP_EINVAL, "comms is NULL or n outside [1, %d]", kMaxDevices);
    for (int i = 0; i < n; ++i)
        if (!comms[i] || !comms[i]->e || (!comms[i]->nccl && !comms[i]->loop))
            return fail(RAFTHIP_EINVAL, "comms[%d] is not a live communicator", i);
    for (int i = 1; i < n; ++i)
        if ((comms[i]->loop != nullptr) != (comms[0]->loop != nullptr) || comms[i]->nranks != comms[0]->nranks)
            return fail(RAFTHIP_EINVAL, "comms[%d] and comms[0] belong to different communicators", i);
    if (comms[0]->loop && n != comms[0]->nranks)
        return fail(RAFTHIP_EINVAL, "loopback exchange: %d of %d ranks in the call (all of them run in one thread)", n, comms[0]->nranks);
    return RAFTHIP_OK;
}

// The second half of a communicator's creation, after the in-place all-gather of every rank's two
// words (pair capacity asked for, plane stride): the capacity every rank uses is the largest one
// asked for, the dense width the largest plane stride; then the payload buffers.
int comm_adopt_agreement(rafthip_comm* c) {
    rafthip_engine* e = c->e;
    int rc;
    HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, (size_t)c->nranks * 2 * sizeof(u64), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int k = 0; k < c->nranks; ++k) {
        c->cap = c->h_counts[2 * k] > c->cap ? c->h_counts[2 * k] : c->cap;
        c->width = c->h_counts[2 * k + 1] > c->width ? c->h_counts[2 * k + 1] : c->width;
    }
    c->cap = round_up(c->cap, kGatherGranule);
    if (c->width > e->stride && (rc = ensure(e, c->d_plane_send, (size_t)c->width * sizeof(u64)))) return rc;  // zeroed
    if ((rc = comm_alloc_payload(c))) return rc;
    return ensure(e, c->d_out, (size_t)c->cap * c->nranks * sizeof(Pair), false);
}

}  // namespace

extern "C" {

static_assert(sizeof(rafthip_comm_id) == sizeof(ncclUniqueId), "rafthip_comm_id must hold an ncclUniqueId");

int rafthip_comm_unique_id(rafthip_comm_id* id) {
    if (!id) return fail(RAFTHIP_EINVAL, "id is NULL");
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    memcpy(id->bytes, &u, sizeof(u));
    return RAFTHIP_OK;
}

// After the nccl handle exists: agree on the per-rank pair capacity (the largest any rank asked
// for), then allocate.  max_pairs_per_rank 0 = this engine's group count.
static int comm_agree_and_alloc(rafthip_comm* c, uint64_t max_pairs_per_rank) {
    rafthip_engine* e = c->e;
    int rc = comm_finish_create(c);
    if (rc) return rc;
    c->h_counts[2 * c->rank] = max_pairs_per_rank ? max_pairs_per_rank : e->G;  // pinned: safe source of an async copy
    c->h_counts[2 * c->rank + 1] = e->stride;
    HIP_TRY(hipMemcpyAsync(c->d_counts + 2 * c->rank, c->h_counts + 2 * c->rank, 2 * sizeof(u64), hipMemcpyHostToDevice, e->stream));
    return RAFTHIP_OK;
}

int rafthip_comm_create(rafthip_engine* e, const rafthip_comm_id* id, int rank, int nranks, uint64_t group_base,
                        uint64_t max_pairs_per_rank, rafthip_comm** out) {
    if (!e || !id || !out) return fail(RAFTHIP_EINVAL, "engine/id/out is NULL");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(RAFTHIP_EINVAL, "rank %d outside [0, %d)", rank, nranks);
    DeviceGuard guard(e->device);
    rafthip_comm* c = new (std::nothrow) rafthip_comm();
    if (!c) return fail(RAFTHIP_ENOMEM, "host allocation failed");
    c->e = e;
    c->rank = rank;
    c->nranks = nranks;
    c->group_base = group_base;
    ncclUniqueId u;
    memcpy(&u, id->bytes, sizeof(u));
    int rc = RAFTHIP_OK;
    do {
        ncclResult_t r = ncclCommInitRank(&c->nccl, nranks, u, rank);
        if (r != ncclSuccess) {
            c->nccl = nullptr;
            rc = fail(RAFTHIP_EIO, "ncclCommInitRank: %s", ncclGetErrorString(r));
            break;
        }
        int cnt = 0;
        r = ncclCommCount(c->nccl, &cnt);
        if (r != ncclSuccess || cnt != nranks) {
            rc = fail(RAFTHIP_EIO, "ncclCommCount: the communicator has %d ranks, %d were asked for (%s)", cnt, nranks,
                      r == ncclSuccess ? "ok" : ncclGetErrorString(r));
            break;
        }
        if ((rc = comm_agree_and_alloc(c, max_pairs_per_rank))) break;
        // the capacity every rank uses = the largest one asked for, the dense width = the largest plane
        // stride: an in-place all-gather of both + max
        const GatherArgs ga{c->d_counts + 2 * rank, c->d_counts, 2};
        if ((rc = all_gather_many(&c, 1, &ga))) break;
        rc = comm_adopt_agreement(c);
    } while (0);
    if (rc) {
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        comm_free(c);
        memcpy(g_err, keep, sizeof(keep));
        return rc;
    }
    *out = c;
    return RAFTHIP_OK;
}

int rafthip_comm_create_all(rafthip_engine** engines, int n, const uint64_t* group_base, uint64_t max_pairs_per_rank,
                            rafthip_comm** out) {
    if (!engines || !out || !group_base) return fail(RAFTHIP_EINVAL, "engines/group_base/out is NULL");
    if (n < 1 || n > kMaxDevices) return fail(RAFTHIP_EINVAL, "n %d outside [1, %d]", n, kMaxDevices);
    int devs[kMaxDevices];
    u64 cap = max_pairs_per_rank;
    for (int i = 0; i < n; ++i) {
        out[i] = nullptr;
        if (!engines[i]) return fail(RAFTHIP_EINVAL, "engines[%d] is NULL", i);
        devs[i] = engines[i]->device;
        for (int k = 0; k < i; ++k)
            if (devs[k] == devs[i]) return fail(RAFTHIP_EINVAL, "engines %d and %d share device %d: one rank per GPU", k, i, devs[i]);
        if (!max_pairs_per_rank && engines[i]->G > cap) cap = engines[i]->G;
    }
    u64 width = 0;
    for (int i = 0; i < n; ++i) width = engines[i]->stride > width ? engines[i]->stride : width;
    cap = round_up(cap, kGatherGranule);
    ncclComm_t handles[kMaxDevices];
    NCCL_TRY(ncclCommInitAll(handles, n, devs));
    int rc = RAFTHIP_OK;
    for (int i = 0; i < n; ++i) {
        rafthip_comm* c = new (std::nothrow) rafthip_comm();
        if (!c) {
            rc = fail(RAFTHIP_ENOMEM, "host allocation failed");
            (void)ncclCommDestroy(handles[i]);
            continue;
        }
        out[i] = c;
        c->e = engines[i];
        c->nccl = handles[i];
        c->rank = i;
        c->nranks = n;
        c->group_base = group_base[i];
        c->cap = cap;
        c->width = width;
        if (rc) continue;
        DeviceGuard guard(c->e->device);
        if ((rc = comm_finish_create(c))) continue;
        if ((rc = comm_alloc_payload(c))) continue;
        if ((rc = ensure(c->e, c->d_out, (size_t)c->cap * n * sizeof(Pair), false))) continue;
        if (width > c->e->stride) rc = ensure(c->e, c->d_plane_send, (size_t)width * sizeof(u64));
    }
    if (rc) {
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        for (int i = 0; i < n; ++i) {
            comm_free(out[i]);
            out[i] = nullptr;
        }
        memcpy(g_err, keep, sizeof(keep));
    }
    return rc;
}

// rafthip_internal.h — n engines on ONE GPU as the n ranks of a loopback communicator.  Creation walks
// the per-process path of rafthip_comm_create rank by rank (each rank's two words, the in-place
// all-gather, the agreement), not the host-side shortcut of rafthip_comm_create_all.
int rafthip_internal_comm_create_loopback(rafthip_engine** engines, int n, const uint64_t* group_base,
                                          const uint64_t* max_pairs_per_rank, rafthip_comm** out) {
    if (!engines || !out || !group_base) return fail(RAFTHIP_EINVAL, "engines/group_base/out is NULL");
    if (n < 1 || n > kMaxDevices) return fail(RAFTHIP_EINVAL, "n %d outside [1, %d]", n, kMaxDevices);
    for (int i = 0; i < n; ++i) {
        out[i] = nullptr;
        if (!engines[i]) return fail(RAFTHIP_EINVAL, "engines[%d] is NULL", i);
        if (engines[i]->device != engines[0]->device) return fail(RAFTHIP_EINVAL, "loopback ranks share one GPU; engines[%d] is on another", i);
        for (int k = 0; k < i; ++k)
            if (engines[k] == engines[i]) return fail(RAFTHIP_EINVAL, "engines[%d] and engines[%d] are the same engine", k, i);
    }
    DeviceGuard guard(engines[0]->device);
    LoopGroup* g = new (std::nothrow) LoopGroup();
    if (!g) return fail(RAFTHIP_ENOMEM, "host allocation failed");
    g->n = n;
    int rc = RAFTHIP_OK;
    for (int i = 0; i < n && !rc; ++i) {
        hipError_t err = hipEventCreateWithFlags(&g->ready[i], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&g->done[i], hipEventDisableTiming);
        if (err != hipSuccess) rc = fail(RAFTHIP_EIO, "hipEventCreate: %s", hipGetErrorString(err));
    }
    for (int i = 0; i < n && !rc; ++i) {
        rafthip_comm* c = new (std::nothrow) rafthip_comm();
        if (!c) {
            rc = fail(RAFTHIP_ENOMEM, "host allocation failed");
            break;
        }
        out[i] = c;
        c->e = engines[i];
        c->loop = g;
        g->refs++;
        g->member[i] = c;
        c->rank = i;
        c->nranks = n;
        c->group_base = group_base[i];
        rc = comm_agree_and_alloc(c, max_pairs_per_rank ? max_pairs_per_rank[i] : 0);
    }
    if (!rc) {
        GatherArgs ga[kMaxDevices];
        for (int i = 0; i < n; ++i) ga[i] = GatherArgs{out[i]->d_counts + 2 * i, out[i]->d_counts, 2};
        rc = all_gather_many(out, n, ga);
    }
    for (int i = 0; i < n && !rc; ++i) rc = comm_adopt_agreement(out[i]);
    if (rc) {
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        const bool orphan = g->refs == 0;
        for (int i = 0; i < n; ++i) {
            comm_free(out[i]);
            out[i] = nullptr;
        }
        if (orphan) {
            for (int r = 0; r < n; ++r) {
                if (g->ready[r]) (void)hipEventDestroy(g->ready[r]);
                if (g->done[r]) (void)hipEventDestroy(g->done[r]);
            }
            delete g;
        }
        memcpy(g_err, keep, sizeof(keep));
    }
    return rc;
}

int rafthip_comm_destroy(rafthip_comm* c) {
    comm_free(c);
    return RAFTHIP_OK;
}

int rafthip_comm_info(const rafthip_comm* c, int* rank, int* nranks, uint64_t* max_pairs_per_rank) {
    if (!c) return fail(RAFTHIP_EINVAL, "comm is NULL");
    if (rank) *rank = c->rank;
    if (nranks) {
        *nranks = c->nranks;
        if (c->nccl) {  // RCCL's own count: what a scaling run reports as "ranks the communicator really has"
            int cnt = 0;
            NCCL_TRY(ncclCommCount(c->nccl, &cnt));
            *nranks = cnt;
        }
    }
    if (max_pairs_per_rank) *max_pairs_per_rank = c->cap;
    return RAFTHIP_OK;
}

int rafthip_commit_allgather(rafthip_comm* c, int advance, rafthip_commit_pair* dst, uint64_t max_pairs,
                             uint64_t* counts, uint64_t* n_total) {
    int rc = check_comms(&c, 1);
    if (rc) return rc;
    if (max_pairs > 0 && !dst) return fail(RAFTHIP_EINVAL, "dst is NULL with max_pairs > 0");
    return advance ? exchange_many<kAdvance>(&c, 1, &dst, &max_pairs, &counts, n_total)
                   : exchange_many<kChanged>(&c, 1, &dst, &max_pairs, &counts, n_total);
}

int rafthip_commit_allgather_dense(rafthip_comm* c, int advance, uint64_t* dst, uint64_t dst_elems, uint64_t* width) {
    int rc = check_comms(&c, 1);
    if (rc) return rc;
    return exchange_dense(&c, 1, advance, &dst, &dst_elems, width);
}

int rafthip_commit_allgather_dense_many(rafthip_comm** comms, int n, int advance, uint64_t** dst, const uint64_t* dst_elems,
                                        uint64_t* width) {
    int rc = check_comms(comms, n);
    if (rc) return rc;
    return exchange_dense(comms, n, advance, dst, dst_elems, width);
}

int rafthip_commit_allgather_many(rafthip_comm** comms, int n, int advance, rafthip_commit_pair** dst,
                                  const uint64_t* max_pairs, uint64_t** counts, uint64_t* n_total) {
    int rc = check_comms(comms, n);
    if (rc) return rc;
    return advance ? exchange_many<kAdvance>(comms, n, dst, max_pairs, counts, n_total)
                   : exchange_many<kChanged>(comms, n, dst, max_pairs, counts, n_total);
}

// ---- stateless one-shot forms ---------------------------------------------

// Host planes in, host result out, nothing resident between calls.  What bounds this form is the
// PCIe link (48 B per group at 5 voters), so the work is arranged around it (tools/pin_probe.hip, on
// the MI355X box: a pageable hipMemcpy moves 31 GB/s up and 18 GB/s down, page-locked memory 57 / 54,
// two host threads copy pageable -> page-locked at 80 GB/s, and hipHostRegister of caller memory is
// refused):
//   - a context (stream, events, page-locked staging slots) is taken from a process-wide pool and
//     returned after the call: no engine, no device allocation, no page-locking per call;
//   - the batch is cut into chunks; helper threads (kept with the context) copy the caller's planes
//     chunk by chunk into staging slots; the calling thread launches a chunk's kernel as soon as the
//     chunk is staged — the kernel reads its slot over the link itself (zero-copy: the transfer and
//     the quorum arithmetic are one pass) and writes its results into a page-locked result slot; a
//     third role, the retirer thread, copies finished result slots to the caller's buffer.  Up to
//     eight chunks are in flight, so the link never waits for the host.
// Device pointers take the resident path instead (an engine for the duration of the call).
}  // extern "C"

namespace {

constexpr int kShotSlots = 8;
constexpr size_t kShotChunkBytes = 4u << 20;  // staged input bytes per chunk
constexpr int kShotCopiers = 3;

struct ShotCtx {
    int device = -1;
    hipStream_t stream = nullptr;   // even chunks
    hipStream_t stream2 = nullptr;  // odd chunks: a chunk's kernel ramps up over the link while its predecessor drains
    hipEvent_t done[kShotSlots] = {};
    char* in[kShotSlots] = {};   // page-locked staging, host address
    char* out[kShotSlots] = {};
    char* d_in[kShotSlots] = {};  // the same memory as the device addresses it
    char* d_out[kShotSlots] = {};
    size_t in_cap = 0, out_cap = 0;
    void* d_zero = nullptr;  // device zeros: the outgoing half of a non-joint call
    size_t zero_cap = 0;
    // Staging helpers.  They live as long as the context (the pool keeps it until the process ends):
    // starting and joining two threads costs more than staging a 2^16-group batch.
    std::mutex mu;
    std::condition_variable wake, idle;
    std::vector<std::thread> helpers;
    const std::function<void()>* job = nullptr;  // what the helpers run; valid while `running` > 0 or unclaimed
    u64 generation = 0;
    int unclaimed = 0, running = 0;
};

// Run `job` on `n` helper threads of the context (started on first use); shot_helpers_wait returns when all
// of them have left it.
int shot_helpers_start(ShotCtx* c, int n, const std::function<void()>* job) {
    std::unique_lock<std::mutex> lock(c->mu);
    while ((int)c->helpers.size() < n) {
        try {
            c->helpers.emplace_back([c] {
                u64 seen = 0;
                std::unique_lock<std::mutex> l(c->mu);
                for (;;) {
                    c->wake.wait(l, [&] { return c->generation != seen && c->unclaimed > 0; });
                    seen = c->generation;
                    --c->unclaimed;
                    const std::function<void()>* j = c->job;
                    l.unlock();
                    (*j)();
                    l.lock();
                    if (--c->running == 0) c->idle.notify_all();
                }
            });
            c->helpers.back().detach();  // parked on the condition variable for the life of the process
        } catch (...) {
            return fail(RAFTHIP_ENOMEM, "cannot start a staging thread");
        }
    }
    c->job = job;
    ++c->generation;
    c->unclaimed = c->running = n;
    lock.unlock();
    c->wake.notify_all();
    return RAFTHIP_OK;
}

void shot_helpers_wait(ShotCtx* c) {
    std::unique_lock<std::mutex> lock(c->mu);
    c->idle.wait(lock, [&] { return c->running == 0; });
    c->job = nullptr;
}

std::mutex g_shot_mu;
std::vector<ShotCtx*> g_shot_pool;

// memcpy into a staging slot with streaming stores (dst 16-byte aligned): the next reader of these
// bytes is the GPU over PCIe, not this core, so they should not be left dirty in its cache.
void stage_copy(char* dst, const char* src, size_t n) {
    if (getenv("RAFTHIP_SHOT_PLAIN_COPY") || (reinterpret_cast<uintptr_t>(dst) & 15)) {
        memcpy(dst, src, n);
        return;
    }
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 16));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 32));
        const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 48));
        _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i), a);
        _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 32), c);
        _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 48), d);
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
    _mm_sfence();
}

void shot_release_buffers(ShotCtx* c) {
    for (int k = 0; k < kShotSlots; ++k) {
        if (c->in[k]) (void)hipHostFree(c->in[k]);
        if (c->out[k]) (void)hipHostFree(c->out[k]);
        c->in[k] = c->out[k] = c->d_in[k] = c->d_out[k] = nullptr;
    }
    c->in_cap = c->out_cap = 0;
}

// A context for `device` with staging slots of at least the given sizes (device current).
int shot_acquire(int device, size_t in_bytes, size_t out_bytes, size_t zero_bytes, ShotCtx** out) {
    ShotCtx* c = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_shot_mu);
        for (size_t i = 0; i < g_shot_pool.size(); ++i)
            if (g_shot_pool[i]->device == device) {
                c = g_shot_pool[i];
                g_shot_pool.erase(g_shot_pool.begin() + (long)i);
                break;
            }
    }
    if (!c) {
        c = new (std::nothrow) ShotCtx();
        if (!c) return fail(RAFTHIP_ENOMEM, "host allocation failed");
        c->device = device;
    }
    *out = c;  // from here on the caller returns it with shot_return, also on failure
    if (!c->stream) HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    if (!c->stream2) HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    for (int k = 0; k < kShotSlots; ++k)
        if (!c->done[k]) HIP_TRY(hipEventCreateWithFlags(&c->done[k], hipEventDisableTiming));
    if (c->in_cap < in_bytes || c->out_cap < out_bytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream2));
        shot_release_buffers(c);
        for (int k = 0; k < kShotSlots; ++k) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->in[k]), in_bytes, hipHostMallocDefault));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->out[k]), out_bytes, hipHostMallocDefault));
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_in[k]), c->in[k], 0));
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_out[k]), c->out[k], 0));
        }
        c->in_cap = in_bytes;
        c->out_cap = out_bytes;
    }
    if (c->zero_cap < zero_bytes) {
        if (c->d_zero) HIP_TRY(hipFree(c->d_zero));
        c->d_zero = nullptr;
        c->zero_cap = 0;
        HIP_TRY(hipMalloc(&c->d_zero, zero_bytes));
        HIP_TRY(hipMemsetAsync(c->d_zero, 0, zero_bytes, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // both streams read it
        c->zero_cap = zero_bytes;
    }
    return RAFTHIP_OK;
}

void shot_return(ShotCtx* c) {
    if (!c) return;
    std::lock_guard<std::mutex> lock(g_shot_mu);
    g_shot_pool.push_back(c);
}

// One-shot pass over host planes.  cell = bytes per plane element (8: Match, 1: vote cells),
// res = bytes per result (8 / 1).  launch(view) runs the kernel over a chunk-sized engine view.
template <typename Launch>
int shot_run(int device, u64 G, u32 S, size_t cell, size_t res, const void* planes, const void* in_mask,
             const void* out_mask, void* result, Launch launch) {
    const size_t mb = (size_t)mask_bytes_for(S);
    const size_t per_group = cell * S + 2 * mb;
    size_t chunk_bytes = kShotChunkBytes;
    int want_copiers = kShotCopiers;
    if (const char* v = getenv("RAFTHIP_SHOT_CHUNK_BYTES")) chunk_bytes = (size_t)strtoull(v, nullptr, 0);  // tuning knobs
    if (const char* v = getenv("RAFTHIP_SHOT_COPIERS")) want_copiers = atoi(v);
    if (want_copiers < 1) want_copiers = 1;
    if (want_copiers > 16) want_copiers = 16;
    u64 chunk = kPlaneAlign;  // groups per full chunk = slot capacity; a multiple of the plane granule
    while (chunk * 2 * per_group <= chunk_bytes && chunk < (1ull << 24)) chunk *= 2;
    if (chunk > round_up(G, kPlaneAlign)) chunk = round_up(G, kPlaneAlign);
    // Chunk table.  The first kernel cannot start before its chunk is staged, so the batch opens with
    // two quarter chunks and two half chunks and only then runs at full chunks.
    std::vector<u64> start;
    const bool ramp = G >= 4 * chunk;  // a batch of a few chunks is not worth the extra launches
    for (u64 g = 0, k = 0; g < G; ++k) {
        start.push_back(g);
        u64 len = !ramp ? chunk : k < 2 ? chunk / 4 : k < 4 ? chunk / 2 : chunk;
        if (len < kPlaneAlign) len = kPlaneAlign;
        g += len;
    }
    const u64 nchunks = start.size();
    start.push_back(G);
    const size_t planes_b = (size_t)chunk * cell * S, mask_b = (size_t)chunk * mb;
    DeviceGuard guard(device);
    if (!guard.ok) return fail(RAFTHIP_ENODEV, "hipSetDevice(%d) failed", device);
    ShotCtx* c = nullptr;
    int rc = shot_acquire(device, planes_b + 2 * mask_b, (size_t)chunk * res, out_mask ? 0 : mask_b, &c);
    if (rc) {
        shot_return(c);
        return rc;
    }
    // Three roles.  The calling thread launches: chunk k's kernel goes out as soon as the chunk is
    // staged, alternating between the two streams, so that several chunks are queued on the link at any
    // time.  Copier threads stage: chunk k goes to slot k % kShotSlots once chunk k - kShotSlots has
    // been retired.  One retirer thread waits for each chunk's event in order and copies its results to
    // the caller's buffer.  A batch of a few chunks does all three in line on the calling thread.
    std::atomic<u64> staged[kShotSlots];  // chunks staged into this slot so far
    std::atomic<u64> launched{0};         // chunks whose kernel and event are in their stream
    std::atomic<u64> retired{0};          // chunks whose results have left their slot
    std::atomic<u64> next_chunk{0};
    std::atomic<int> role{0};
    std::atomic<bool> abort{false};
    std::atomic<int> retire_error{(int)hipSuccess};
    for (auto& r : staged) r.store(0);
    // Waiting between the roles: a few hundred pause instructions (a chunk is ~25 us of link time, most hand-offs
    // arrive within that), then the thread sleeps on a condition variable until a counter moves — no runnable
    // thread burns a host core while it has nothing to do (RAFTHIP_SHOT_SPIN=1 restores the yield() loops of
    // round 2 for A/B runs, tools/oneshot_rate.py).  Every counter update is followed by signal().
    std::mutex pmu;
    std::condition_variable pcv;
    const bool spin_only = getenv("RAFTHIP_SHOT_SPIN") != nullptr;
    auto signal = [&] {
        if (spin_only) return;
        { std::lock_guard<std::mutex> l(pmu); }  // a waiter is either before its predicate check or already asleep
        pcv.notify_all();
    };
    auto wait_until = [&](auto&& ready) {  // returns when ready() or abort
        if (spin_only) {
            while (!ready() && !abort.load()) std::this_thread::yield();
            return;
        }
        for (int i = 0; i < 256; ++i) {
            if (ready() || abort.load()) return;
            _mm_pause();
        }
        std::unique_lock<std::mutex> l(pmu);
        pcv.wait(l, [&] { return ready() || abort.load(); });
    };
    const bool trace = getenv("RAFTHIP_SHOT_TRACE") != nullptr;
    double t_ready = 0, t_launch = 0, t_event = 0, t_out = 0;  // t_event / t_out: written by the one thread that retires
    auto tick = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = tick();
    auto stage_chunk = [&](u64 k) {
        const int slot = (int)(k % kShotSlots);
        wait_until([&] { return retired.load(std::memory_order_acquire) + kShotSlots > k; });
        if (abort.load()) return;
        const u64 g0 = start[k], cnt = start[k + 1] - g0;
        char* dst = c->in[slot];
        for (u32 s = 0; s < S; ++s)
            stage_copy(dst + (size_t)s * chunk * cell, static_cast<const char*>(planes) + ((size_t)s * G + g0) * cell, cnt * cell);
        stage_copy(dst + planes_b, static_cast<const char*>(in_mask) + g0 * mb, cnt * mb);
        if (out_mask) stage_copy(dst + planes_b + mask_b, static_cast<const char*>(out_mask) + g0 * mb, cnt * mb);
        staged[slot].fetch_add(1, std::memory_order_release);
        signal();
    };
    auto retire = [&](u64 k) -> hipError_t {
        const int slot = (int)(k % kShotSlots);
        const double t0 = tick();
        const hipError_t err = hipEventSynchronize(c->done[slot]);
        if (err != hipSuccess) return err;
        const double t1 = tick();
        const u64 g0 = start[k], cnt = start[k + 1] - g0;
        memcpy(static_cast<char*>(result) + g0 * res, c->out[slot], cnt * res);
        retired.store(k + 1, std::memory_order_release);
        signal();
        t_event += t1 - t0;
        t_out += tick() - t1;
        return hipSuccess;
    };
    auto launch_chunk = [&](u64 k) -> int {
        const int slot = (int)(k % kShotSlots);
        rafthip_engine view;  // a chunk-sized view over the staging slot: the kernels read it over the link
        view.device = device;
        view.G = start[k + 1] - start[k];
        view.S = S;
        view.stride = chunk;
        view.mbytes = (int)mb;
        view.stream = (k & 1) ? c->stream2 : c->stream;
        view.uniform = false;
        view.in_mask = c->d_in[slot] + planes_b;
        view.out_mask = out_mask ? static_cast<void*>(c->d_in[slot] + planes_b + mask_b) : c->d_zero;
        int r = launch(view, c->d_in[slot], c->d_out[slot]);
        if (!r) {
            hipError_t err = hipEventRecord(c->done[slot], view.stream);
            if (err != hipSuccess) r = fail(RAFTHIP_EIO, "hipEventRecord: %s", hipGetErrorString(err));
        }
        return r;
    };
    int nhelpers = nchunks >= 4 ? want_copiers + 1 : 0;
    const std::function<void()> helper = [&] {
        if (role.fetch_add(1) == 0) {  // the retirer
            (void)hipSetDevice(device);
            for (u64 k = 0; k < nchunks; ++k) {
                wait_until([&] { return launched.load(std::memory_order_acquire) > k; });
                if (abort.load()) return;
                const hipError_t err = retire(k);
                if (err != hipSuccess) {
                    retire_error.store((int)err);
                    abort.store(true);
                    signal();
                    return;
                }
            }
            return;
        }
        for (;;) {  // a copier
            const u64 k = next_chunk.fetch_add(1);
            if (k >= nchunks || abort.load()) return;
            stage_chunk(k);
        }
    };
    next_chunk.store(1);  // the calling thread stages chunk 0 itself while the helpers wake up
    if (nhelpers && shot_helpers_start(c, nhelpers, &helper)) nhelpers = 0;  // no helper threads: everything in line
    for (u64 k = 0; k < nchunks && !rc; ++k) {
        const int slot = (int)(k % kShotSlots);
        const double t0 = tick();
        if (!nhelpers) {
            if (k >= (u64)kShotSlots) {
                const hipError_t err = retire(k - kShotSlots);
                if (err != hipSuccess) rc = fail(RAFTHIP_EIO, "hipEventSynchronize: %s", hipGetErrorString(err));
            }
            if (!rc) stage_chunk(k);
        } else if (k == 0) {
            stage_chunk(0);
        }
        if (!rc) wait_until([&] { return staged[slot].load(std::memory_order_acquire) >= k / kShotSlots + 1; });
        if (rc || abort.load()) break;
        const double t1 = tick();
        t_ready += t1 - t0;
        rc = launch_chunk(k);
        if (!rc) {
            launched.store(k + 1, std::memory_order_release);
            signal();
        }
        t_launch += tick() - t1;
    }
    if (rc) {
        abort.store(true);
        signal();
    }
    if (nhelpers) {
        shot_helpers_wait(c);
        if (!rc && retire_error.load() != (int)hipSuccess)
            rc = fail(RAFTHIP_EIO, "hipEventSynchronize: %s", hipGetErrorString((hipError_t)retire_error.load()));
    } else {
        for (u64 k = nchunks > (u64)kShotSlots ? nchunks - kShotSlots : 0; k < nchunks && !rc; ++k) {
            const hipError_t err = retire(k);
            if (err != hipSuccess) rc = fail(RAFTHIP_EIO, "hipEventSynchronize: %s", hipGetErrorString(err));
        }
    }
    if (trace)
        fprintf(stderr, "shot: G %llu chunks %llu total %.0f us: wait-staged %.0f launch %.0f | retirer: wait-gpu %.0f copy-out %.0f\n",
                (u64)G, (u64)nchunks, (tick() - t_begin) * 1e6, t_ready * 1e6, t_launch * 1e6, t_event * 1e6, t_out * 1e6);
    if (rc) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(c->stream2);
    }
    shot_return(c);
    return rc;
}

bool host_pointers(std::initializer_list<const void*> ps) {
    for (const void* p : ps)
        if (p && is_device_ptr(p)) return false;
    return true;
}

}  // namespace

extern "C" {

int rafthip_batch_committed_index(int device, uint64_t num_groups, uint32_t num_slots,
                                  const uint64_t* match, const void* in_mask, const void* out_mask,
                                  uint64_t* out) {
    if (!match || !out) return fail(RAFTHIP_EINVAL, "match/out is NULL");
    if (!in_mask) return fail(RAFTHIP_EINVAL, "in_mask is NULL");
    if (num_groups == 0) return fail(RAFTHIP_EINVAL, "num_groups must be >= 1");
    if (num_slots == 0 || num_slots > RAFTHIP_MAX_SLOTS)
        return fail(RAFTHIP_EINVAL, "num_slots %u outside [1, %u]", num_slots, RAFTHIP_MAX_SLOTS);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(RAFTHIP_ENODEV, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(RAFTHIP_ENODEV, "device %d outside [0, %d)", device, ndev);
    if (host_pointers({match, in_mask, out_mask, out}))
        return shot_run(device, num_groups, num_slots, 8, 8, match, in_mask, out_mask, out,
                        [](rafthip_engine& view, char* d_in, char* d_out) {
                            view.match = reinterpret_cast<u64*>(d_in);
                            view.commit = reinterpret_cast<u64*>(d_out);
                            u32 bg = 0;
                            return launch_commit<kStore>(&view, &bg);
                        });
    rafthip_engine* e = nullptr;
    int rc = oneshot_engine(device, num_groups, num_slots, in_mask, out_mask, &e);
    if (rc) return rc;
    rc = rafthip_upload_match(e, 0, num_groups, match, num_groups);
    if (!rc) rc = rafthip_commit(e);
    if (!rc) rc = rafthip_read_commit(e, 0, num_groups, out);
    char keep[sizeof(g_err)];
    memcpy(keep, g_err, sizeof(keep));
    rafthip_destroy(e);
    memcpy(g_err, keep, sizeof(keep));
    return rc;
}

int rafthip_batch_vote_result(int device, uint64_t num_groups, uint32_t num_slots,
                              const uint8_t* votes, const void* in_mask, const void* out_mask,
                              uint8_t* out) {
    if (!votes || !out) return fail(RAFTHIP_EINVAL, "votes/out is NULL");
    if (!in_mask) return fail(RAFTHIP_EINVAL, "in_mask is NULL");
    if (num_groups == 0) return fail(RAFTHIP_EINVAL, "num_groups must be >= 1");
    if (num_slots == 0 || num_slots > RAFTHIP_MAX_SLOTS)
        return fail(RAFTHIP_EINVAL, "num_slots %u outside [1, %u]", num_slots, RAFTHIP_MAX_SLOTS);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(RAFTHIP_ENODEV, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(RAFTHIP_ENODEV, "device %d outside [0, %d)", device, ndev);
    if (host_pointers({votes, in_mask, out_mask, out}))
        return shot_run(device, num_groups, num_slots, 1, 1, votes, in_mask, out_mask, out,
                        [](rafthip_engine& view, char* d_in, char* d_out) {
                            view.votes = reinterpret_cast<uint8_t*>(d_in);
                            view.vote_result = reinterpret_cast<uint8_t*>(d_out);
                            return launch_tally_votes(&view);  // granted / rejected NULL: not computed
                        });
    rafthip_engine* e = nullptr;
    int rc = oneshot_engine(device, num_groups, num_slots, in_mask, out_mask, &e);
    if (rc) return rc;
    rc = rafthip_upload_votes(e, 0, num_groups, votes, num_groups);
    if (!rc) rc = rafthip_tally_votes(e);
    if (!rc) rc = rafthip_read_vote_result(e, 0, num_groups, out, nullptr, nullptr);
    char keep[sizeof(g_err)];
    memcpy(keep, g_err, sizeof(keep));
    rafthip_destroy(e);
    memcpy(g_err, keep, sizeof(keep));
    return rc;
}

}  // extern "C"
