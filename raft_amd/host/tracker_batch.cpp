// tracker_batch.cpp — BatchProgressTracker: the id-keyed bookkeeping of
// tracker.ProgressTracker (tracker/tracker.go:117-290) for many groups, above
// one librafthip.so engine.  No quorum arithmetic happens here: this file owns
// the id -> slot maps and host shadows of the small planes, and calls the GPU.
#include "tracker_batch.hpp"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>

namespace raft {
namespace tracker {

int BatchProgressTracker::fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_, sizeof(err_), fmt, ap);
    va_end(ap);
    return code;
}

int BatchProgressTracker::lib(int rc) {
    if (rc) snprintf(err_, sizeof(err_), "%s", rafthip_last_error());
    return rc;
}

int BatchProgressTracker::Make(int device, uint64_t groups, uint32_t max_slots, BatchProgressTracker** out) {
    if (!out) return RAFTHIP_EINVAL;
    *out = nullptr;
    BatchProgressTracker* t = new (std::nothrow) BatchProgressTracker();
    if (!t) return RAFTHIP_ENOMEM;
    rafthip_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.device = device;
    cfg.num_groups = groups;
    cfg.num_slots = max_slots;
    int rc = rafthip_create(&cfg, &t->e_);
    if (rc) {
        delete t;
        return rc;  // text in rafthip_last_error()
    }
    t->G_ = groups;
    t->S_ = max_slots;
    t->mb_ = (uint32_t)rafthip_mask_bytes(t->e_);
    t->ids_.assign(groups * max_slots, 0);
    t->used_.assign(groups, 0);
    t->in_.assign(groups * t->mb_, 0);
    t->out_.assign(groups * t->mb_, 0);
    t->learner_.assign(groups * t->mb_, 0);
    t->votes_.assign((size_t)max_slots * groups, RAFTHIP_CELL_NONE);
    t->active_.assign(groups * t->mb_, 0);
    t->rdacks_.assign(groups * t->mb_, 0);
    t->self_.assign(groups * t->mb_, 0);
    t->pairs_.resize(groups);
    // the engine must see zeroed bit planes even if no bit is ever set
    t->touch(t->active_dirty_, 0);
    t->touch(t->active_dirty_, groups - 1);
    t->touch(t->rdacks_dirty_, 0);
    t->touch(t->rdacks_dirty_, groups - 1);
    t->touch(t->votes_dirty_, 0);
    t->touch(t->votes_dirty_, groups - 1);
    *out = t;
    return RAFTHIP_OK;
}

BatchProgressTracker::~BatchProgressTracker() {
    if (e_) rafthip_destroy(e_);
}

uint64_t BatchProgressTracker::get_mask(const std::vector<uint8_t>& plane, uint64_t g) const {
    uint64_t m = 0;
    memcpy(&m, plane.data() + g * mb_, mb_);  // little-endian host
    return m;
}

void BatchProgressTracker::put_mask(std::vector<uint8_t>& plane, uint64_t g, uint64_t m) {
    memcpy(plane.data() + g * mb_, &m, mb_);
}

void BatchProgressTracker::touch(uint64_t (&range)[2], uint64_t g) {
    if (range[0] > range[1]) {
        range[0] = range[1] = g;
    } else {
        if (g < range[0]) range[0] = g;
        if (g > range[1]) range[1] = g;
    }
}

int BatchProgressTracker::slot_of(uint64_t group, uint64_t id) const {
    const uint64_t used = used_[group];
    const uint64_t* ids = ids_.data() + group * S_;
    for (uint32_t s = 0; s < S_; ++s)
        if (((used >> s) & 1ull) && ids[s] == id) return (int)s;
    return -1;
}

int BatchProgressTracker::push_config(uint64_t group) {
    return lib(rafthip_upload_config(e_, group, 1, in_.data() + group * mb_, out_.data() + group * mb_,
                                     learner_.data() + group * mb_));
}

int BatchProgressTracker::SetConfig(uint64_t group, const quorum::MajorityConfig& incoming,
                                    const quorum::MajorityConfig& outgoing,
                                    const quorum::MajorityConfig& learners, uint64_t last_index) {
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    quorum::MajorityConfig all = incoming;
    all.insert(outgoing.begin(), outgoing.end());
    for (uint64_t id : learners) {
        // Learners and voters never intersect — confchange/confchange.go:305-316
        if (all.count(id)) return fail(RAFTHIP_EINVAL, "id %llu is both a voter and a learner", (unsigned long long)id);
    }
    all.insert(learners.begin(), learners.end());
    if (all.size() > S_)
        return fail(RAFTHIP_EINVAL, "group %llu needs %zu Progress rows; the tracker was made with %u slots",
                    (unsigned long long)group, all.size(), S_);
    int rc = Flush();  // queued acks were addressed with the old id -> slot map
    if (rc) return rc;

    uint64_t* ids = ids_.data() + group * S_;
    uint64_t used = used_[group];
    uint64_t active = get_mask(active_, group), rdacks = get_mask(rdacks_, group);
    std::vector<uint64_t> row;  // the group's Match cells, fetched only if a slot changes hands
    bool row_dirty = false;
    auto fetch_row = [&]() -> int {
        if (!row.empty()) return RAFTHIP_OK;
        row.assign(S_, 0);
        return lib(rafthip_read_match(e_, group, 1, row.data(), 1));
    };
    // Progress of ids that left the config is deleted (confchange.go:233-247).
    for (uint32_t s = 0; s < S_; ++s) {
        if (!((used >> s) & 1ull) || all.count(ids[s])) continue;
        if ((rc = fetch_row())) return rc;
        row[s] = 0;
        row_dirty = true;
        uint8_t& cell = votes_[(size_t)s * G_ + group];
        if (cell != RAFTHIP_CELL_NONE) {  // Votes is keyed by id and survives the conf change (tracker.go:122)
            stray_votes_[group][ids[s]] = (cell == RAFTHIP_CELL_GRANT);
            cell = RAFTHIP_CELL_NONE;
            touch(votes_dirty_, group);
        }
        if ((rdacks >> s) & 1ull) stray_acks_[group].insert(ids[s]);  // acks are keyed by id too (read_only.go:29-37)
        if ((get_mask(self_, group) >> s) & 1ull) {  // the leader itself left the config
            put_mask(self_, group, 0);
            touch(self_dirty_, group);
        }
        used &= ~(1ull << s);
        active &= ~(1ull << s);
        rdacks &= ~(1ull << s);
        ids[s] = 0;
    }
    // New ids get a fresh Progress: Match 0, RecentActive true (confchange.go:257-270).
    std::vector<uint32_t> fresh;  // slots that changed hands: their per-peer state must not survive
    for (uint64_t id : all) {
        bool have = false;
        for (uint32_t s = 0; s < S_; ++s) have |= ((used >> s) & 1ull) && ids[s] == id;
        if (have) continue;
        uint32_t s = 0;
        while ((used >> s) & 1ull) ++s;
        ids[s] = id;
        used |= 1ull << s;
        active |= 1ull << s;
        fresh.push_back(s);
        auto sv = stray_votes_.find(group);
        if (sv != stray_votes_.end()) {
            auto it = sv->second.find(id);
            if (it != sv->second.end()) {
                votes_[(size_t)s * G_ + group] = it->second ? RAFTHIP_CELL_GRANT : RAFTHIP_CELL_REJECT;
                touch(votes_dirty_, group);
                sv->second.erase(it);
            }
        }
        auto sa = stray_acks_.find(group);
        if (sa != stray_acks_.end() && sa->second.erase(id)) rdacks |= 1ull << s;
    }
    used_[group] = used;
    uint64_t in = 0, out = 0, lrn = 0;
    for (uint32_t s = 0; s < S_; ++s) {
        if (!((used >> s) & 1ull)) continue;
        if (incoming.count(ids[s])) in |= 1ull << s;
        if (outgoing.count(ids[s])) out |= 1ull << s;
        if (learners.count(ids[s])) lrn |= 1ull << s;
    }
    put_mask(in_, group, in);
    put_mask(out_, group, out);
    put_mask(learner_, group, lrn);
    put_mask(active_, group, active);
    put_mask(rdacks_, group, rdacks);
    touch(active_dirty_, group);
    touch(rdacks_dirty_, group);
    if (row_dirty && (rc = lib(rafthip_upload_match(e_, group, 1, row.data(), 1)))) return rc;
    if (progress_) {
        // The slot may have held another peer: {StateProbe, Next = LastIndex, empty Inflights} for the new one
        // only (Changer.initProgress, confchange.go:257-270 — `Next: c.LastIndex`, so a peer added at LastIndex 0
        // starts with Next 0, as confchange/testdata/*.txt print it); the surviving peers keep theirs.
        for (uint32_t s : fresh)
            if ((rc = lib(rafthip_init_progress_row(e_, group, s, RAFTHIP_STATE_PROBE, last_index, 0))))
                return rc;
    }
    return push_config(group);
}

int BatchProgressTracker::ResetProgress(uint64_t group) {
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    int rc = Flush();
    if (rc) return rc;
    if ((rc = lib(rafthip_reset_match(e_, group, 1)))) return rc;
    put_mask(active_, group, 0);
    touch(active_dirty_, group);
    return ResetVotes(group);
}

bool BatchProgressTracker::MaybeUpdate(uint64_t group, uint64_t id, uint64_t n) {
    if (group >= G_) return false;
    const int s = slot_of(group, id);
    if (s < 0) return false;
    rafthip_ack a;
    a.group = (uint32_t)group;
    a.slot = (uint32_t)s;
    a.index = n;
    acks_.push_back(a);
    return true;
}

int BatchProgressTracker::Flush() {
    int rc = RAFTHIP_OK;
    if (!acks_.empty()) {
        rc = lib(rafthip_scatter_acks(e_, acks_.data(), acks_.size()));
        acks_.clear();
        if (rc) return rc;
    }
    if (!msgs_.empty()) {
        if (progress_) return FlushProgress(nullptr);
        rc = lib(rafthip_ingest_msgs(e_, msgs_.data(), msgs_.size()));
        msgs_.clear();
    }
    return rc;
}

int BatchProgressTracker::Committed(std::vector<CommitPair>* changed) {
    int rc = Flush();
    if (rc) return rc;
    static_assert(sizeof(CommitPair) == sizeof(rafthip_commit_pair), "pair layout");
    uint64_t n = 0;
    rc = lib(rafthip_commit_changed(e_, reinterpret_cast<rafthip_commit_pair*>(pairs_.data()), pairs_.size(), &n));
    if (rc) return rc;
    if (changed) changed->assign(pairs_.begin(), pairs_.begin() + (n < pairs_.size() ? n : pairs_.size()));
    return RAFTHIP_OK;
}

int BatchProgressTracker::CommittedOf(uint64_t group, uint64_t* commit) {
    if (!commit) return fail(RAFTHIP_EINVAL, "commit is NULL");
    return lib(rafthip_read_commit(e_, group, 1, commit));
}

int BatchProgressTracker::Match(uint64_t group, uint64_t id, uint64_t* match) {
    if (!match) return fail(RAFTHIP_EINVAL, "match is NULL");
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    const int s = slot_of(group, id);
    if (s < 0) return fail(RAFTHIP_EINVAL, "group %llu has no Progress for id %llu", (unsigned long long)group,
                           (unsigned long long)id);
    int rc = Flush();
    if (rc) return rc;
    std::vector<uint64_t> row(S_, 0);
    if ((rc = lib(rafthip_read_match(e_, group, 1, row.data(), 1)))) return rc;
    *match = row[s];
    return RAFTHIP_OK;
}

int BatchProgressTracker::ResetVotes(uint64_t group) {
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    for (uint32_t s = 0; s < S_; ++s) votes_[(size_t)s * G_ + group] = RAFTHIP_CELL_NONE;
    stray_votes_.erase(group);
    touch(votes_dirty_, group);
    return RAFTHIP_OK;
}

int BatchProgressTracker::RecordVote(uint64_t group, uint64_t id, bool v) {
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    const int s = slot_of(group, id);
    if (s < 0) {
        // recorded all the same (tracker.go:260-265); it counts only if the id
        // later becomes a voter before the next ResetVotes
        stray_votes_[group].emplace(id, v);
        return RAFTHIP_OK;
    }
    uint8_t& cell = votes_[(size_t)s * G_ + group];
    if (cell == RAFTHIP_CELL_NONE) {  // the first vote of an id sticks
        cell = v ? RAFTHIP_CELL_GRANT : RAFTHIP_CELL_REJECT;
        touch(votes_dirty_, group);
    }
    return RAFTHIP_OK;
}

int BatchProgressTracker::TallyVotes() {
    if (votes_dirty_[0] <= votes_dirty_[1]) {
        const uint64_t first = votes_dirty_[0], count = votes_dirty_[1] - votes_dirty_[0] + 1;
        int rc = lib(rafthip_upload_votes(e_, first, count, votes_.data() + first, G_));
        if (rc) return rc;
        votes_dirty_[0] = 1;
        votes_dirty_[1] = 0;
    }
    return lib(rafthip_tally_votes(e_));
}

int BatchProgressTracker::VoteResultOf(uint64_t group, int* granted, int* rejected, quorum::VoteResult* result) {
    uint8_t r = 0, g = 0, j = 0;
    int rc = lib(rafthip_read_vote_result(e_, group, 1, &r, &g, &j));
    if (rc) return rc;
    if (granted) *granted = g;
    if (rejected) *rejected = j;
    if (result) *result = (quorum::VoteResult)r;
    return RAFTHIP_OK;
}

int BatchProgressTracker::SetRecentActive(uint64_t group, uint64_t id, bool active) {
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    const int s = slot_of(group, id);
    if (s < 0) return fail(RAFTHIP_EINVAL, "group %llu has no Progress for id %llu", (unsigned long long)group,
                           (unsigned long long)id);
    uint64_t m = get_mask(active_, group);
    m = active ? (m | (1ull << s)) : (m & ~(1ull << s));
    put_mask(active_, group, m);
    touch(active_dirty_, group);
    return RAFTHIP_OK;
}

int BatchProgressTracker::QuorumActive(std::vector<uint8_t>* active) {
    if (!active) return fail(RAFTHIP_EINVAL, "active is NULL");
    {
        // queued responses carry the same bit updates the shadow already has;
        // apply them before the shadow overwrites the device plane
        int frc = Flush();
        if (frc) return frc;
    }
    if (active_dirty_[0] <= active_dirty_[1]) {
        const uint64_t first = active_dirty_[0], count = active_dirty_[1] - active_dirty_[0] + 1;
        int rc = lib(rafthip_upload_active(e_, first, count, active_.data() + first * mb_));
        if (rc) return rc;
        active_dirty_[0] = 1;
        active_dirty_[1] = 0;
    }
    int rc = lib(rafthip_quorum_active(e_));
    if (rc) return rc;
    active->assign(G_, 0);
    return lib(rafthip_read_quorum_active(e_, 0, G_, active->data()));
}

int BatchProgressTracker::RecvAck(uint64_t group, uint64_t id) {
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    const int s = slot_of(group, id);
    if (s < 0) {
        // kept all the same (read_only.go:74); it counts only if the id becomes
        // a voter while the request is still pending
        stray_acks_[group].insert(id);
        return RAFTHIP_OK;
    }
    put_mask(rdacks_, group, get_mask(rdacks_, group) | (1ull << s));
    touch(rdacks_dirty_, group);
    return RAFTHIP_OK;
}

int BatchProgressTracker::ResetAcks(uint64_t group) {
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    put_mask(rdacks_, group, 0);
    stray_acks_.erase(group);
    touch(rdacks_dirty_, group);
    return RAFTHIP_OK;
}

int BatchProgressTracker::AckQuorum(std::vector<quorum::VoteResult>* result) {
    if (!result) return fail(RAFTHIP_EINVAL, "result is NULL");
    {
        int frc = Flush();
        if (frc) return frc;
    }
    if (rdacks_dirty_[0] <= rdacks_dirty_[1]) {
        const uint64_t first = rdacks_dirty_[0], count = rdacks_dirty_[1] - rdacks_dirty_[0] + 1;
        int rc = lib(rafthip_upload_acks(e_, first, count, rdacks_.data() + first * mb_));
        if (rc) return rc;
        rdacks_dirty_[0] = 1;
        rdacks_dirty_[1] = 0;
    }
    int rc = lib(rafthip_ack_quorum(e_));
    if (rc) return rc;
    result->assign(G_, quorum::VotePending);
    static_assert(sizeof(quorum::VoteResult) == 1, "VoteResult is uint8 (quorum/quorum.go:48)");
    return lib(rafthip_read_ack_quorum(e_, 0, G_, reinterpret_cast<uint8_t*>(result->data())));
}

// ---- the leader-side tick ----------------------------------------------------

int BatchProgressTracker::SetLeader(uint64_t group, uint64_t self, uint64_t term_first, uint64_t last_index) {
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    int rc = Flush();
    if (rc) return rc;
    if ((rc = lib(rafthip_upload_term_range(e_, group, 1, &term_first, &last_index)))) return rc;
    const int s = slot_of(group, self);  // a leader removed from the config has no Progress row
    put_mask(self_, group, s < 0 ? 0 : (1ull << s));
    touch(self_dirty_, group);
    if (s >= 0) {
        put_mask(active_, group, get_mask(active_, group) | (1ull << s));
        touch(active_dirty_, group);
    }
    return RAFTHIP_OK;
}

int BatchProgressTracker::SetFollower(uint64_t group) {
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    int rc = Flush();
    if (rc) return rc;
    const uint64_t first = 1, last = 0;  // empty range
    if ((rc = lib(rafthip_upload_term_range(e_, group, 1, &first, &last)))) return rc;
    put_mask(self_, group, 0);
    touch(self_dirty_, group);
    return RAFTHIP_OK;
}

int BatchProgressTracker::SetCommitted(uint64_t group, uint64_t committed) {
    int rc = Flush();
    if (rc) return rc;
    return lib(rafthip_upload_commit(e_, group, 1, &committed));
}

bool BatchProgressTracker::Appended(uint64_t group, uint64_t last_index) {
    if (group >= G_) return false;
    rafthip_msg m;
    memset(&m, 0, sizeof(m));
    m.group = (uint32_t)group;
    m.type = RAFTHIP_MSG_APPENDED;
    m.index = last_index;
    msgs_.push_back(m);
    if (progress_) aux_.push_back(0);
    return true;
}

bool BatchProgressTracker::Step(uint64_t group, uint64_t from, int type, uint64_t index, bool reject,
                                bool read_ctx) {
    if (group >= G_) return false;
    if (type != RAFTHIP_MSG_APP_RESP && type != RAFTHIP_MSG_HEARTBEAT_RESP) return false;
    const int s = slot_of(group, from);
    if (s < 0) return false;
    // The bit planes are host-shadowed (uploaded lazily before a tally), so the
    // shadow takes the same update the kernel applies on the device.
    put_mask(active_, group, get_mask(active_, group) | (1ull << s));
    touch(active_dirty_, group);
    if (type == RAFTHIP_MSG_HEARTBEAT_RESP && read_ctx) {
        put_mask(rdacks_, group, get_mask(rdacks_, group) | (1ull << s));
        touch(rdacks_dirty_, group);
    }
    rafthip_msg m;
    memset(&m, 0, sizeof(m));
    m.group = (uint32_t)group;
    m.slot = (uint8_t)s;
    m.type = (uint8_t)type;
    m.reject = reject ? 1 : 0;
    m.flags = read_ctx ? RAFTHIP_MSGF_READ_CTX : 0;
    m.index = index;
    msgs_.push_back(m);
    if (progress_) aux_.push_back(0);
    return true;
}

int BatchProgressTracker::MaybeCommit(std::vector<CommitPair>* advanced) {
    int rc = Flush();
    if (rc) return rc;
    uint64_t n = 0;
    rc = lib(rafthip_commit_advance(e_, reinterpret_cast<rafthip_commit_pair*>(pairs_.data()), pairs_.size(), &n));
    if (rc) return rc;
    if (advanced) advanced->assign(pairs_.begin(), pairs_.begin() + (n < pairs_.size() ? n : pairs_.size()));
    return RAFTHIP_OK;
}

int BatchProgressTracker::SyncActiveAndSelf() {
    int rc;
    if (active_dirty_[0] <= active_dirty_[1]) {
        const uint64_t first = active_dirty_[0], count = active_dirty_[1] - active_dirty_[0] + 1;
        if ((rc = lib(rafthip_upload_active(e_, first, count, active_.data() + first * mb_)))) return rc;
        active_dirty_[0] = 1;
        active_dirty_[1] = 0;
    }
    if (self_dirty_[0] <= self_dirty_[1]) {
        const uint64_t first = self_dirty_[0], count = self_dirty_[1] - self_dirty_[0] + 1;
        if ((rc = lib(rafthip_upload_self(e_, first, count, self_.data() + first * mb_)))) return rc;
        self_dirty_[0] = 1;
        self_dirty_[1] = 0;
    }
    return RAFTHIP_OK;
}

int BatchProgressTracker::SetTickConfig(uint32_t election_tick, uint32_t heartbeat_tick, bool check_quorum) {
    return lib(rafthip_tick_config(e_, election_tick, heartbeat_tick, check_quorum ? 1 : 0));
}

int BatchProgressTracker::SetTickState(uint64_t group, uint32_t election_elapsed, uint32_t heartbeat_elapsed,
                                       uint32_t randomized_timeout, int state, bool promotable, bool transferring) {
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    if (state < 0 || state > 3) return fail(RAFTHIP_EINVAL, "state %d is no StateType", state);
    const uint8_t role = (uint8_t)state;
    const uint8_t flags = (uint8_t)((promotable ? RAFTHIP_TICKF_PROMOTABLE : 0u) | (transferring ? RAFTHIP_TICKF_TRANSFER : 0u));
    return lib(rafthip_upload_tick_state(e_, group, 1, &election_elapsed, &heartbeat_elapsed, &randomized_timeout, &role, &flags));
}

int BatchProgressTracker::HeardFromLeader(const std::vector<uint32_t>& groups) {
    uint64_t outside = 0;
    int rc = lib(rafthip_tick_heard_from_leader(e_, groups.data(), groups.size(), &outside));
    if (rc) return rc;
    if (outside) return fail(RAFTHIP_ERANGE, "%llu group ids outside [0, %llu)", (unsigned long long)outside, (unsigned long long)G_);
    return RAFTHIP_OK;
}

int BatchProgressTracker::Tick(uint32_t event_mask, std::vector<TickEvent>* events, uint64_t counts[5]) {
    int rc = Flush();  // queued responses set RecentActive before the tick looks at it
    if (rc) return rc;
    if ((rc = SyncActiveAndSelf())) return rc;
    uint64_t c[RAFTHIP_TICK_ACTIONS] = {0, 0, 0, 0, 0};
    if ((rc = lib(rafthip_tick(e_, c)))) return rc;
    if (counts) memcpy(counts, c, sizeof(c));
    // the groups that ran MsgCheckQuorum: their RecentActive was reset on the device, the shadow follows
    // (raft.go:1236-1242; after a step-down every bit is clear and the group is not led here any more)
    const uint32_t mask = (event_mask | RAFTHIP_TICK_CHECK_QUORUM) & 0xFFu;
    uint64_t cap = 0;
    for (int b = 0; b < RAFTHIP_TICK_ACTIONS; ++b)
        if ((mask >> b) & 1u) cap += c[b];
    if (cap > G_) cap = G_;
    std::vector<rafthip_tick_event> ev(cap);
    uint64_t n = 0;
    if ((rc = lib(rafthip_read_tick_events(e_, mask, ev.data(), cap, &n)))) return rc;
    if (n > cap) return fail(RAFTHIP_EIO, "tick events: %llu listed, at most %llu counted", (unsigned long long)n,
                             (unsigned long long)cap);
    if (events) events->clear();
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t g = ev[i].group;
        if (ev[i].action & RAFTHIP_TICK_STEP_DOWN) {
            put_mask(active_, g, 0);
            put_mask(self_, g, 0);
            touch(self_dirty_, g);  // the engine keeps the self mask: cleared with the next sync
        } else if (ev[i].action & RAFTHIP_TICK_CHECK_QUORUM) {
            put_mask(active_, g, get_mask(active_, g) & get_mask(self_, g));
        }
        if (events && (ev[i].action & event_mask)) events->push_back(TickEvent{ev[i].group, ev[i].action});
    }
    return RAFTHIP_OK;
}

int BatchProgressTracker::CheckQuorum(std::vector<uint8_t>* active) {
    if (!active) return fail(RAFTHIP_EINVAL, "active is NULL");
    int rc = Flush();  // queued responses set RecentActive on the device too; same bits as the shadow
    if (rc) return rc;
    if ((rc = SyncActiveAndSelf())) return rc;
    if ((rc = lib(rafthip_check_quorum(e_)))) return rc;
    active->assign(G_, 0);
    if ((rc = lib(rafthip_read_quorum_active(e_, 0, G_, active->data())))) return rc;
    // the engine cleared every RecentActive but self; the shadow follows
    for (size_t i = 0; i < active_.size(); ++i) active_[i] &= self_[i];
    return RAFTHIP_OK;
}

int BatchProgressTracker::HeartbeatCommit(uint64_t group, uint64_t id, uint64_t* commit) {
    if (!commit) return fail(RAFTHIP_EINVAL, "commit is NULL");
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    const int s = slot_of(group, id);
    if (s < 0) return fail(RAFTHIP_EINVAL, "group %llu has no Progress for id %llu", (unsigned long long)group,
                           (unsigned long long)id);
    int rc = Flush();
    if (rc) return rc;
    std::vector<uint64_t> row(S_, 0);
    if ((rc = lib(rafthip_heartbeat_commit(e_, group, 1, row.data(), 1)))) return rc;
    *commit = row[s];
    return RAFTHIP_OK;
}

// ---- per-peer replication state ------------------------------------------------

int BatchProgressTracker::BcastHeartbeat(uint64_t group, uint64_t term, const std::string& ctx,
                                         std::map<uint64_t, std::string>* msgs) {
    if (!msgs) return fail(RAFTHIP_EINVAL, "msgs is NULL");
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    if (ctx.size() > 8) return fail(RAFTHIP_EINVAL, "a ReadIndex context of %zu bytes: the engine tracks at most 8", ctx.size());
    int rc = Flush();
    if (rc) return rc;
    // the lookup state of the wire path for this group: ids behind the rows, r.Term, the pending context
    if ((rc = lib(rafthip_upload_ids(e_, group, 1, ids_.data() + group * S_, 1)))) return rc;
    if ((rc = lib(rafthip_upload_term(e_, group, 1, &term)))) return rc;
    uint64_t word = 0;
    for (size_t i = 0; i < ctx.size(); ++i) word |= (uint64_t)(uint8_t)ctx[i] << (8 * i);
    const uint8_t len = (uint8_t)ctx.size();
    if ((rc = lib(rafthip_upload_read_ctx(e_, group, 1, &word, &len)))) return rc;
    if ((rc = lib(rafthip_upload_self(e_, group, 1, self_.data() + group * mb_)))) return rc;
    std::vector<uint8_t> bytes((size_t)S_ * RAFTHIP_HEARTBEAT_STRIDE), lens(S_);
    if ((rc = lib(rafthip_emit_heartbeats(e_, group, 1, bytes.data(), lens.data())))) return rc;
    msgs->clear();
    const uint64_t* ids = ids_.data() + group * S_;
    for (uint32_t s = 0; s < S_; ++s)
        if (lens[s])
            (*msgs)[ids[s]] = std::string(reinterpret_cast<const char*>(bytes.data()) + (size_t)s * RAFTHIP_HEARTBEAT_STRIDE, lens[s]);
    return RAFTHIP_OK;
}

int BatchProgressTracker::PlanAppends(uint64_t group, bool send_if_empty, std::map<uint64_t, SendPlan>* plan) {
    if (!plan) return fail(RAFTHIP_EINVAL, "plan is NULL");
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    if (!progress_) return fail(RAFTHIP_EINVAL, "EnableProgress must come first");
    int rc = Flush();
    if (rc) return rc;
    // RecentActive and the leader's slot as the shadows hold them (the device copies are refreshed lazily; queued
    // responses set RecentActive on the device too, same bits as the shadow)
    if (active_dirty_[0] <= active_dirty_[1]) {
        const uint64_t first = active_dirty_[0], count = active_dirty_[1] - active_dirty_[0] + 1;
        if ((rc = lib(rafthip_upload_active(e_, first, count, active_.data() + first * mb_)))) return rc;
        active_dirty_[0] = 1;
        active_dirty_[1] = 0;
    }
    if ((rc = lib(rafthip_upload_self(e_, group, 1, self_.data() + group * mb_)))) return rc;
    std::vector<uint8_t> kinds(S_);
    std::vector<uint64_t> next(S_);
    if ((rc = lib(rafthip_plan_appends(e_, group, 1, send_if_empty ? 1 : 0, kinds.data(), next.data(), 1)))) return rc;
    plan->clear();
    const uint64_t* ids = ids_.data() + group * S_;
    const uint64_t self = get_mask(self_, group);
    for (uint32_t s = 0; s < S_; ++s)
        if (((used_[group] >> s) & 1ull) && !((self >> s) & 1ull)) (*plan)[ids[s]] = SendPlan{kinds[s], next[s]};
    return RAFTHIP_OK;
}

int BatchProgressTracker::EnableProgress(uint32_t max_inflight, uint64_t max_inflight_bytes) {
    int rc = Flush();
    if (rc) return rc;
    if ((rc = lib(rafthip_progress_config(e_, max_inflight, max_inflight_bytes)))) return rc;
    progress_ = true;
    max_inflight_ = max_inflight;
    max_inflight_bytes_ = max_inflight_bytes;
    return RAFTHIP_OK;
}

int BatchProgressTracker::InitProgress(uint64_t group, const std::map<uint64_t, ProgressInit>& rows,
                                       uint64_t default_next) {
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    if (!progress_) return fail(RAFTHIP_EINVAL, "EnableProgress must come first");
    int rc = Flush();
    if (rc) return rc;
    std::vector<uint8_t> state(S_, RAFTHIP_STATE_PROBE);
    std::vector<uint64_t> match(S_, 0), next(S_, default_next), pending(S_, 0);
    for (const auto& kv : rows) {
        const int s = slot_of(group, kv.first);
        if (s < 0) return fail(RAFTHIP_EINVAL, "group %llu has no Progress for id %llu", (unsigned long long)group,
                               (unsigned long long)kv.first);
        state[s] = (uint8_t)(kv.second.state | (kv.second.msg_app_flow_paused ? RAFTHIP_PROGRESS_PAUSED : 0));
        match[s] = kv.second.match;
        next[s] = kv.second.next;
        pending[s] = kv.second.pending_snapshot;
    }
    if ((rc = lib(rafthip_upload_match(e_, group, 1, match.data(), 1)))) return rc;
    return lib(rafthip_upload_progress(e_, group, 1, state.data(), next.data(), pending.data(), 1));
}

int BatchProgressTracker::SetFirstIndex(uint64_t group, uint64_t first_index) {
    int rc = Flush();
    if (rc) return rc;
    return lib(rafthip_upload_first_index(e_, group, 1, &first_index));
}

bool BatchProgressTracker::StepEvent(uint64_t group, uint64_t peer, int type, uint64_t index, bool reject, bool flag,
                                     uint64_t aux) {
    if (group >= G_ || !progress_) return false;
    const int s = slot_of(group, peer);
    if (s < 0) return false;
    if (type == RAFTHIP_MSG_APP_RESP || type == RAFTHIP_MSG_HEARTBEAT_RESP) {  // the shadowed bit planes follow
        put_mask(active_, group, get_mask(active_, group) | (1ull << s));
        touch(active_dirty_, group);
        if (type == RAFTHIP_MSG_HEARTBEAT_RESP && flag) {
            put_mask(rdacks_, group, get_mask(rdacks_, group) | (1ull << s));
            touch(rdacks_dirty_, group);
        }
    }
    rafthip_msg m;
    memset(&m, 0, sizeof(m));
    m.group = (uint32_t)group;
    m.slot = (uint8_t)s;
    m.type = (uint8_t)type;
    m.reject = reject ? 1 : 0;
    m.flags = flag ? 1 : 0;  // RAFTHIP_MSGF_READ_CTX / "carried entries"
    m.index = index;
    msgs_.push_back(m);
    aux_.push_back(aux);
    return true;
}

int BatchProgressTracker::FlushProgress(std::vector<uint8_t>* actions) {
    if (!progress_) return fail(RAFTHIP_EINVAL, "EnableProgress must come first");
    int rc = RAFTHIP_OK;
    if (!acks_.empty()) {
        rc = lib(rafthip_scatter_acks(e_, acks_.data(), acks_.size()));
        acks_.clear();
        if (rc) return rc;
    }
    if (actions) actions->assign(msgs_.size(), 0);
    if (!msgs_.empty()) {
        rc = lib(rafthip_step_progress(e_, msgs_.data(), aux_.data(), msgs_.size(), actions ? actions->data() : nullptr));
        msgs_.clear();
        aux_.clear();
    }
    return rc;
}

int BatchProgressTracker::ProgressOf(uint64_t group, uint64_t id, ProgressView* out) {
    if (!out) return fail(RAFTHIP_EINVAL, "out is NULL");
    if (group >= G_) return fail(RAFTHIP_ERANGE, "group %llu outside [0, %llu)", (unsigned long long)group,
                                 (unsigned long long)G_);
    if (!progress_) return fail(RAFTHIP_EINVAL, "EnableProgress must come first");
    const int s = slot_of(group, id);
    if (s < 0) return fail(RAFTHIP_EINVAL, "group %llu has no Progress for id %llu", (unsigned long long)group,
                           (unsigned long long)id);
    int rc = Flush();
    if (rc) return rc;
    std::vector<uint8_t> state(S_, 0);
    std::vector<uint64_t> match(S_, 0), next(S_, 0), pending(S_, 0), ibytes(S_, 0);
    std::vector<uint32_t> meta(S_, 0);
    if ((rc = lib(rafthip_read_match(e_, group, 1, match.data(), 1)))) return rc;
    if ((rc = lib(rafthip_read_progress(e_, group, 1, state.data(), next.data(), pending.data(), meta.data(),
                                        ibytes.data(), 1))))
        return rc;
    out->state = state[s] & 3;
    out->msg_app_flow_paused = (state[s] & RAFTHIP_PROGRESS_PAUSED) != 0;
    out->is_paused = out->state == RAFTHIP_STATE_SNAPSHOT ? true : out->msg_app_flow_paused;
    out->match = match[s];
    out->next = next[s];
    out->pending_snapshot = pending[s];
    out->recent_active = ((get_mask(active_, group) >> s) & 1) != 0;
    out->inflight_count = meta[s] >> 16;
    out->inflight_bytes = ibytes[s];
    out->inflight_full = out->inflight_count == max_inflight_ ||
                         (max_inflight_bytes_ != 0 && ibytes[s] >= max_inflight_bytes_);
    return RAFTHIP_OK;
}

}  // namespace tracker
}  // namespace raft
