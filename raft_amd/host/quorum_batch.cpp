// quorum_batch.cpp — marshalling behind quorum::BatchCommittedIndex /
// BatchVoteResult.  No quorum arithmetic happens here: groups are laid out as
// [slot][group] planes and handed to librafthip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "quorum.hpp"
#include "rafthip.h"

namespace raft {
namespace quorum {

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int from_lib(int rc) {
    if (rc) snprintf(g_err, sizeof(g_err), "%s", rafthip_last_error());
    return rc;
}

uint32_t mask_bytes_for(uint32_t s) { return s <= 8 ? 1 : s <= 16 ? 2 : s <= 32 ? 4 : 8; }

void put_mask(std::vector<uint8_t>& plane, uint64_t g, uint32_t mb, uint64_t m) {
    memcpy(plane.data() + g * mb, &m, mb);  // little-endian host
}

// Sorted ids of IDs(cfg) per group and the widest group.
int assign_slots(const std::vector<JointConfig>& cfgs, std::vector<std::vector<uint64_t>>* slots, Planes* p) {
    uint32_t s = 1;
    slots->resize(cfgs.size());
    for (size_t i = 0; i < cfgs.size(); ++i) {
        const MajorityConfig ids = IDs(cfgs[i]);
        if (ids.size() > RAFTHIP_MAX_SLOTS)
            return fail(RAFTHIP_EINVAL, "group %zu has %zu voters; the batch path supports at most %u", i,
                        ids.size(), RAFTHIP_MAX_SLOTS);
        (*slots)[i].assign(ids.begin(), ids.end());
        if (ids.size() > s) s = (uint32_t)ids.size();
    }
    p->groups = cfgs.size();
    p->slots = s;
    p->mask_bytes = mask_bytes_for(s);
    p->in_mask.assign(p->groups * p->mask_bytes, 0);
    p->out_mask.assign(p->groups * p->mask_bytes, 0);
    for (size_t i = 0; i < cfgs.size(); ++i) {
        uint64_t in = 0, out = 0;
        for (size_t k = 0; k < (*slots)[i].size(); ++k) {
            const uint64_t id = (*slots)[i][k];
            if (cfgs[i][0].count(id)) in |= 1ull << k;
            if (cfgs[i][1].count(id)) out |= 1ull << k;
        }
        put_mask(p->in_mask, i, p->mask_bytes, in);
        put_mask(p->out_mask, i, p->mask_bytes, out);
    }
    return RAFTHIP_OK;
}
}  // namespace

const char* LastError() { return g_err; }

std::string IndexString(Index i) { return i == kIndexInf ? "\xE2\x88\x9E" : std::to_string(i); }

bool MapAckIndexer::AckedIndex(uint64_t voter_id, Index* idx) const {
    auto it = m.find(voter_id);
    if (it == m.end()) return false;
    *idx = it->second;
    return true;
}

const char* VoteResultString(VoteResult r) {
    switch (r) {
        case VotePending: return "VotePending";
        case VoteLost: return "VoteLost";
        case VoteWon: return "VoteWon";
    }
    return "VoteResult(?)";
}

MajorityConfig IDs(const JointConfig& c) {
    MajorityConfig m = c[0];
    m.insert(c[1].begin(), c[1].end());
    return m;
}

int MarshalCommitted(const std::vector<JointConfig>& cfgs, const std::vector<const AckedIndexer*>& ls,
                     Planes* p) {
    if (!p) return fail(RAFTHIP_EINVAL, "out is NULL");
    if (cfgs.size() != ls.size())
        return fail(RAFTHIP_EINVAL, "len(cfgs) %zu != len(ls) %zu", cfgs.size(), ls.size());
    std::vector<std::vector<uint64_t>> slots;
    int rc = assign_slots(cfgs, &slots, p);
    if (rc) return rc;
    p->cells.clear();
    p->match.assign((size_t)p->slots * p->groups, 0);
    for (size_t i = 0; i < cfgs.size(); ++i) {
        if (!ls[i]) return fail(RAFTHIP_EINVAL, "ls[%zu] is NULL", i);
        for (size_t k = 0; k < slots[i].size(); ++k) {
            Index idx = 0;
            // a voter the indexer does not know stays 0 — quorum/majority.go:149-161
            if (ls[i]->AckedIndex(slots[i][k], &idx)) p->match[k * p->groups + i] = idx;
        }
    }
    return RAFTHIP_OK;
}

int MarshalVotes(const std::vector<JointConfig>& cfgs, const std::vector<std::map<uint64_t, bool>>& votes,
                 Planes* p) {
    if (!p) return fail(RAFTHIP_EINVAL, "out is NULL");
    if (cfgs.size() != votes.size())
        return fail(RAFTHIP_EINVAL, "len(cfgs) %zu != len(votes) %zu", cfgs.size(), votes.size());
    std::vector<std::vector<uint64_t>> slots;
    int rc = assign_slots(cfgs, &slots, p);
    if (rc) return rc;
    p->match.clear();
    p->cells.assign((size_t)p->slots * p->groups, RAFTHIP_CELL_NONE);
    for (size_t i = 0; i < cfgs.size(); ++i) {
        for (size_t k = 0; k < slots[i].size(); ++k) {
            // votes of ids outside the config are never looked at — quorum/majority.go:188-197
            auto it = votes[i].find(slots[i][k]);
            if (it != votes[i].end())
                p->cells[k * p->groups + i] = it->second ? RAFTHIP_CELL_GRANT : RAFTHIP_CELL_REJECT;
        }
    }
    return RAFTHIP_OK;
}

int BatchCommittedIndex(const std::vector<JointConfig>& cfgs, const std::vector<const AckedIndexer*>& ls,
                        std::vector<Index>* out, int device) {
    if (!out) return fail(RAFTHIP_EINVAL, "out is NULL");
    Planes p;
    int rc = MarshalCommitted(cfgs, ls, &p);
    if (rc) return rc;
    out->assign(p.groups, 0);
    if (p.groups == 0) return RAFTHIP_OK;
    return from_lib(rafthip_batch_committed_index(device, p.groups, p.slots, p.match.data(), p.in_mask.data(),
                                                  p.out_mask.data(), out->data()));
}

int BatchVoteResult(const std::vector<JointConfig>& cfgs, const std::vector<std::map<uint64_t, bool>>& votes,
                    std::vector<VoteResult>* out, int device) {
    if (!out) return fail(RAFTHIP_EINVAL, "out is NULL");
    Planes p;
    int rc = MarshalVotes(cfgs, votes, &p);
    if (rc) return rc;
    out->assign(p.groups, VotePending);
    if (p.groups == 0) return RAFTHIP_OK;
    static_assert(sizeof(VoteResult) == 1, "VoteResult is uint8 (quorum/quorum.go:48)");
    return from_lib(rafthip_batch_vote_result(device, p.groups, p.slots, p.cells.data(), p.in_mask.data(),
                                              p.out_mask.data(), reinterpret_cast<uint8_t*>(out->data())));
}

}  // namespace quorum
}  // namespace raft
