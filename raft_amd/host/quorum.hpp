// quorum.hpp — C++ mirror of the reference's Go package `quorum`, for hosts that
// sit above the C ABI of librafthip.so (include/rafthip.h).
//
// The reference is Go and no Go toolchain exists in the build image, so the
// host side of the drop-in boundary is written in C++ with the reference's own
// names, argument meaning and error behaviour; INTEGRATION.md shows the Go/cgo
// file that does the same marshalling.  Per-group types keep the reference's
// shape (a config is a set of voter ids, an indexer is an interface); only the
// two Batch* functions are new, and they do no quorum arithmetic themselves:
// they lay the groups out as [slot][group] planes and call the GPU.
#pragma once
#include <stdint.h>

#include <array>
#include <map>
#include <set>
#include <string>
#include <vector>

namespace raft {
namespace quorum {

// quorum/quorum.go:23 — a Raft log position; UINT64_MAX prints as "∞".
using Index = uint64_t;
constexpr Index kIndexInf = ~0ull;
std::string IndexString(Index i);  // quorum/quorum.go:25-30

// quorum/quorum.go:34-36
class AckedIndexer {
   public:
    virtual ~AckedIndexer() = default;
    // Returns true and sets *idx if the voter is known (Go: (idx, found)).
    virtual bool AckedIndex(uint64_t voter_id, Index* idx) const = 0;
};

// quorum/quorum.go:38-43 — the map-backed indexer the reference's tests use.
class MapAckIndexer : public AckedIndexer {
   public:
    std::map<uint64_t, Index> m;
    bool AckedIndex(uint64_t voter_id, Index* idx) const override;
};

// quorum/quorum.go:48-58 (0 is not a valid value)
enum VoteResult : uint8_t { VotePending = 1, VoteLost = 2, VoteWon = 3 };
const char* VoteResultString(VoteResult r);  // quorum/voteresult_string.go:16-26

// quorum/majority.go:25 — a set of voter ids.
using MajorityConfig = std::set<uint64_t>;
// quorum/joint.go:19 — [0] incoming, [1] outgoing (empty when not joint).
using JointConfig = std::array<MajorityConfig, 2>;
// quorum/joint.go:31-39
MajorityConfig IDs(const JointConfig& c);

// Flat [slot][group] planes for a batch of groups: what both Batch* functions
// hand to rafthip_batch_*.  Slot k of group i is the k-th smallest id of
// IDs(cfgs[i]); unused slots stay zero and belong to neither half.
struct Planes {
    uint64_t groups = 0;
    uint32_t slots = 0;           // max over groups of |IDs|, at least 1
    uint32_t mask_bytes = 1;      // 1/2/4/8 for slots <= 8/16/32/64
    std::vector<uint64_t> match;  // [slots][groups] (committed)
    std::vector<uint8_t> cells;   // [slots][groups] (votes: 0 none, 1 rejected, 2 granted)
    std::vector<uint8_t> in_mask, out_mask;  // groups * mask_bytes, little-endian
};

// status codes are those of rafthip.h (0 ok, negative errno-style); the text
// of the last failure on this thread is in LastError().
const char* LastError();

// Marshalling only (no GPU): exposed so it can be tested on a CPU-only host.
int MarshalCommitted(const std::vector<JointConfig>& cfgs, const std::vector<const AckedIndexer*>& ls,
                     Planes* out);
int MarshalVotes(const std::vector<JointConfig>& cfgs, const std::vector<std::map<uint64_t, bool>>& votes,
                 Planes* out);

// New entry point beside the per-group API: out[i] = cfgs[i].CommittedIndex(ls[i])
// (quorum/joint.go:49-56 over quorum/majority.go:126-172) for every i, on GPU
// `device`.  The reference panics on misuse; here len(cfgs) != len(ls), a null
// indexer or more than 64 voters in a group return RAFTHIP_EINVAL.
int BatchCommittedIndex(const std::vector<JointConfig>& cfgs, const std::vector<const AckedIndexer*>& ls,
                        std::vector<Index>* out, int device = 0);

// out[i] = cfgs[i].VoteResult(votes[i]) (quorum/joint.go:61-75 over
// quorum/majority.go:178-207).
int BatchVoteResult(const std::vector<JointConfig>& cfgs, const std::vector<std::map<uint64_t, bool>>& votes,
                    std::vector<VoteResult>* out, int device = 0);

}  // namespace quorum
}  // namespace raft
