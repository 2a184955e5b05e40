// tracker_batch.hpp — C++ mirror of the quorum-facing part of the reference's
// tracker.ProgressTracker (tracker/tracker.go:117-290) for MANY groups at once,
// on top of one librafthip.so engine (one GPU).
//
// Method names and meaning follow the reference, with a leading group
// argument; the batched evaluations (Committed, TallyVotes, QuorumActive,
// AckQuorum) cover every group in one GPU pass.  The class owns what the
// reference keeps in Go maps and the kernels cannot see: the id -> slot map of
// every group.  It does no quorum arithmetic itself.
//
// Thread-unsafe, like RawNode (rawnode.go:31): one caller at a time.
#pragma once
#include <stdint.h>

#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "quorum.hpp"
#include "rafthip.h"

namespace raft {
namespace tracker {

struct CommitPair {  // = rafthip_commit_pair
    uint64_t group;
    uint64_t commit;
};

class BatchProgressTracker {
   public:
    // Stands where tracker.MakeProgressTracker (tracker/tracker.go:129-145)
    // stands per group: every group starts with an empty config and no Progress.
    static int Make(int device, uint64_t groups, uint32_t max_slots, BatchProgressTracker** out);
    ~BatchProgressTracker();
    BatchProgressTracker(const BatchProgressTracker&) = delete;
    BatchProgressTracker& operator=(const BatchProgressTracker&) = delete;

    uint64_t groups() const { return G_; }
    uint32_t max_slots() const { return S_; }
    const char* LastError() const { return err_; }

    // The group's tracker.Config after a conf change (what confchange.Changer
    // hands to raft.switchToConfig, raft.go:1949): incoming = Voters[0],
    // outgoing = Voters[1], learners = Learners (LearnersNext ids are still
    // voters of the outgoing half, confchange.go:297-304).  Ids that stay keep
    // their Progress; removed ids lose it; new ids start with Match 0 and
    // RecentActive true (confchange.go:257-270).
    // With EnableProgress, a new id's row is also re-created the way Changer.initProgress
    // does — StateProbe, Next = last_index (confchange.go:262), empty Inflights — and the rows of the
    // ids that stay are left alone (last_index = Changer.LastIndex, the leader's last index).
    int SetConfig(uint64_t group, const quorum::MajorityConfig& incoming,
                  const quorum::MajorityConfig& outgoing, const quorum::MajorityConfig& learners,
                  uint64_t last_index = 0);

    // raft.reset on a role change (raft.go:774-784): every Progress of the
    // group is re-created — Match 0, RecentActive false — and votes are reset.
    int ResetProgress(uint64_t group);

    // Progress.MaybeUpdate's Match part (tracker/progress.go:168-177) for the
    // ack of peer `id`; queued and applied by the next Committed()/Flush().
    // Returns false when the group has no Progress for id (the reference drops
    // such messages, raft.go:1327-1331) or the group is out of range.
    bool MaybeUpdate(uint64_t group, uint64_t id, uint64_t n);
    int Flush();

    // ProgressTracker.Committed (tracker/tracker.go:179-181) for every group;
    // *changed receives the groups whose commit index differs from the last
    // call, sorted by group — the input of raft.maybeCommit (raft.go:755-758).
    int Committed(std::vector<CommitPair>* changed);
    int CommittedOf(uint64_t group, uint64_t* commit);  // after Committed()
    int Match(uint64_t group, uint64_t id, uint64_t* match);

    // tracker/tracker.go:254-265
    int ResetVotes(uint64_t group);
    int RecordVote(uint64_t group, uint64_t id, bool v);  // first vote of an id sticks
    // ProgressTracker.TallyVotes (tracker/tracker.go:269-290) for every group.
    int TallyVotes();
    int VoteResultOf(uint64_t group, int* granted, int* rejected, quorum::VoteResult* result);

    // Progress.RecentActive (set at raft.go:1337,1528; cleared at raft.go:1238-1242)
    int SetRecentActive(uint64_t group, uint64_t id, bool active);
    // ProgressTracker.QuorumActive (tracker/tracker.go:217-227) for every group.
    int QuorumActive(std::vector<uint8_t>* active);

    // readOnly.recvAck (read_only.go:68-76) for the group's pending ReadIndex
    // request, and the quorum test of raft.go:1552 for every group.
    int RecvAck(uint64_t group, uint64_t id);
    int ResetAcks(uint64_t group);
    int AckQuorum(std::vector<quorum::VoteResult>* result);

    // ---- the leader-side tick with the steps either side of the quorum maths ----
    // (use either Committed() or MaybeCommit() on one tracker: the first keeps the
    //  last quorum index in the engine's commit plane, the second raftLog.committed)

    // becomeLeader (raft.go:905-936): `self` is r.id; the entries carrying r.Term
    // are the log indexes [term_first, last_index] (the empty entry appended at
    // becomeLeader is the first one).  Also marks self RecentActive (raft.go:920).
    int SetLeader(uint64_t group, uint64_t self, uint64_t term_first, uint64_t last_index);
    // Any other role: raftLog.maybeCommit can never pass (term(i) != r.Term).
    int SetFollower(uint64_t group);
    // raftLog.committed as the host knows it (restore, follower-side commits).
    int SetCommitted(uint64_t group, uint64_t committed);
    // appendEntry on the leader (raft.go:787-819): lastIndex grew.
    bool Appended(uint64_t group, uint64_t last_index);
    // stepLeader's bookkeeping for one response (raft.go:1327-1561): type is
    // RAFTHIP_MSG_APP_RESP or RAFTHIP_MSG_HEARTBEAT_RESP; read_ctx says the
    // heartbeat response names the group's pending ReadIndex request.  Returns
    // false when there is no Progress for `from` (raft.go:1327-1331).
    bool Step(uint64_t group, uint64_t from, int type, uint64_t index, bool reject, bool read_ctx);
    // raft.maybeCommit for every group (raft.go:755-758): *advanced receives the
    // groups whose raftLog.committed moved, sorted by group.
    int MaybeCommit(std::vector<CommitPair>* advanced);
    // The MsgCheckQuorum step for every group (raft.go:1231-1243): verdicts as
    // QuorumActive(), then RecentActive = false for all but each group's self.
    int CheckQuorum(std::vector<uint8_t>* active);
    // RawNode.Tick for every group (rawnode.go:64-66 -> tickElection / tickHeartbeat, raft.go:823-862).
    // SetTickConfig: ElectionTick, HeartbeatTick, CheckQuorum of raft.Config (validated as raft.go:296-302).
    // SetTickState: what reset / becomeLeader / becomeFollower / a transfer leave in one group — electionElapsed,
    // heartbeatElapsed, randomizedElectionTimeout (drawn by the caller, raft.go:1988-1990), r.state (StateType,
    // raft.go:48-53), promotable() (raft.go:1883-1886) and leadTransferee != None.
    // Tick: one tick for all groups.  counts[5]: groups that stepped MsgHup, ran MsgCheckQuorum, stepped down,
    // timed a transfer out, stepped MsgBeat.  *events (may be NULL): the groups with an action of `event_mask`
    // (RAFTHIP_TICK_*), in group order.  MsgCheckQuorum is done for the groups it was due for, RecentActive reset
    // included (the shadow follows); a group that stepped down is no longer led here (as SetFollower).
    struct TickEvent {
        uint32_t group;
        uint32_t action;
    };
    int SetTickConfig(uint32_t election_tick, uint32_t heartbeat_tick, bool check_quorum);
    int SetTickState(uint64_t group, uint32_t election_elapsed, uint32_t heartbeat_elapsed, uint32_t randomized_timeout,
                     int state, bool promotable, bool transferring);
    int Tick(uint32_t event_mask, std::vector<TickEvent>* events, uint64_t counts[5]);
    // stepFollower's r.electionElapsed = 0 for the groups whose leader sent a MsgApp, MsgHeartbeat or MsgSnap of the
    // current term since the last Tick (raft.go:1681-1692).
    int HeardFromLeader(const std::vector<uint32_t>& groups);
    // sendHeartbeat's commit for peer `id` (raft.go:669-685), after MaybeCommit().
    int HeartbeatCommit(uint64_t group, uint64_t id, uint64_t* commit);

    // bcastHeartbeat for one led group (raft.go:698-716 -> sendHeartbeat :669-685 -> r.send :502-533), after
    // MaybeCommit(): *msgs receives, per peer id but the leader's own, the serialized raftpb.Message
    // {Type: MsgHeartbeat, To: id, From: self, Term: term, Commit: min(Match, committed), Context: ctx} exactly as
    // Message.Marshal writes it (raftpb/raft.pb.go:937-1025).  ctx is the last pending ReadIndex request
    // context (readOnly.lastPendingRequestCtx, read_only.go:116-121), at most 8 bytes here; empty = nil.
    int BcastHeartbeat(uint64_t group, uint64_t term, const std::string& ctx, std::map<uint64_t, std::string>* msgs);

    // bcastAppend for one led group as a plan (raft.go:689-696): per peer id but the leader's own, what
    // maybeSendAppend(id, send_if_empty) decides right now (raft.go:600-666) — RAFTHIP_SEND_NONE / _APPEND /
    // _EMPTY / _SNAPSHOT — and pr.Next (a MsgApp's Index is Next-1, its entries start at Next).  Needs
    // EnableProgress; steps what is queued first.
    struct SendPlan {
        int kind;
        uint64_t next;
    };
    int PlanAppends(uint64_t group, bool send_if_empty, std::map<uint64_t, SendPlan>* plan);

    // ---- per-peer replication state: tracker.Progress + tracker.Inflights ----------
    // (tracker/progress.go:30-98, tracker/inflights.go:28-40)

    // The limits tracker.MakeProgressTracker takes (tracker/tracker.go:129-145):
    // Config.MaxInflightMsgs / MaxInflightBytes.  From here on Step/StepEvent run
    // the whole stepLeader branch (Probe/Replicate/Snapshot, Next, Inflights).
    int EnableProgress(uint32_t max_inflight, uint64_t max_inflight_bytes);
    struct ProgressInit {
        int state;  // RAFTHIP_STATE_*
        uint64_t match, next, pending_snapshot;
        bool msg_app_flow_paused;
    };
    // Every Progress of the group at once, as raft.reset / becomeLeader /
    // a conf change leave them (raft.go:774-784,915-920); ids without an entry
    // get the reset value {StateProbe, Match 0, Next `default_next`}.  Inflights
    // start empty.
    int InitProgress(uint64_t group, const std::map<uint64_t, ProgressInit>& rows, uint64_t default_next);
    int SetFirstIndex(uint64_t group, uint64_t first_index);  // raftLog.firstIndex()
    // One event for the Progress of `peer`, queued: type is a raftpb.MessageType
    // (MsgAppResp 4, MsgHeartbeatResp 9, MsgUnreachable 10, MsgSnapStatus 11) or
    // RAFTHIP_MSG_SENT_APP / RAFTHIP_MSG_SENT_SNAP; flag = read_ctx for a heartbeat
    // response, "carried entries" for SENT_APP; aux = nextProbeIdx of a rejection,
    // payload bytes of SENT_APP.  Returns false when the group has no Progress
    // for peer (raft.go:1327-1331).
    bool StepEvent(uint64_t group, uint64_t peer, int type, uint64_t index, bool reject, bool flag, uint64_t aux);
    // Steps everything queued, in order; *actions gets one RAFTHIP_ACT_* byte set
    // per StepEvent / Step / Appended call since the last flush.
    int FlushProgress(std::vector<uint8_t>* actions);
    struct ProgressView {
        int state;
        uint64_t match, next, pending_snapshot;
        bool msg_app_flow_paused, recent_active, is_paused;  // IsPaused(), progress.go:225-236
        uint32_t inflight_count;
        uint64_t inflight_bytes;
        bool inflight_full;  // Inflights.Full(), inflights.go:131-133
    };
    int ProgressOf(uint64_t group, uint64_t id, ProgressView* out);

    rafthip_engine* engine() { return e_; }

   private:
    BatchProgressTracker() = default;
    int fail(int code, const char* fmt, ...);
    int lib(int rc);
    int slot_of(uint64_t group, uint64_t id) const;  // -1 if the group has no Progress for id
    uint64_t get_mask(const std::vector<uint8_t>& plane, uint64_t g) const;
    void put_mask(std::vector<uint8_t>& plane, uint64_t g, uint64_t m);
    void touch(uint64_t (&range)[2], uint64_t g);
    int push_config(uint64_t group);

    rafthip_engine* e_ = nullptr;
    uint64_t G_ = 0;
    uint32_t S_ = 0, mb_ = 1;
    char err_[512] = "";

    std::vector<uint64_t> ids_;      // [G][S] voter/learner id in each slot
    std::vector<uint64_t> used_;     // [G] slots that hold a Progress
    std::vector<uint8_t> in_, out_, learner_;  // [G] masks, mb_ bytes each
    std::vector<rafthip_ack> acks_;  // queued MaybeUpdate calls
    std::vector<rafthip_msg> msgs_;  // queued Step / StepEvent / Appended calls
    std::vector<uint64_t> aux_;      // one per queued msg once EnableProgress ran
    bool progress_ = false;
    uint32_t max_inflight_ = 0;
    uint64_t max_inflight_bytes_ = 0;
    std::vector<uint8_t> self_;      // [G] masks: the leader's own slot
    uint64_t self_dirty_[2] = {1, 0};
    int SyncActiveAndSelf();  // dirty ranges of the two shadows to the engine
    std::vector<CommitPair> pairs_;  // landing buffer of Committed(), G entries
    // Votes is keyed by id, not by Progress (tracker/tracker.go:122,260-265):
    // votes of ids that hold no slot right now wait here until ResetVotes.
    std::unordered_map<uint64_t, std::map<uint64_t, bool>> stray_votes_;
    std::unordered_map<uint64_t, std::set<uint64_t>> stray_acks_;  // same for readIndexStatus.acks

    // host shadows of the small planes, uploaded lazily over their dirty range
    std::vector<uint8_t> votes_;   // [S][G]
    std::vector<uint8_t> active_;  // [G] masks
    std::vector<uint8_t> rdacks_;  // [G] masks
    uint64_t votes_dirty_[2] = {1, 0}, active_dirty_[2] = {1, 0}, rdacks_dirty_[2] = {1, 0};
};

}  // namespace tracker
}  // namespace raft
