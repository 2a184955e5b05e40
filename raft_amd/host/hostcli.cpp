// hostcli.cpp — line-oriented driver for the C++ host mirror, in the spirit of
// the reference's datadriven harness (quorum/datadriven_test.go:36-250): one
// command per line on stdin, results on stdout.  tests/test_host_mirror.py
// feeds it the reference's golden cases and random scenarios.
//
// Stateless (quorum::Batch*), batched until `run` / `marshal`:
//   committed in=1,2,3 out=4,5 idx=1:10,3:20      ids absent from idx are "_" (not found)
//   vote      in=1,2,3 out=     votes=1:y,2:n
//   run [device]     -> "committed <i> <value|inf>" / "vote <i> <VoteWon|...>" per queued line
//   marshal          -> the [slot][group] planes and masks, without touching the GPU
//
// Stateful (tracker::BatchProgressTracker):
//   tracker <groups> <slots> [device]
//   setconfig <g> in=.. out=.. learners=..
//   reset <g> | ack <g> <id> <n> | commit | commitof <g> | matchof <g> <id>
//   recordvote <g> <id> y|n | resetvotes <g> | tally | voteof <g>
//   active <g> <id> 0|1 | quorumactive | recvack <g> <id> | resetacks <g> | ackquorum
//   leader <g> <self> <term_first> <last_index> | follower <g> | setcommitted <g> <c>
//   appended <g> <last_index> | step <g> <from> <type> <index> <reject 0|1> <ctx 0|1>
//   maybecommit | checkquorum | hbcommit <g> <id>
//   plan <g> <send_if_empty 0|1>     -> "plan <g> <id>:<kind>:<next>,..."
//   heartbeats <g> <term> [ctx-hex]  -> "heartbeat <g> <to> <hex bytes>" per peer, then "heartbeats-done <n>"
//   progress <max_inflight> <max_bytes> | firstindex <g> <i>
//   initprogress <g> <default_next> rows=<id>:<state>:<match>:<next>:<pending>:<paused>,...
//   ev <g> <peer> <type> <index> <reject 0|1> <flag 0|1> <aux>     queued
//   flushprogress  -> "actions a0,a1,..." | progressof <g> <id>
// Any failure prints "error <code> <text>" and processing continues.
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>

#include "quorum.hpp"
#include "tracker_batch.hpp"

using namespace raft;

namespace {

std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out;
    std::string cur;
    std::stringstream ss(s);
    while (std::getline(ss, cur, sep))
        if (!cur.empty()) out.push_back(cur);
    return out;
}

std::string arg(const std::vector<std::string>& toks, const char* key) {
    const std::string k = std::string(key) + "=";
    for (const auto& t : toks)
        if (t.compare(0, k.size(), k) == 0) return t.substr(k.size());
    return "";
}

quorum::MajorityConfig ids(const std::string& s) {
    quorum::MajorityConfig c;
    for (const auto& t : split(s, ',')) c.insert(strtoull(t.c_str(), nullptr, 10));
    return c;
}

void print_planes(const quorum::Planes& p, bool votes) {
    printf("planes groups=%" PRIu64 " slots=%u mask_bytes=%u\n", p.groups, p.slots, p.mask_bytes);
    for (uint64_t g = 0; g < p.groups; ++g) {
        uint64_t in = 0, out = 0;
        memcpy(&in, p.in_mask.data() + g * p.mask_bytes, p.mask_bytes);
        memcpy(&out, p.out_mask.data() + g * p.mask_bytes, p.mask_bytes);
        printf("group %" PRIu64 " in=%" PRIu64 " out=%" PRIu64 " cells=", g, in, out);
        for (uint32_t s = 0; s < p.slots; ++s) {
            if (votes)
                printf("%s%u", s ? "," : "", (unsigned)p.cells[(size_t)s * p.groups + g]);
            else
                printf("%s%" PRIu64, s ? "," : "", p.match[(size_t)s * p.groups + g]);
        }
        printf("\n");
    }
}

}  // namespace

int main() {
    std::vector<quorum::JointConfig> ccfgs, vcfgs;
    std::vector<std::unique_ptr<quorum::MapAckIndexer>> idx;
    std::vector<std::map<uint64_t, bool>> votes;
    std::unique_ptr<tracker::BatchProgressTracker> trk;

    std::string line;
    while (std::getline(std::cin, line)) {
        const std::vector<std::string> t = split(line, ' ');
        if (t.empty() || t[0][0] == '#') continue;
        const std::string& cmd = t[0];
        auto u = [&](size_t i) -> uint64_t { return i < t.size() ? strtoull(t[i].c_str(), nullptr, 10) : 0; };
        int rc = 0;
        const char* err = "";

        if (cmd == "committed") {
            ccfgs.push_back(quorum::JointConfig{ids(arg(t, "in")), ids(arg(t, "out"))});
            auto l = std::make_unique<quorum::MapAckIndexer>();
            for (const auto& kv : split(arg(t, "idx"), ',')) {
                const auto p = kv.find(':');
                l->m[strtoull(kv.substr(0, p).c_str(), nullptr, 10)] = strtoull(kv.substr(p + 1).c_str(), nullptr, 10);
            }
            idx.push_back(std::move(l));
        } else if (cmd == "vote") {
            vcfgs.push_back(quorum::JointConfig{ids(arg(t, "in")), ids(arg(t, "out"))});
            std::map<uint64_t, bool> m;
            for (const auto& kv : split(arg(t, "votes"), ',')) {
                const auto p = kv.find(':');
                m[strtoull(kv.substr(0, p).c_str(), nullptr, 10)] = kv[p + 1] == 'y';
            }
            votes.push_back(std::move(m));
        } else if (cmd == "run" || cmd == "marshal") {
            std::vector<const quorum::AckedIndexer*> ls;
            for (const auto& l : idx) ls.push_back(l.get());
            if (cmd == "marshal") {
                quorum::Planes p;
                if (!ccfgs.empty() && !(rc = quorum::MarshalCommitted(ccfgs, ls, &p))) print_planes(p, false);
                if (!rc && !vcfgs.empty() && !(rc = quorum::MarshalVotes(vcfgs, votes, &p))) print_planes(p, true);
            } else {
                const int dev = (int)u(1);
                std::vector<quorum::Index> ci;
                std::vector<quorum::VoteResult> vr;
                if (!(rc = quorum::BatchCommittedIndex(ccfgs, ls, &ci, dev)))
                    for (size_t i = 0; i < ci.size(); ++i)
                        printf("committed %zu %s\n", i, ci[i] == quorum::kIndexInf ? "inf" : std::to_string(ci[i]).c_str());
                if (!rc && !(rc = quorum::BatchVoteResult(vcfgs, votes, &vr, dev)))
                    for (size_t i = 0; i < vr.size(); ++i) printf("vote %zu %s\n", i, quorum::VoteResultString(vr[i]));
            }
            err = quorum::LastError();
            ccfgs.clear();
            vcfgs.clear();
            idx.clear();
            votes.clear();
        } else if (cmd == "tracker") {
            tracker::BatchProgressTracker* p = nullptr;
            rc = tracker::BatchProgressTracker::Make((int)u(3), u(1), (uint32_t)u(2), &p);
            err = rafthip_last_error();
            trk.reset(p);
            if (!rc) printf("tracker ok\n");
        } else if (!trk) {
            rc = RAFTHIP_EINVAL;
            err = "no tracker";
        } else {
            if (cmd == "setconfig") {
                const std::string li = arg(t, "last");  // Changer.LastIndex, for the Progress of new peers
                rc = trk->SetConfig(u(1), ids(arg(t, "in")), ids(arg(t, "out")), ids(arg(t, "learners")),
                                    li.empty() ? 0 : strtoull(li.c_str(), nullptr, 10));
            } else if (cmd == "reset") {
                rc = trk->ResetProgress(u(1));
            } else if (cmd == "ack") {
                if (!trk->MaybeUpdate(u(1), u(2), u(3))) printf("dropped\n");
            } else if (cmd == "commit") {
                std::vector<tracker::CommitPair> ch;
                if (!(rc = trk->Committed(&ch))) {
                    for (const auto& p : ch)
                        printf("changed %" PRIu64 " %s\n", p.group,
                               p.commit == quorum::kIndexInf ? "inf" : std::to_string(p.commit).c_str());
                    printf("commit-done %zu\n", ch.size());
                }
            } else if (cmd == "commitof") {
                uint64_t c = 0;
                if (!(rc = trk->CommittedOf(u(1), &c)))
                    printf("commitof %" PRIu64 " %s\n", u(1), c == quorum::kIndexInf ? "inf" : std::to_string(c).c_str());
            } else if (cmd == "matchof") {
                uint64_t m = 0;
                if (!(rc = trk->Match(u(1), u(2), &m))) printf("matchof %" PRIu64 " %" PRIu64 " %" PRIu64 "\n", u(1), u(2), m);
            } else if (cmd == "recordvote") {
                rc = trk->RecordVote(u(1), u(2), t.size() > 3 && t[3] == "y");
            } else if (cmd == "resetvotes") {
                rc = trk->ResetVotes(u(1));
            } else if (cmd == "tally") {
                rc = trk->TallyVotes();
            } else if (cmd == "voteof") {
                int g = 0, r = 0;
                quorum::VoteResult res = quorum::VotePending;
                if (!(rc = trk->VoteResultOf(u(1), &g, &r, &res)))
                    printf("voteof %" PRIu64 " %d %d %s\n", u(1), g, r, quorum::VoteResultString(res));
            } else if (cmd == "active") {
                rc = trk->SetRecentActive(u(1), u(2), u(3) != 0);
            } else if (cmd == "quorumactive") {
                std::vector<uint8_t> a;
                if (!(rc = trk->QuorumActive(&a))) {
                    printf("quorumactive ");
                    for (uint8_t x : a) putchar('0' + x);
                    printf("\n");
                }
            } else if (cmd == "recvack") {
                rc = trk->RecvAck(u(1), u(2));
            } else if (cmd == "resetacks") {
                rc = trk->ResetAcks(u(1));
            } else if (cmd == "ackquorum") {
                std::vector<quorum::VoteResult> a;
                if (!(rc = trk->AckQuorum(&a))) {
                    printf("ackquorum ");
                    for (auto x : a) putchar('0' + (int)x);
                    printf("\n");
                }
            } else if (cmd == "leader") {
                rc = trk->SetLeader(u(1), u(2), u(3), u(4));
            } else if (cmd == "follower") {
                rc = trk->SetFollower(u(1));
            } else if (cmd == "setcommitted") {
                rc = trk->SetCommitted(u(1), u(2));
            } else if (cmd == "appended") {
                if (!trk->Appended(u(1), u(2))) printf("dropped\n");
            } else if (cmd == "step") {
                if (!trk->Step(u(1), u(2), (int)u(3), u(4), u(5) != 0, u(6) != 0)) printf("dropped\n");
            } else if (cmd == "maybecommit") {
                std::vector<tracker::CommitPair> adv;
                if (!(rc = trk->MaybeCommit(&adv))) {
                    for (const auto& p : adv) printf("advanced %" PRIu64 " %" PRIu64 "\n", p.group, p.commit);
                    printf("maybecommit-done %zu\n", adv.size());
                }
            } else if (cmd == "checkquorum") {
                std::vector<uint8_t> a;
                if (!(rc = trk->CheckQuorum(&a))) {
                    printf("checkquorum ");
                    for (uint8_t x : a) putchar('0' + x);
                    printf("\n");
                }
            } else if (cmd == "tickconfig") {
                rc = trk->SetTickConfig((uint32_t)u(1), (uint32_t)u(2), u(3) != 0);
            } else if (cmd == "tickstate") {  // group el hb rt state promotable transferring
                rc = trk->SetTickState(u(1), (uint32_t)u(2), (uint32_t)u(3), (uint32_t)u(4), (int)u(5), u(6) != 0, u(7) != 0);
            } else if (cmd == "heard") {  // groups=g,g,...
                std::vector<uint32_t> gs;
                for (const auto& x : split(arg(t, "groups"), ',')) gs.push_back((uint32_t)strtoul(x.c_str(), nullptr, 10));
                rc = trk->HeardFromLeader(gs);
            } else if (cmd == "tick") {  // event mask
                std::vector<tracker::BatchProgressTracker::TickEvent> ev;
                uint64_t c[5];
                if (!(rc = trk->Tick((uint32_t)u(1), &ev, c))) {
                    for (const auto& x : ev) printf("tickevent %u %u\n", x.group, x.action);
                    printf("tick %" PRIu64 " %" PRIu64 " %" PRIu64 " %" PRIu64 " %" PRIu64 "\n", c[0], c[1], c[2], c[3], c[4]);
                }
            } else if (cmd == "progress") {
                rc = trk->EnableProgress((uint32_t)u(1), u(2));
            } else if (cmd == "firstindex") {
                rc = trk->SetFirstIndex(u(1), u(2));
            } else if (cmd == "initprogress") {
                std::map<uint64_t, tracker::BatchProgressTracker::ProgressInit> rows;
                for (const auto& r : split(arg(t, "rows"), ',')) {
                    const auto f = split(r, ':');
                    if (f.size() != 6) continue;
                    auto n = [&](size_t i) { return strtoull(f[i].c_str(), nullptr, 10); };
                    rows[n(0)] = {(int)n(1), n(2), n(3), n(4), n(5) != 0};
                }
                rc = trk->InitProgress(u(1), rows, u(2));
            } else if (cmd == "ev") {
                if (!trk->StepEvent(u(1), u(2), (int)u(3), u(4), u(5) != 0, u(6) != 0, u(7))) printf("dropped\n");
            } else if (cmd == "flushprogress") {
                std::vector<uint8_t> a;
                if (!(rc = trk->FlushProgress(&a))) {
                    printf("actions");
                    for (size_t i = 0; i < a.size(); ++i) printf("%c%u", i ? ',' : ' ', (unsigned)a[i]);
                    printf("\n");
                }
            } else if (cmd == "progressof") {
                tracker::BatchProgressTracker::ProgressView v;
                if (!(rc = trk->ProgressOf(u(1), u(2), &v)))
                    printf("progressof %" PRIu64 " %" PRIu64 " state=%d match=%" PRIu64 " next=%" PRIu64 " pending=%" PRIu64
                           " paused=%d is_paused=%d active=%d inflight=%u bytes=%" PRIu64 " full=%d\n",
                           u(1), u(2), v.state, v.match, v.next, v.pending_snapshot, (int)v.msg_app_flow_paused,
                           (int)v.is_paused, (int)v.recent_active, v.inflight_count, v.inflight_bytes,
                           (int)v.inflight_full);
            } else if (cmd == "hbcommit") {
                uint64_t c = 0;
                if (!(rc = trk->HeartbeatCommit(u(1), u(2), &c)))
                    printf("hbcommit %" PRIu64 " %" PRIu64 " %" PRIu64 "\n", u(1), u(2), c);
            } else if (cmd == "plan") {
                std::map<uint64_t, tracker::BatchProgressTracker::SendPlan> plan;
                if (!(rc = trk->PlanAppends(u(1), u(2) != 0, &plan))) {
                    printf("plan %" PRIu64, u(1));
                    bool firstp = true;
                    for (const auto& kv : plan) {
                        printf("%c%" PRIu64 ":%d:%" PRIu64, firstp ? ' ' : ',', kv.first, kv.second.kind, kv.second.next);
                        firstp = false;
                    }
                    printf("\n");
                }
            } else if (cmd == "heartbeats") {
                std::string ctx;
                if (t.size() > 3)
                    for (size_t i = 0; i + 1 < t[3].size(); i += 2) ctx.push_back((char)strtoul(t[3].substr(i, 2).c_str(), nullptr, 16));
                std::map<uint64_t, std::string> msgs;
                if (!(rc = trk->BcastHeartbeat(u(1), u(2), ctx, &msgs))) {
                    for (const auto& kv : msgs) {
                        printf("heartbeat %" PRIu64 " %" PRIu64 " ", u(1), kv.first);
                        for (unsigned char c : kv.second) printf("%02x", c);
                        printf("\n");
                    }
                    printf("heartbeats-done %zu\n", msgs.size());
                }
            } else {
                rc = RAFTHIP_EINVAL;
                err = "unknown command";
            }
            if (rc && !*err) err = trk->LastError();
        }
        if (rc) printf("error %d %s\n", rc, err);
        fflush(stdout);
    }
    return 0;
}
