// Package tracker — batch_cgo.go: the stateful engine of librafthip.so for a
// multi-raft host, next to the untouched per-group ProgressTracker.  Thin cgo
// over include/rafthip.h, zero logic: the id -> slot maps, the routing of
// messages to the engine that owns a group and every send stay with the host
// (raft_amd/host/tracker_batch.{hpp,cpp} is the same layer in C++, tested in
// tests/test_host_mirror.py; this file was written to the ABI but not
// compiled: the build image has no Go toolchain).
package tracker

/*
#cgo LDFLAGS: -lrafthip
#include <rafthip.h>
*/
import "C"

import (
	"fmt"
	"runtime"
	"unsafe"
)

// BatchEngine holds the quorum-facing state of many groups on one GPU.  Like
// RawNode it is not safe for concurrent use (rawnode.go:31); one engine per
// GPU, different engines from different OS-thread-locked goroutines.
type BatchEngine struct {
	e      *C.rafthip_engine
	groups uint64
	slots  uint32
}

// Ack is one non-reject MsgAppResp resolved to (group, slot): raft.go:1333,1477.
type Ack = C.rafthip_ack

// Msg is one response or send-side event: include/rafthip.h, rafthip_msg.
type Msg = C.rafthip_msg

// CommitPair is a group whose commit index moved: the input of raft.maybeCommit.
type CommitPair = C.rafthip_commit_pair

// guarded runs one library call and, if it failed, fetches its error text.  The
// text is thread-local in the library (rafthip_last_error), and a goroutine may
// move to another OS thread between two cgo calls: the goroutine is locked to its
// thread for the pair.
func guarded(call func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	rc := call()
	if rc == 0 {
		return nil
	}
	return fmt.Errorf("rafthip %d: %s", int(rc), C.GoString(C.rafthip_last_error()))
}

// NewBatchEngine stands where MakeProgressTracker (tracker.go:129-145) stands per
// group; maxInflight / maxBytes are its arguments (0, 0 leaves the per-peer
// replication state with the host).
func NewBatchEngine(device int, groups uint64, slots uint32, maxInflight uint32, maxBytes uint64) (*BatchEngine, error) {
	cfg := C.rafthip_config{struct_size: C.uint32_t(unsafe.Sizeof(C.rafthip_config{})), device: C.int32_t(device),
		num_groups: C.uint64_t(groups), num_slots: C.uint32_t(slots)}
	b := &BatchEngine{groups: groups, slots: slots}
	if err := guarded(func() C.int { return C.rafthip_create(&cfg, &b.e) }); err != nil {
		return nil, err
	}
	if maxInflight > 0 {
		if err := guarded(func() C.int { return C.rafthip_progress_config(b.e, C.uint32_t(maxInflight), C.uint64_t(maxBytes)) }); err != nil {
			b.Close()
			return nil, err
		}
	}
	return b, nil
}

func (b *BatchEngine) Close() { C.rafthip_destroy(b.e); b.e = nil }

// TickBuffer is page-locked memory for the records of one tick: the kernels read
// it over the link themselves (DESIGN.md, "Acks over PCIe").
func TickBuffer[T any](n int) ([]T, func(), error) {
	var p unsafe.Pointer
	var zero T
	if err := guarded(func() C.int { return C.rafthip_host_alloc(&p, C.uint64_t(uintptr(n)*unsafe.Sizeof(zero))) }); err != nil {
		return nil, nil, err
	}
	return unsafe.Slice((*T)(p), n), func() { C.rafthip_host_free(p) }, nil
}

// SetConfig: raft.switchToConfig (raft.go:1949) for one group; masks are slot bitmasks.
func (b *BatchEngine) SetConfig(group uint64, incoming, outgoing, learners uint64) error {
	return guarded(func() C.int { return C.rafthip_upload_config(b.e, C.uint64_t(group), 1, unsafe.Pointer(&incoming), unsafe.Pointer(&outgoing),
		unsafe.Pointer(&learners)) }) // little-endian: the low rafthip_mask_bytes() bytes are read
}

// ScatterAcks: Progress.MaybeUpdate's Match part for a tick's acks (progress.go:168-177).
func (b *BatchEngine) ScatterAcks(acks []Ack) error {
	if len(acks) == 0 {
		return nil
	}
	return guarded(func() C.int { return C.rafthip_scatter_acks(b.e, &acks[0], C.uint64_t(len(acks))) })
}

// Step: stepLeader for a tick's responses and send records (raft.go:1333-1586).
// aux[i] is nextProbeIdx of a rejection / payload bytes of a SENT_APP record;
// actions[i] tells which send the reference performs next (RAFTHIP_ACT_*).
func (b *BatchEngine) Step(msgs []Msg, aux []uint64, actions []uint8) error {
	if len(msgs) == 0 {
		return nil
	}
	return guarded(func() C.int { return C.rafthip_step_progress(b.e, &msgs[0], (*C.uint64_t)(unsafe.Pointer(&aux[0])), C.uint64_t(len(msgs)),
		(*C.uint8_t)(unsafe.Pointer(&actions[0]))) })
}

// StepWire: the same from serialized raftpb.Message payloads (message i =
// bytes[offsets[i]:offsets[i+1]] for group groups[i]); status[i] is RAFTHIP_WIRE_*.
func (b *BatchEngine) StepWire(bytes []byte, offsets []uint64, groups []uint32, status, actions []uint8) error {
	if len(groups) == 0 {
		return nil
	}
	return guarded(func() C.int { return C.rafthip_step_wire(b.e, (*C.uint8_t)(unsafe.Pointer(&bytes[0])), (*C.uint64_t)(unsafe.Pointer(&offsets[0])),
		(*C.uint32_t)(unsafe.Pointer(&groups[0])), C.uint64_t(len(groups)), nil, (*C.uint8_t)(unsafe.Pointer(&status[0])), nil,
		(*C.uint8_t)(unsafe.Pointer(&actions[0]))) })
}

// Campaign: group starts (respType = MsgPreVoteResp or MsgVoteResp, stepCandidate's myVoteRespType,
// raft.go:1628-1633) or ends (0) polling vote responses; becomePreCandidate / becomeCandidate reset
// the votes (raft.go:784 via reset, :893-896).  StepWire then records the group's vote responses.
func (b *BatchEngine) Campaign(group uint64, respType uint8) error {
	if respType != 0 {
		if err := guarded(func() C.int { return C.rafthip_reset_votes(b.e, C.uint64_t(group), 1) }); err != nil {
			return err
		}
	}
	return guarded(func() C.int { return C.rafthip_upload_campaign(b.e, C.uint64_t(group), 1, (*C.uint8_t)(unsafe.Pointer(&respType))) })
}

// TallyVotes: ProgressTracker.TallyVotes for every group (tracker/tracker.go:267-290); result[g] is
// quorum.VoteResult (1 pending, 2 lost, 3 won).
func (b *BatchEngine) TallyVotes(result []uint8) error {
	if err := guarded(func() C.int { return C.rafthip_tally_votes(b.e) }); err != nil {
		return err
	}
	return guarded(func() C.int { return C.rafthip_read_vote_result(b.e, 0, C.uint64_t(len(result)), (*C.uint8_t)(unsafe.Pointer(&result[0])), nil, nil) })
}

// MaybeCommit: raft.maybeCommit for every group (raft.go:755-758, log.go:447-456);
// returns the groups whose raftLog.committed moved, sorted by group.
func (b *BatchEngine) MaybeCommit(pairs []CommitPair) (int, error) {
	var n C.uint64_t
	var p *CommitPair
	if len(pairs) > 0 {
		p = &pairs[0]
	}
	err := guarded(func() C.int { return C.rafthip_commit_advance(b.e, p, C.uint64_t(len(pairs)), &n) })
	if int(n) > len(pairs) {
		n = C.uint64_t(len(pairs))
	}
	return int(n), err
}

// CheckQuorum: the MsgCheckQuorum step for every group (raft.go:1231-1243);
// active[g] == 0 means group g steps down.
func (b *BatchEngine) CheckQuorum(active []uint8) error {
	if err := guarded(func() C.int { return C.rafthip_check_quorum(b.e) }); err != nil {
		return err
	}
	return guarded(func() C.int { return C.rafthip_read_quorum_active(b.e, 0, C.uint64_t(len(active)), (*C.uint8_t)(unsafe.Pointer(&active[0]))) })
}

// Tick actions and state flags (rafthip.h RAFTHIP_TICK_*, RAFTHIP_TICKF_*).
const (
	TickHup           = 1  // tickElection fired: r.Step(MsgHup), raft.go:826-831
	TickCheckQuorum   = 2  // MsgCheckQuorum has been stepped on the device, raft.go:1231-1243
	TickStepDown      = 4  // ... and the quorum was not active: the rest of becomeFollower -> reset is the host's
	TickAbortTransfer = 8  // raft.go:846-849
	TickBeat          = 16 // r.Step(MsgBeat): bcastHeartbeat, raft.go:856-861
	TickPromotable    = 1  // r.promotable(), raft.go:1883-1886
	TickTransfer      = 2  // r.leadTransferee != None
)

// TickEvent is one group of the last Tick with one of the actions asked for.
type TickEvent struct {
	Group  uint32
	Action uint32
}

// SetTickConfig: ElectionTick, HeartbeatTick, CheckQuorum of raft.Config for every group (raft.go:128-138,222-224;
// refused as Config.validate refuses them, raft.go:296-302).
func (b *BatchEngine) SetTickConfig(electionTick, heartbeatTick int, checkQuorum bool) error {
	cq := C.int(0)
	if checkQuorum {
		cq = 1
	}
	return guarded(func() C.int { return C.rafthip_tick_config(b.e, C.uint32_t(electionTick), C.uint32_t(heartbeatTick), cq) })
}

// SetTickState: what reset / becomeLeader / becomeFollower / a leadership transfer leave in group g:
// r.electionElapsed, r.heartbeatElapsed, r.randomizedElectionTimeout, r.state and the two flags.
func (b *BatchEngine) SetTickState(g uint64, electionElapsed, heartbeatElapsed, randomizedTimeout uint32, state uint8, flags uint8) error {
	return guarded(func() C.int {
		return C.rafthip_upload_tick_state(b.e, C.uint64_t(g), 1, (*C.uint32_t)(&electionElapsed), (*C.uint32_t)(&heartbeatElapsed),
			(*C.uint32_t)(&randomizedTimeout), (*C.uint8_t)(&state), (*C.uint8_t)(&flags))
	})
}

// TickMany: Tick for the shards (engines) of one GPU in one launch (rafthip_tick_many); counts[i] belongs to
// engines[i].  The handles are C pointers (as in CommitMany), so the array may be Go memory.
func TickMany(engines []*BatchEngine, counts [][5]uint64) error {
	hs := make([]*C.rafthip_engine, len(engines))
	for i, b := range engines {
		hs[i] = b.e
	}
	var p *C.uint64_t
	if len(counts) > 0 {
		p = (*C.uint64_t)(unsafe.Pointer(&counts[0][0]))
	}
	return guarded(func() C.int { return C.rafthip_tick_many(&hs[0], C.uint32_t(len(hs)), p) })
}

// HeardFromLeader: stepFollower's r.electionElapsed = 0 (raft.go:1681-1692) for the groups whose leader sent a MsgApp,
// MsgHeartbeat or MsgSnap of the current term since the last Tick.
func (b *BatchEngine) HeardFromLeader(groups []uint32) error {
	if len(groups) == 0 {
		return nil
	}
	return guarded(func() C.int {
		return C.rafthip_tick_heard_from_leader(b.e, (*C.uint32_t)(unsafe.Pointer(&groups[0])), C.uint64_t(len(groups)), nil)
	})
}

// Tick: RawNode.Tick for every group (rawnode.go:64-66 -> tickElection / tickHeartbeat, raft.go:823-862) in one
// pass.  counts[i] = groups whose action has bit i (Hup, CheckQuorum, StepDown, AbortTransfer, Beat); events receives
// the groups with an action of mask, in group order, and the call returns how many there are.
func (b *BatchEngine) Tick(mask uint32, events []TickEvent, counts *[5]uint64) (int, error) {
	if err := guarded(func() C.int { return C.rafthip_tick(b.e, (*C.uint64_t)(unsafe.Pointer(&counts[0]))) }); err != nil {
		return 0, err
	}
	var n C.uint64_t
	var p *C.rafthip_tick_event
	if len(events) > 0 {
		p = (*C.rafthip_tick_event)(unsafe.Pointer(&events[0]))
	}
	err := guarded(func() C.int { return C.rafthip_read_tick_events(b.e, C.uint32_t(mask), p, C.uint64_t(len(events)), &n) })
	return int(n), err
}

// HeartbeatsDue: Heartbeats for the groups that stepped MsgBeat in the last Tick only (raft.go:856-861).
func (b *BatchEngine) HeartbeatsDue(bytes, lens []uint8) error {
	return guarded(func() C.int { return C.rafthip_emit_heartbeats_due(b.e, 0, C.uint64_t(b.groups), (*C.uint8_t)(unsafe.Pointer(&bytes[0])),
		(*C.uint8_t)(unsafe.Pointer(&lens[0]))) })
}

// HeartbeatCommit: sendHeartbeat's commit per peer (raft.go:669-685), [slot][group].
func (b *BatchEngine) HeartbeatCommit(dst []uint64) error {
	return guarded(func() C.int { return C.rafthip_heartbeat_commit(b.e, 0, C.uint64_t(b.groups), (*C.uint64_t)(unsafe.Pointer(&dst[0])),
		C.uint64_t(b.groups)) })
}

// Heartbeats: bcastHeartbeat as wire bytes (raft.go:698-716, Message.Marshal): the MsgHeartbeat to
// slot s of group g is bytes[(s*groups+g)*72:][:lens[s*groups+g]]; length 0 = none to send.
func (b *BatchEngine) Heartbeats(bytes, lens []uint8) error {
	return guarded(func() C.int { return C.rafthip_emit_heartbeats(b.e, 0, C.uint64_t(b.groups), (*C.uint8_t)(unsafe.Pointer(&bytes[0])),
		(*C.uint8_t)(unsafe.Pointer(&lens[0]))) })
}

// PlanAppends: bcastAppend as a plan (raft.go:689-696): plan[s*groups+g] is what maybeSendAppend(to, sendIfEmpty)
// decides for the peer in slot s of group g (C.RAFTHIP_SEND_*), next[s*groups+g] its pr.Next.
func (b *BatchEngine) PlanAppends(sendIfEmpty bool, plan []uint8, next []uint64) error {
	sie := C.int(0)
	if sendIfEmpty {
		sie = 1
	}
	return guarded(func() C.int { return C.rafthip_plan_appends(b.e, 0, C.uint64_t(b.groups), sie, (*C.uint8_t)(unsafe.Pointer(&plan[0])),
		(*C.uint64_t)(unsafe.Pointer(&next[0])), C.uint64_t(b.groups)) })
}

// CommitMany: ProgressTracker.Committed for the shards several engines hold on ONE GPU, as one kernel
// launch (rafthip_commit_many): the shards do not each pay a kernel boundary.
func CommitMany(engines []*BatchEngine) error {
	hs := make([]*C.rafthip_engine, len(engines))
	for i, b := range engines {
		hs[i] = b.e
	}
	return guarded(func() C.int { return C.rafthip_commit_many(&hs[0], C.uint32_t(len(hs))) })
}

// MaybeCommitMany is MaybeCommit (advance) or the changed-pair report for several
// shards of one GPU in one call.  pairs[i] must be memory from TickBuffer (C
// memory: cgo forbids handing C an array of Go pointers), room[i] its capacity in
// pairs; counts[i] receives shard i's pair count.
func MaybeCommitMany(engines []*BatchEngine, advance bool, pairs []*CommitPair, room []uint64, counts []uint64) error {
	hs := make([]*C.rafthip_engine, len(engines))
	ds := make([]*C.rafthip_commit_pair, len(engines))
	for i, b := range engines {
		hs[i] = b.e
		ds[i] = (*C.rafthip_commit_pair)(unsafe.Pointer(pairs[i]))
	}
	adv := C.int(0)
	if advance {
		adv = 1
	}
	return guarded(func() C.int {
		return C.rafthip_commit_report_many(&hs[0], C.uint32_t(len(hs)), adv, &ds[0],
			(*C.uint64_t)(unsafe.Pointer(&room[0])), (*C.uint64_t)(unsafe.Pointer(&counts[0])))
	})
}

// Exchange is one rank of the path's only collective (rafthip_comm_*): after the commit pass the
// changed (or advanced) pairs of every GPU of the node are all-gathered over RCCL, so that the host
// loop that feeds raft.maybeCommit (raft.go:755-758) and Ready's HardState.Commit (rawnode.go:141-156)
// sees the whole node's set.  One Exchange per engine; ranks are goroutines locked to OS threads, one
// per GPU, each calling Gather in the same tick.
type Exchange struct {
	c      *C.rafthip_comm
	nranks int
}

// ExchangeID draws the communicator id on one rank; hand it to the others.
func ExchangeID() (id C.rafthip_comm_id, err error) {
	err = guarded(func() C.int { return C.rafthip_comm_unique_id(&id) })
	return
}

// NewExchange is collective over the nranks ranks.  groupBase is the first global group id of this
// engine's shard (shards are contiguous group-id ranges, ascending with rank).
func NewExchange(b *BatchEngine, id C.rafthip_comm_id, rank, nranks int, groupBase uint64) (*Exchange, error) {
	x := &Exchange{nranks: nranks}
	err := guarded(func() C.int {
		return C.rafthip_comm_create(b.e, &id, C.int(rank), C.int(nranks), C.uint64_t(groupBase), 0, &x.c)
	})
	if err != nil {
		return nil, err
	}
	return x, nil
}

func (x *Exchange) Close() { C.rafthip_comm_destroy(x.c); x.c = nil }

// Gather: raft.maybeCommit on this rank's engine (advance) or the plain changed-commit pass, then the
// all-gather: pairs receives the pairs of ALL ranks with global group ids, sorted by group; counts[r]
// is what rank r reported.
func (x *Exchange) Gather(advance bool, pairs []CommitPair, counts []uint64) (int, error) {
	var total C.uint64_t
	adv := C.int(0)
	if advance {
		adv = 1
	}
	var p *CommitPair
	if len(pairs) > 0 {
		p = &pairs[0]
	}
	err := guarded(func() C.int {
		return C.rafthip_commit_allgather(x.c, adv, p, C.uint64_t(len(pairs)), (*C.uint64_t)(unsafe.Pointer(&counts[0])), &total)
	})
	return int(total), err
}

// DenseWidth is the number of plane entries every rank contributes to GatherDense.
func (x *Exchange) DenseWidth() (int, error) {
	var w C.uint64_t
	err := guarded(func() C.int { return C.rafthip_commit_allgather_dense(x.c, 0, nil, 0, &w) })
	return int(w), err
}

// GatherDense is Gather for ticks in which most groups move: the commit pass, then
// one all-gather of the commit planes.  planes holds nranks*DenseWidth() entries;
// planes[r*width+g] is the commit index of rank r's group g.
func (x *Exchange) GatherDense(advance bool, planes []uint64) error {
	adv := C.int(0)
	if advance {
		adv = 1
	}
	return guarded(func() C.int {
		return C.rafthip_commit_allgather_dense(x.c, adv, (*C.uint64_t)(unsafe.Pointer(&planes[0])), C.uint64_t(len(planes)), nil)
	})
}
