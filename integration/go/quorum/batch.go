// Package quorum — batched entry points beside the unchanged per-group API.
//
// This file is meant to be dropped into go.etcd.io/raft/v3/quorum next to
// majority.go / joint.go.  It contains the API and the marshalling only; the
// two hip* functions it calls live in batch_cgo.go and bind librafthip.so
// (include/rafthip.h).  The build image of this repository has no Go
// toolchain, so these files were written to the ABI but not compiled here;
// raft_amd/host/quorum_batch.cpp is the same marshalling in C++ and is tested
// (tests/test_host_mirror.py).
package quorum

import (
	"encoding/binary"
	"fmt"
	"sort"
)

// maxBatchVoters is RAFTHIP_MAX_SLOTS: the widest group the GPU path takes.
const maxBatchVoters = 64

// BatchCommittedIndex returns cfgs[i].CommittedIndex(ls[i]) for every i,
// evaluated on the GPU.  The result equals looping JointConfig.CommittedIndex.
func BatchCommittedIndex(cfgs []JointConfig, ls []AckedIndexer) ([]Index, error) {
	if len(cfgs) != len(ls) {
		return nil, fmt.Errorf("quorum: len(cfgs) %d != len(ls) %d", len(cfgs), len(ls))
	}
	g := len(cfgs)
	if g == 0 {
		return nil, nil
	}
	slots, s, err := assignSlots(cfgs)
	if err != nil {
		return nil, err
	}
	match := make([]uint64, s*g) // [slot][group]
	in, out := make([]uint64, g), make([]uint64, g)
	for i, c := range cfgs {
		for k, id := range slots[i] {
			// a voter the indexer does not know stays 0, as majority.go:149-161 leaves it
			if idx, ok := ls[i].AckedIndex(id); ok {
				match[k*g+i] = uint64(idx)
			}
			if _, ok := c[0][id]; ok {
				in[i] |= 1 << uint(k)
			}
			if _, ok := c[1][id]; ok {
				out[i] |= 1 << uint(k)
			}
		}
	}
	res := make([]Index, g)
	if err := hipBatchCommittedIndex(g, s, match, packMasks(in, s), packMasks(out, s), res); err != nil {
		return nil, err
	}
	return res, nil
}

// BatchVoteResult returns cfgs[i].VoteResult(votes[i]) for every i.
func BatchVoteResult(cfgs []JointConfig, votes []map[uint64]bool) ([]VoteResult, error) {
	if len(cfgs) != len(votes) {
		return nil, fmt.Errorf("quorum: len(cfgs) %d != len(votes) %d", len(cfgs), len(votes))
	}
	g := len(cfgs)
	if g == 0 {
		return nil, nil
	}
	slots, s, err := assignSlots(cfgs)
	if err != nil {
		return nil, err
	}
	cells := make([]uint8, s*g) // [slot][group]: 0 no vote, 1 rejected, 2 granted
	in, out := make([]uint64, g), make([]uint64, g)
	for i, c := range cfgs {
		for k, id := range slots[i] {
			// votes of ids outside the config are never looked at (majority.go:188-197)
			if v, ok := votes[i][id]; ok {
				if v {
					cells[k*g+i] = 2
				} else {
					cells[k*g+i] = 1
				}
			}
			if _, ok := c[0][id]; ok {
				in[i] |= 1 << uint(k)
			}
			if _, ok := c[1][id]; ok {
				out[i] |= 1 << uint(k)
			}
		}
	}
	res := make([]VoteResult, g)
	if err := hipBatchVoteResult(g, s, cells, packMasks(in, s), packMasks(out, s), res); err != nil {
		return nil, err
	}
	return res, nil
}

// assignSlots gives every group its sorted voter ids (slot k = k-th smallest
// id of Voters[0] ∪ Voters[1]) and returns the widest group's size (at least 1).
func assignSlots(cfgs []JointConfig) (slots [][]uint64, s int, err error) {
	slots = make([][]uint64, len(cfgs))
	s = 1
	for i, c := range cfgs {
		ids := make([]uint64, 0, len(c[0])+len(c[1]))
		for id := range c.IDs() {
			ids = append(ids, id)
		}
		sort.Slice(ids, func(a, b int) bool { return ids[a] < ids[b] })
		if len(ids) > maxBatchVoters {
			return nil, 0, fmt.Errorf("quorum: group %d has %d voters; the batch path supports at most %d",
				i, len(ids), maxBatchVoters)
		}
		slots[i] = ids
		if len(ids) > s {
			s = len(ids)
		}
	}
	return slots, s, nil
}

// maskBytes is rafthip_mask_bytes for s slots: 1, 2, 4 or 8.
func maskBytes(s int) int {
	switch {
	case s <= 8:
		return 1
	case s <= 16:
		return 2
	case s <= 32:
		return 4
	default:
		return 8
	}
}

// packMasks narrows one 64-bit slot mask per group to maskBytes(s) little-endian bytes.
func packMasks(m []uint64, s int) []byte {
	w := maskBytes(s)
	b := make([]byte, w*len(m))
	var tmp [8]byte
	for i, v := range m {
		binary.LittleEndian.PutUint64(tmp[:], v)
		copy(b[i*w:(i+1)*w], tmp[:w])
	}
	return b
}
