package quorum

import (
	"math/rand"
	"testing"
)

// TestBatchMatchesPerGroup is the test a maintainer runs where a Go toolchain
// and an MI355X exist: the batched entry points must equal looping the
// per-group API on random configs shaped like quick_test.go:47-64.
func TestBatchMatchesPerGroup(t *testing.T) {
	rng := rand.New(rand.NewSource(1))
	const g = 10000
	cfgs := make([]JointConfig, g)
	ls := make([]AckedIndexer, g)
	votes := make([]map[uint64]bool, g)
	for i := range cfgs {
		cfgs[i] = JointConfig{MajorityConfig{}, MajorityConfig{}}
		l := mapAckIndexer{}
		v := map[uint64]bool{}
		for h := 0; h < 2; h++ {
			if h == 1 && rng.Intn(2) == 0 {
				continue
			}
			for k, n := 0, rng.Intn(6); k < n; k++ {
				id := uint64(rng.Intn(9) + 1)
				cfgs[i][h][id] = struct{}{}
				if rng.Intn(4) != 0 {
					l[id] = Index(rng.Intn(10))
				}
				if rng.Intn(3) != 0 {
					v[id] = rng.Intn(2) == 0
				}
			}
		}
		ls[i], votes[i] = l, v
	}
	ci, err := BatchCommittedIndex(cfgs, ls)
	if err != nil {
		t.Fatal(err)
	}
	vr, err := BatchVoteResult(cfgs, votes)
	if err != nil {
		t.Fatal(err)
	}
	for i := range cfgs {
		if want := cfgs[i].CommittedIndex(ls[i]); ci[i] != want {
			t.Fatalf("group %d: CommittedIndex %v, batch %v", i, want, ci[i])
		}
		if want := cfgs[i].VoteResult(votes[i]); vr[i] != want {
			t.Fatalf("group %d: VoteResult %v, batch %v", i, want, vr[i])
		}
	}
}
