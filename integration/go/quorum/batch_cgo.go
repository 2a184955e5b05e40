package quorum

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

// Device selects the GPU the stateless batch calls run on.
var Device = 0

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

// Go slices of pointer-free element types may be passed to C for the duration
// of the call (cgo pointer rules); librafthip copies them to the device with
// hipMemcpy and keeps no reference.  Index is uint64 and VoteResult is uint8
// with Pending=1, Lost=2, Won=3 (quorum.go:23,48-58) — exactly what the library
// writes, so the result slices are filled in place.

func hipBatchCommittedIndex(g, s int, match []uint64, in, out []byte, res []Index) error {
	return guarded(func() C.int { return C.rafthip_batch_committed_index(C.int(Device), C.uint64_t(g), C.uint32_t(s),
		(*C.uint64_t)(unsafe.Pointer(&match[0])), unsafe.Pointer(&in[0]), unsafe.Pointer(&out[0]),
		(*C.uint64_t)(unsafe.Pointer(&res[0]))) })
}

func hipBatchVoteResult(g, s int, cells []uint8, in, out []byte, res []VoteResult) error {
	return guarded(func() C.int { return C.rafthip_batch_vote_result(C.int(Device), C.uint64_t(g), C.uint32_t(s),
		(*C.uint8_t)(unsafe.Pointer(&cells[0])), unsafe.Pointer(&in[0]), unsafe.Pointer(&out[0]),
		(*C.uint8_t)(unsafe.Pointer(&res[0]))) })
}
