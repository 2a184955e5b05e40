"""raft_amd — MI355X-native batched quorum engine for etcd-style Raft.

The product is `lib/librafthip.so` (hand-written HIP for gfx950 behind the C
ABI in include/rafthip.h).  This package is host-side plumbing over that ABI:

  raft_amd.capi      ctypes binding, one method per C entry point
  raft_amd.multigpu  group-range sharding + the all-gather of changed commits

There is no CPU compute path here: without the built library and a gfx950 GPU
every call fails loudly.
"""
from .capi import (Engine, PinnedArray, RafthipError, batch_committed_index, batch_vote_result,  # noqa: F401
                   library_path, load_library, ACK_DTYPE, PAIR_DTYPE, MSG_DTYPE, mask_dtype,
                   MSG_APP_RESP, MSG_HEARTBEAT_RESP, MSG_APPENDED, MSGF_READ_CTX)
