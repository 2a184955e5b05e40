"""CPU-side checks of the boundary: the library loads, exports every symbol
include/rafthip.h declares, and fails loudly (no fallback) without a GPU."""
import ctypes
import os
import re

import pytest

import raft_amd
from raft_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_binding_agree():
    hdr = open(os.path.join(ROOT, "include", "rafthip.h")).read()
    declared = set(re.findall(r"\b(rafthip_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"rafthip_mask_bytes_for"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)


def test_library_exports_every_symbol():
    L = ctypes.CDLL(capi.library_path())
    for name in capi.SYMBOLS:
        assert hasattr(L, name), name
    assert raft_amd.load_library().rafthip_abi_version() == 1


def test_every_entry_point_cites_the_reference():
    hdr = open(os.path.join(ROOT, "include", "rafthip.h")).read()
    # each compute/state entry point's comment names a reference file:line
    for fn in ("rafthip_commit", "rafthip_tally_votes", "rafthip_quorum_active", "rafthip_ack_quorum",
               "rafthip_scatter_acks", "rafthip_batch_committed_index", "rafthip_batch_vote_result"):
        i = hdr.index("int %s(" % fn)
        block = hdr[hdr.rindex("/*", 0, i):i]
        assert re.search(r"\.go:\d+", block), fn


def test_no_gpu_means_loud_failure():
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(raft_amd.RafthipError) as ei:
        raft_amd.Engine(16, 3)
    assert ei.value.code == -19


def test_argument_validation_without_gpu():
    L = raft_amd.load_library()
    assert L.rafthip_create(None, None) == -22
    assert b"NULL" in L.rafthip_last_error()
    assert L.rafthip_commit(None) == -22
    assert L.rafthip_scatter_acks(None, None, 0) == -22


def test_product_never_touches_the_oracle():
    """The product path must not import/call/link anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raft_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.lower(), (dirpath, f)
