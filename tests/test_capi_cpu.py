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


def test_create_validates_its_config_before_touching_a_device():
    """Misuse is an error return with a message, never an abort (the reference panics)."""
    import ctypes as C
    L = raft_amd.load_library()
    h = C.c_void_p()

    def create(size=None, groups=16, slots=3, flags=0):
        cfg = capi._Config(C.sizeof(capi._Config) if size is None else size, 0, groups, slots, flags, None)
        rc = L.rafthip_create(C.byref(cfg), C.byref(h))
        return rc, L.rafthip_last_error().decode()

    for kw, word in ((dict(size=8), "struct_size"), (dict(groups=0), "num_groups"),
                     (dict(groups=1 << 32), "32-bit group id"), (dict(slots=0), "num_slots"),
                     (dict(slots=65), "num_slots"), (dict(flags=32), "flags"), (dict(flags=6), "exclude each other"),
                     (dict(flags=24), "exclude each other")):
        rc, msg = create(**kw)
        assert rc == -22 and word in msg, (kw, rc, msg)
        assert h.value is None
    # NULL engine on every entry point that takes one
    for name in ("rafthip_synchronize", "rafthip_tally_votes", "rafthip_quorum_active", "rafthip_ack_quorum",
                 "rafthip_check_quorum"):
        assert getattr(L, name)(None) == -22, name
    assert L.rafthip_mask_bytes(None) == -22
    assert L.rafthip_destroy(None) == 0  # like free(NULL)
    n = C.c_uint64(7)
    assert L.rafthip_commit_changed(None, None, 0, C.byref(n)) == -22
    assert L.rafthip_commit_advance(None, None, 0, C.byref(n)) == -22
    assert L.rafthip_ingest_msgs(None, None, 0) == -22
    assert L.rafthip_batch_committed_index(0, 4, 3, None, None, None, None) == -22


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C99 on its own."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "rafthip.h"\nint main(void) { return (int)sizeof(rafthip_msg) - 16; }\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
                               "-c", src, "-o", os.path.join(d, "t.o")])


def test_header_is_plain_c_and_the_smoke_program_calls_every_entry_point(tmp_path):
    """include/rafthip.h compiles as C11 with warnings on, tools/abi_smoke.c (the program the ASan pass runs on
    the GPU box) links against the in-tree library, and it calls every declared function."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = os.path.join(ROOT, "tools", "abi_smoke.c")
    out = str(tmp_path / "abi_smoke")
    libdir = os.path.dirname(capi.library_path())
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-L", libdir,
                        "-lrafthip", "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--unresolved-symbols=ignore-in-shared-libs", "-o", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(src).read()
    missing = [s for s in capi.SYMBOLS if not re.search(r"\b%s\(" % s, text)]
    assert not missing, missing
    # the plain-C latency probe of DESIGN.md section 4 ("a small tick from host memory") builds the same way
    probe = os.path.join(ROOT, "tools", "small_tick_probe.c")
    r = subprocess.run(["gcc", "-O2", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), probe,
                        "-L", libdir, "-lrafthip", "-Wl,-rpath-link,/opt/rocm/lib",
                        "-Wl,--unresolved-symbols=ignore-in-shared-libs", "-o", str(tmp_path / "small_tick_probe")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
