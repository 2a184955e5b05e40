"""The compile-time sorting networks of raft_amd/csrc/sortnet.h, checked on the
host (they replace insertionSort, quorum/majority.go:115-122)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_batcher_networks_sort(tmp_path):
    exe = str(tmp_path / "sortnet_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "raft_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tests", "sortnet_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout
    assert "Batcher<64>: 543 comparators, ok" in out.stdout
