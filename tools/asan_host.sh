#!/bin/bash
# AddressSanitizer pass over the HOST code of librafthip.so and the C++ host
# mirror (GPU code is not instrumented: GPU ASan is unavailable on this pool).
# Builds into tools/bin/asan/ and runs the host-mirror GPU tests through the
# instrumented CLI and tools/abi_smoke.c through the instrumented library.  Usage (on a GPU box): tools/asan_host.sh
set -e -o pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/bin/asan
mkdir -p "$out"
if [ "$1" = "build" ] || [ ! -x "$out/rafthost_cli" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -Xarch_host -fsanitize=address \
      -I"$root/include" -I"$root/raft_amd/csrc" -o "$out/librafthip.so" "$root/raft_amd/csrc/rafthip.hip" \
      -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address -I"$root/include" -I"$root/raft_amd/host" \
      -o "$out/rafthost_cli" "$root"/raft_amd/host/{hostcli,quorum_batch,tracker_batch}.cpp \
      -L"$out" -lrafthip -Wl,-rpath,'$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib
  # every entry point of include/rafthip.h once, from plain C, with exactly-sized heap buffers
  /opt/rocm/lib/llvm/bin/clang -O1 -g -std=c11 -fsanitize=address -I"$root/include" -o "$out/abi_smoke" \
      "$root/tools/abi_smoke.c" -L"$out" -lrafthip -Wl,-rpath,'$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib
fi
[ "$1" = "build" ] && exit 0
cd "$root"
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 timeout -k 10 300 "$out/abi_smoke"
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 RAFTHOST_CLI="$out/rafthost_cli" \
  timeout -k 10 600 python3 -u -m pytest tests/test_host_mirror.py -m gpu -x -q --timeout 300 --timeout-method thread
