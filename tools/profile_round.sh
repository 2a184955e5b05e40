#!/bin/bash
# One GPU-box call: bench line, rocprofv3 kernel stats and HBM PMC passes for
# bench.py and for tools/bench_kernels.py.  Usage: tools/profile_round.sh <tag>
# Writes everything under gpurun_out/<tag>_*; copy what you keep into profiles/.
set -e -o pipefail
tag=${1:-r03_a}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python3 -u bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
# the driver's own command line, and the same under the profiler: every rafthip_commit_many launch of a run covers the
# same number of shards (bench.py: 5 at --steps 20 --warmup 5, 10 at 500/50, 12 at 60/12), so the profiler's
# per-kernel average is over equal launches; --no-extras keeps the other users of the GPU out of the trace
timeout -k 10 300 python3 -u bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver.json 2> $out/${tag}_bench_driver.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -- python3 bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-extras > $out/${tag}_prof.json 2> $out/${tag}_prof.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof20 -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $out/${tag}_prof20.json 2> $out/${tag}_prof20.log
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -- python3 bench.py --steps 60 --warmup 12 --no-cpu-baseline --no-extras > $out/${tag}_pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -- python3 bench.py --steps 60 --warmup 12 --no-cpu-baseline --no-extras > $out/${tag}_pmc_write.log 2>&1
python3 tools/summarize_prof.py stats $out/${tag}_prof $out/${tag}_kernel_stats.csv
python3 tools/summarize_prof.py stats $out/${tag}_prof20 $out/${tag}_kernel_stats_steps20.csv
python3 tools/summarize_prof.py pmc $out/${tag}_pmc_summary.csv $out/${tag}_pmc_fetch $out/${tag}_pmc_write
python3 tools/summarize_prof.py traffic $out/${tag}_pmc_summary.csv "k_commit_uniform_many<5>" $((12 * 1048576)) 48 $out/${tag}_pmc_commit_uniform5.json
# every kernel at the BASELINE config sizes
timeout -k 10 300 python3 -u tools/bench_kernels.py > $out/${tag}_kernels.txt 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kprof -- python3 tools/bench_kernels.py --iters 10 > $out/${tag}_kprof.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_kpmc_fetch -- python3 tools/bench_kernels.py --iters 3 > $out/${tag}_kpmc_fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_kpmc_write -- python3 tools/bench_kernels.py --iters 3 > $out/${tag}_kpmc_write.log 2>&1
python3 tools/summarize_prof.py stats $out/${tag}_kprof $out/${tag}_kernels_kernel_stats.csv
python3 tools/summarize_prof.py pmc $out/${tag}_kernels_pmc_summary.csv $out/${tag}_kpmc_fetch $out/${tag}_kpmc_write
# the raw per-dispatch traces are large: keep only the summaries
rm -rf $out/${tag}_prof $out/${tag}_prof20 $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_kprof $out/${tag}_kpmc_fetch $out/${tag}_kpmc_write
cat $out/${tag}_bench.json
