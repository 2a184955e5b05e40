#!/bin/bash
# One GPU-box call for the leader-side tick (tools/tick_bench.py): the tick with the Progress stepper in its
# partitioned form (K9p, the default at this size) and pinned to K9's global lists, rocprofv3 kernel stats of the
# first, FETCH / WRITE PMC passes, and the K8 / K9 / K9p rows of tools/bench_kernels.py.
# Usage: tools/profile_tick.sh <tag>; writes gpurun_out/<tag>_*; copy what you keep into profiles/.
set -e -o pipefail
tag=${1:-r03_b}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/tick_bench.py > $out/${tag}_tick_bench.json 2> $out/${tag}_tick_bench.err
RAFTHIP_STEP_LINKED=1 timeout -k 10 300 python3 tools/tick_bench.py > $out/${tag}_tick_bench_linked.json 2> $out/${tag}_tick_bench_linked.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_tick_prof -- python3 tools/tick_bench.py > $out/${tag}_tick_prof.json 2> $out/${tag}_tick_prof.log
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_tick_pmc_fetch -- python3 tools/tick_bench.py > $out/${tag}_tick_pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_tick_pmc_write -- python3 tools/tick_bench.py > $out/${tag}_tick_pmc_write.log 2>&1
python3 tools/summarize_prof.py stats $out/${tag}_tick_prof $out/${tag}_tick_kernel_stats.csv
python3 tools/summarize_prof.py pmc $out/${tag}_tick_pmc_summary.csv $out/${tag}_tick_pmc_fetch $out/${tag}_tick_pmc_write
timeout -k 10 300 python3 -u tools/bench_kernels.py --only "(K" --iters 10 > $out/${tag}_k789.txt 2>&1
rm -rf $out/${tag}_tick_prof $out/${tag}_tick_pmc_fetch $out/${tag}_tick_pmc_write
cut -c1-300 $out/${tag}_tick_bench.json
