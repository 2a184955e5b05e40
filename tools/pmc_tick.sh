# SQ / TCP / TCC counter passes (separate rocprofv3 --pmc runs) over tools/tick_bench.py for the decode, the partitioned
# stepper's apply pass and the heartbeat emitter; prints per-kernel averages.  Usage: bash tools/pmc_tick.sh
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r3b_k8pmc
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --output-format csv -d $out/sq -- python3 tools/tick_bench.py > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $out/tcp -- python3 tools/tick_bench.py > $out/tcp.log 2>&1 || { tail -5 $out/tcp.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/tcc -- python3 tools/tick_bench.py > $out/tcc.log 2>&1 || { tail -5 $out/tcc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
for d in ("sq","tcp","tcc"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/r3b_k8pmc/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "k_ingest_wire" in k or "k_prog_part_apply" in k or "k_emit_heartbeats" in k:
                acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(d, k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
rm -rf $out/sq $out/tcp $out/tcc
