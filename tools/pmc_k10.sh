set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r3d_k10pmc
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $out/sq -- python3 tools/bench_kernels.py --only "(K10" --iters 3 > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r3d_k10pmc/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_emit_heartbeats" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: round(sum(x)/len(x)) for c,x in acc.items()})
PY
rm -rf $out/sq
