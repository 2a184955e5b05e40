# Scalar / vector / LDS instruction counts per kernel of the tick (one rocprofv3 --pmc pass over tools/tick_bench.py):
# a kernel whose scalar instructions per CU approach its duration in cycles runs at the scalar unit's issue rate.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/pmc_salu
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $out/sq -- python3 tools/tick_bench.py > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_salu/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "rafthip" in k:
            acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()):
    m={c: sum(x)/len(x) for c,x in v.items()}
    w=max(m.get("SQ_WAVES",1),1)
    print("%-42s waves %8d  per wave: SALU %6.0f VALU %6.0f LDS %5.0f   SALU per CU %8.0f  busy cycles/32 %8.0f" % (k, w, m["SQ_INSTS_SALU"]/w, m["SQ_INSTS_VALU"]/w, m["SQ_INSTS_LDS"]/w, m["SQ_INSTS_SALU"]/256, m["SQ_BUSY_CYCLES"]/32))
PY
rm -rf $out/sq
