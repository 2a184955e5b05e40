#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small CSVs kept under profiles/.

  summarize_prof.py stats  <dir> <out.csv>          --kernel-trace --stats run: per-kernel calls / avg / total
  summarize_prof.py pmc    <out.csv> <dir> [<dir>…]  --pmc runs: per (counter, kernel) dispatches and avg/min/max KB
  summarize_prof.py traffic <pmc_summary.csv> <kernel substring> <groups> <bytes/group> <out.json>
                                                     HBM bytes per launch of one kernel, with the gfx950
                                                     FETCH_SIZE x2 correction (MI355X_MICROARCH.md, HBM section)
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, suffix):
    return sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))


def stats(d, out):
    files = find(d, "kernel_stats.csv")
    if not files:
        raise SystemExit("no *kernel_stats.csv under %s" % d)
    rows = list(csv.DictReader(open(files[-1])))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r.get("MinNs", ""), r.get("MaxNs", "")])


def pmc(out, dirs):
    acc = defaultdict(list)
    for d in dirs:
        for f in find(d, "counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                acc[(r["Counter_Name"], r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Counter", "Kernel", "GridSize", "Dispatches", "AvgKB", "MinKB", "MaxKB"])
        for (c, k, g), v in sorted(acc.items()):
            w.writerow([c, k, g, len(v), sum(v) / len(v), min(v), max(v)])


def traffic(summary, kernel, groups, bytes_per_group, out):
    import json
    fetch = write = None
    for r in csv.DictReader(open(summary)):
        if kernel in r["Kernel"]:
            if r["Counter"] == "FETCH_SIZE":
                fetch = float(r["AvgKB"])
            elif r["Counter"] == "WRITE_SIZE":
                write = float(r["AvgKB"])
    if fetch is None or write is None:
        raise SystemExit("kernel %r not found with both counters in %s" % (kernel, summary))
    doc = {
        "kernel": kernel, "groups_per_launch": groups, "fetch_size_kb_avg": fetch, "write_size_kb_avg": write,
        "fetch_bytes_corrected": fetch * 1024 * 2, "write_bytes": write * 1024,
        "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
        "algorithmic_bytes_per_launch": groups * bytes_per_group,
        "hbm_bytes_per_group": (fetch * 1024 * 2 + write * 1024) / groups,
        "algorithmic_bytes_per_group": bytes_per_group,
        "note": "FETCH_SIZE (KB) x1024 x2: on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced "
                "streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE (KB) x1024 is exact for "
                "16-B/lane stores.  Separate --pmc passes.",
        "source": summary,
    }
    json.dump(doc, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
    else:
        pmc(sys.argv[2], sys.argv[3:])
