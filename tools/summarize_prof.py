#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small CSVs kept under profiles/.

  summarize_prof.py stats  <dir> <out.csv>          --kernel-trace --stats run: per-kernel calls / avg / total
  summarize_prof.py pmc    <out.csv> <dir> [<dir>…]  --pmc runs: per (counter, kernel) dispatches and avg/min/max KB
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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3:])
