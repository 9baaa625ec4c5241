#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters in a rocprofv3 --pmc counter_collection.csv (largest launches only)."""
import csv, glob, os, sys
from collections import defaultdict


def kernel_name(raw):
    """k_foo<9, true>(LLView, ...) -> k_foo; the per-row-count kernels k_ring_pick6 / 8 / 12 / 22 -> k_ring_pick (bench.py's name)"""
    n = raw.split("(")[0].replace("void ", "").split("<")[0].strip()
    return "k_ring_pick" if n.startswith("k_ring_pick") else n


d = sys.argv[1]
rows = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = kernel_name(r["Kernel_Name"])
        if name.startswith("k_"):
            rows[name][r["Counter_Name"]].append((float(r["Counter_Value"]), float(r["Grid_Size"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
for name in sorted(rows):
    out = []
    for c in sorted(rows[name]):
        lst = rows[name][c]; gmax = max(g for _, g, _ in lst)
        big = [(v, t) for v, g, t in lst if g >= 0.5 * gmax]
        out.append("%s=%.4g" % (c, sum(v for v, _ in big) / len(big)))
    t = [t for v, g, t in next(iter(rows[name].values())) if g >= 0.5 * max(g for _, g, _ in next(iter(rows[name].values())))]
    print("%-22s avg_ns=%.0f  " % (name, sum(t) / len(t)) + "  ".join(out))
