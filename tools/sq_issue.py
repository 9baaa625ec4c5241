#!/usr/bin/env python3
"""Instruction-issue counters per kernel from a rocprofv3 --pmc pass of bench.py, as the stamped file bench.py's roofline.issue reads.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS ... --kernel-trace -f csv -d gpurun_out/pmc_sq -o sq -- python3 bench.py ...
    python tools/sq_issue.py gpurun_out/pmc_sq --batch 16384 --rings 64 --workload synthetic > profiles/sq_issue.json

Per kernel: wave-instructions per FULL-BATCH launch (launches of at least half the kernel's largest grid), averaged.  Stamped with the
library's source digest, ring count, workload and batch like profiles/pmc_traffic.json (bench.py uses it only on a match)."""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def kernel_name(raw):
    n = raw.split("(")[0].replace("void ", "").split("<")[0].strip()
    return "k_ring_pick" if n.startswith("k_ring_pick") else n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir"); ap.add_argument("--batch", type=int, required=True); ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--workload", default="synthetic"); ap.add_argument("--source-digest", default=None)
    args = ap.parse_args()
    if args.source_digest is None:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        args.source_digest = bench.source_digest()
    # per INSTANTIATION first (k_ring_features<9, true> and <12, true> are different launches of one step: the tiers of long rings),
    # then summed under the kernel's name
    inst = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(args.dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_name(r["Kernel_Name"]).startswith("k_"):
                inst[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Counter_Name"]].append((float(r["Counter_Value"]), float(r["Grid_Size"])))
    rows = defaultdict(lambda: defaultdict(float)); launches = defaultdict(int)
    for raw, ctrs in inst.items():
        n = kernel_name(raw)
        for c, lst in ctrs.items():
            gmax = max(g for _, g in lst)
            big = [v for v, g in lst if g >= 0.5 * gmax]
            rows[n][c] += sum(big) / len(big)
            launches[raw] = len(big)
    if not rows:
        raise SystemExit(f"no counter_collection.csv with k_* kernels under {args.dir}")
    out = {"rings": args.rings, "batch": args.batch, "workload": args.workload, "source_digest": args.source_digest,
           "unit": "wave-instructions per step (full-batch launches; the tier launches of one kernel summed)", "kernels": {}}
    names = {"SQ_INSTS_VALU": "valu", "SQ_INSTS_SALU": "salu", "SQ_INSTS_LDS": "lds", "SQ_INSTS_VMEM_RD": "vmem_rd", "SQ_WAVES": "waves",
             "SQ_WAVE_CYCLES": "wave_cycles_x4", "SQ_BUSY_CYCLES": "busy_cycles", "SQ_WAIT_INST_ANY": "wait_inst_any_x4"}
    for k in sorted(rows):
        e = {names.get(c, c): v for c, v in rows[k].items()}
        e["instantiations"] = sorted(r for r in launches if kernel_name(r) == k)
        out["kernels"][k] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
