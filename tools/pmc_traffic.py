#!/usr/bin/env python3
"""HBM traffic per kernel from rocprofv3 PMC passes of `bench.py --calibrate`.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d gpurun_out/pmc_fetch -o fetch -- python3 bench.py --calibrate ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d gpurun_out/pmc_write -o write -- python3 bench.py --calibrate ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --batch 1024 --rings 64 > profiles/pmc_traffic.json

Separate passes (TCC has 4 PMC slots: FETCH_SIZE costs 3, WRITE_SIZE 2).  Both counters are in KiB of L2 <-> fabric
traffic.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read, and
other widths / WRITE_SIZE are uncalibrated -- so every absolute number here is scaled by the factor measured on the
known-byte launch k_calib_copy (1 GiB float4 read + 1 GiB float4 write) of the same process.
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict



def kernel_name(raw):
    """k_foo<9, true>(LLView, ...) -> k_foo; the per-row-count kernels k_ring_pick6 / 8 / 12 / 22 -> k_ring_pick (bench.py's name)"""
    n = raw.split("(")[0].replace("void ", "").split("<")[0].strip()
    return "k_ring_pick" if n.startswith("k_ring_pick") else n


def per_kernel(dirpath, counter):
    files = glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {dirpath}")
    inst = defaultdict(list)                   # per instantiation (template arguments kept): the tier launches of one kernel are summed below
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            inst[row["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append((float(row["Counter_Value"]), float(row.get("Grid_Size", 0) or 0)))
    tot = defaultdict(float); calls = defaultdict(int)
    for raw, lst in inst.items():
        name = kernel_name(raw)
        gmax = max(g for _, g in lst)
        big = [v for v, g in lst if g >= 0.5 * gmax]     # only full-batch launches (the carry / set-up launches cover one scan)
        # per-step figure of this instantiation = its per-launch average; tot / calls below is again "per launch of the kernel"
        tot[name] += sum(big) / len(big); calls[name] = 1
    return tot, calls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir"); ap.add_argument("write_dir")
    ap.add_argument("--batch", type=int, required=True); ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--calib-bytes", type=float, default=float(1 << 30))
    ap.add_argument("--workload", default="synthetic")
    ap.add_argument("--source-digest", default=None, help="bench.py's source_digest() of the library the passes ran (default: computed from this tree)")
    args = ap.parse_args()
    if args.source_digest is None:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        args.source_digest = bench.source_digest()
    fetch, fcalls = per_kernel(args.fetch_dir, "FETCH_SIZE")
    write, wcalls = per_kernel(args.write_dir, "WRITE_SIZE")
    cf = fetch.get("k_calib_copy", 0) / max(1, fcalls.get("k_calib_copy", 1)) * 1024.0
    cw = write.get("k_calib_copy", 0) / max(1, wcalls.get("k_calib_copy", 1)) * 1024.0
    if cf <= 0 or cw <= 0:
        raise SystemExit("k_calib_copy not found: run bench.py with --calibrate")
    kf, kw = args.calib_bytes / cf, args.calib_bytes / cw          # true bytes per reported byte
    out = {"rings": args.rings, "batch": args.batch, "workload": args.workload, "source_digest": args.source_digest, "unit": "bytes",
           "calibration": {"fetch_reported_per_true": 1.0 / kf, "write_reported_per_true": 1.0 / kw,
                           "note": "k_calib_copy: 1 GiB float4 read + 1 GiB float4 write; counters scaled by these factors"},
           "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        if not name.startswith("k_") or name == "k_calib_copy":
            continue
        n = max(fcalls.get(name, 0), wcalls.get(name, 0), 1)
        # launches of the timed region and of warm-up/setup are alike except the single carry extract (batch of 1 scan):
        # per-scan figure = total bytes / total scans processed by that kernel; approximated by launches * batch
        rd = fetch.get(name, 0.0) * 1024.0 * kf / n
        wr = write.get(name, 0.0) * 1024.0 * kw / n
        out["kernels"][name] = {"launches": n, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                                "hbm_bytes_per_scan": (rd + wr) / args.batch}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
