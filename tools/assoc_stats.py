#!/usr/bin/env python3
"""What k_associate's two searches visit: needs a library built with -DLL_ASSOC_STATS
(tools/make_ab_variant.sh stats -DLL_ASSOC_STATS, then LL_STATS_LIB=_ab/stats.so).  Runs the bench workload's hot path over
a batch and prints, for the corner and the plane queries: candidate points scanned per query by the K=1 search and by the
ring-window search, and sweep rounds (bound fetch -> scan -> share) per query.
usage: assoc_stats.py [batch] [synthetic|hdl64] [rings]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LIGHTLOAM_HIP_LIB"] = os.path.abspath(os.environ["LL_STATS_LIB"])
import bench  # noqa: E402
import lightloam_amd  # noqa: E402,F401
from lightloam_amd import api  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wl = sys.argv[2] if len(sys.argv) > 2 else "synthetic"
R = int(sys.argv[3]) if len(sys.argv) > 3 else 64
args = argparse.Namespace(workload=wl, rings=R, distinct=8, batch=B)
base, order, guesses = bench.build_workload(args, B, 0)
extra = {"max_ring_points": 4608} if wl == "hdl64" else {}
extra.update(bench.ring_model_params(args))
ctx = api.Context(api.default_params(R, batch=B + 1, max_points=max(map(len, base)), **extra))
ctx.upload_scan(B, base[order[0]])
ctx.extract(B, 1)
ctx.set_target_from_slot(B)
for i in range(B):
    ctx.upload_scan(i, base[order[i + 1]])
ctx.set_pose_guess(0, B, guesses)
buf = (C.c_ulonglong * 16)()
ctx.hot_path(0, B, None, vote=True)
ctx._ck(ctx.lib.ll_debug_counters(ctx.h, buf, 1))
ctx.hot_path(0, B, None, vote=True)
ctx._ck(ctx.lib.ll_debug_counters(ctx.h, buf, 1))
print(f"{os.path.basename(os.environ['LL_STATS_LIB'])}  workload {wl}  batch {B}")
for name, b in (("corners", 0), ("planes", 4)):
    q = max(1, buf[b + 3])
    fb = buf[12 + b // 4]
    print(f"  {name:8s} queries {buf[b + 3]:9d}  K=1 candidates/query {buf[b] / q:7.1f}  window candidates/query {buf[b + 1] / q:7.1f}  sweep rounds/query {buf[b + 2] / q:5.2f}"
          f"  queries whose ring table was not usable (tie / window outside) {fb / 8 / q:6.3f}")
