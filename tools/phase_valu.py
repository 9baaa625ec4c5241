#!/usr/bin/env python3
"""Instruction counts per phase of k_ring_features: builds copies of the library whose kernel returns after phase i
(-DLL_PHASE_STOP=i) and runs the extract stage; run every build under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_LDS` and difference the per-launch counters (tools/sq_summary.py).

    python tools/phase_valu.py <stop>          # stop = 0..5, or 9 for the full kernel
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
stop = int(sys.argv[1])
out = os.path.join(ROOT, os.environ.get("LL_PHASE_DIR", "gpurun_out"), f"liblightloam_hip_stop{stop}.so")   # LL_PHASE_DIR=_phase: pre-built on the CPU box, travels with the snapshot
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["LIGHTLOAM_HIP_LIB"] = out
import lightloam_amd  # noqa: E402,F401
from lightloam_amd import build, api, synth  # noqa: E402

if not os.path.exists(out):
    build.build_hip(force=True, extra_flags=[] if stop == 9 else [f"-DLL_PHASE_STOP={stop}"])
if len(sys.argv) > 2 and sys.argv[2] == "build":
    sys.exit(0)
B = 256
cfg = synth.default_cfg(64)
scans = [synth.scan(cfg, k) for k in range(4)]
ctx = api.Context(api.default_params(64, batch=B, max_points=max(map(len, scans))))
for i in range(B):
    ctx.upload_scan(i, scans[i % 4])
ctx.extract(0, B)
ctx.synchronize()
