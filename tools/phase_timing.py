#!/usr/bin/env python3
"""Per-phase shader cycles inside k_ring_features: builds an instrumented copy of the library (-DLL_PHASE_TIMING)
under gpurun_out/, runs a batch of S64 scans through it and prints the average cycles per workgroup per phase."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
prebuilt = os.environ.get("LL_PHASE_LIB")           # an instrumented library built elsewhere (tools/make_ab_variant.sh x -DLL_PHASE_TIMING)
out = prebuilt or os.path.join(ROOT, "gpurun_out", "liblightloam_hip_phase.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
os.environ["LIGHTLOAM_HIP_LIB"] = out
import lightloam_amd  # noqa: E402,F401
from lightloam_amd import build, api, synth  # noqa: E402

if not prebuilt:
    build.build_hip(force=True, extra_flags=["-DLL_PHASE_TIMING"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = synth.default_cfg(64)
scans = [synth.scan(cfg, k) for k in range(4)]
ctx = api.Context(api.default_params(64, batch=B, max_points=max(map(len, scans))))
for i in range(B):
    ctx.upload_scan(i, scans[i % 4])
ctx.extract(0, B)
buf = (C.c_ulonglong * 16)()
ctx._ck(ctx.lib.ll_debug_counters(ctx.h, buf, 1))
ctx.extract(0, B)
ctx._ck(ctx.lib.ll_debug_counters(ctx.h, buf, 1))
names = [(3, "lists + bitmap + loads + bounds + voxel keys"), (4, "voxel sort"), (12, "gather + run heads + scan"), (13, "centroid sums (own range)"),
         (5, "runs continued into later threads"), (7, "centroid stores + list gather + offsets"), (6, "small-list stores")]
n = max(1, buf[15])
tot = sum(buf[i] for i, _ in names)
print(f"k_ring_features phase timing over {buf[15]} workgroups (s_memtime cycles per workgroup, thread 0 wall):")
for i, nm in names:
    print(f"  {nm:40s} {buf[i] / n:10.0f}  {100.0 * buf[i] / max(1, tot):5.1f}%")
print(f"  {'total':40s} {tot / n:10.0f}")
g = max(1, buf[14])
print(f"k_build_grid phase timing over {buf[14]} workgroups:")
for i, nm in ((8, "zero + histogram"), (9, "scan + cell starts"), (10, "scatter to cell order"), (11, "ring tables + validity")):
    print(f"  {nm:32s} {buf[i] / g:10.0f}")
