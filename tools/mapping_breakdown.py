#!/usr/bin/env python3
"""Where a laserMapping frame's time goes at the C-ABI level (prepare / optimize / update), medians over the second half of a
synthetic S64 drive.  Prints one JSON line (milliseconds)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightloam_amd  # noqa: E402,F401
from lightloam_amd import api, synth  # noqa: E402
from tools.bench_mapping import pose7  # noqa: E402

rings, frames = 64, 40
cfg = synth.default_cfg(rings)
scans = [synth.scan(cfg, k) for k in range(frames)]
ctx = api.Context(api.default_params(rings, batch=frames, max_points=max(map(len, scans))))
for k, s in enumerate(scans):
    ctx.upload_scan(k, s)
ctx.extract(0, frames)
feats = [ctx.features(k) for k in range(frames)]
cm = api.CubeMap(ctx, rings * 120 + 64, 400000, pool_points=1 << 22)
T = {"prepare": [], "optimize": [], "update": []}
for k in range(frames):
    g = pose7(synth.pose(cfg, k)); g[4:] += [0.05, -0.03, 0.01]
    t0 = time.perf_counter(); cm.prepare(g[4:], feats[k]["less_sharp"], feats[k]["less_flat"])
    t1 = time.perf_counter(); p, ran = cm.optimize(g)
    t2 = time.perf_counter(); cm.update(p)
    t3 = time.perf_counter()
    T["prepare"].append((t1 - t0) * 1e3); T["optimize"].append((t2 - t1) * 1e3); T["update"].append((t3 - t2) * 1e3)
print(json.dumps({k: round(float(np.median(v[frames // 2:])), 4) for k, v in T.items()}))
