#!/usr/bin/env python3
"""Latency of the node-style path (one scan at a time, host in / host out), the way ros/lightloam_*_node.cpp drive the
library: registration = ll_upload_scan + ll_extract_batch + the five downloads; odometry = ll_upload_features +
ll_odometry_frames (3 outer x LM) + ll_set_target_from_slot.  Prints one JSON line (milliseconds, medians over the frames)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightloam_amd  # noqa: E402,F401
from lightloam_amd import api, synth  # noqa: E402

rings = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = synth.default_cfg(rings)
scans = [synth.scan(cfg, k) for k in range(nframes)]
reg = api.Context(api.default_params(rings, batch=2, max_points=max(map(len, scans))))
odo = api.Context(api.default_params(rings, batch=2, max_points=max(map(len, scans))))
t_reg, t_odo = [], []
guess = np.array([0, 0, 0, 1.0, 0.9, 0, 0])
for k, s in enumerate(scans):
    t0 = time.perf_counter()
    reg.upload_scan(0, s)
    reg.extract(0, 1)
    cloud = reg.cloud(0)[0]
    f = reg.features(0)
    t1 = time.perf_counter()
    odo.upload_features(0, f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
    if k > 0:
        guess = odo.odometry_frames(0, 1, pose0=guess, n_outer=3, first_frame_index=k)[0]
    odo.set_target_from_slot(0)
    odo.synchronize()
    t2 = time.perf_counter()
    if k >= 3:
        t_reg.append((t1 - t0) * 1e3); t_odo.append((t2 - t1) * 1e3)
print(json.dumps({"rings": rings, "frames": len(t_reg), "points_per_scan": int(len(scans[0])),
                  "registration_ms": {"median": float(np.median(t_reg)), "p90": float(np.percentile(t_reg, 90))},
                  "odometry_ms": {"median": float(np.median(t_odo)), "p90": float(np.percentile(t_odo, 90))},
                  "note": "host in / host out per scan through the ctypes binding (pinned staging inside the library); "
                          "the reference: scan registration ~27 ms, odometry tens of ms per frame on one CPU core (SURVEY.md)"}))
