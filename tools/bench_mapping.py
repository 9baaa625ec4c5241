#!/usr/bin/env python3
"""laserMapping per-frame cost on one MI355X (BASELINE config 4's stage, single GPU): the synthetic S64 drive through
extract -> odometry guess -> ll_cubemap_process, with the oracle's restatement timed beside it on the host.

    python tools/bench_mapping.py [--frames 40] [--rings 64]

Prints one JSON line.  The map grows for the first frames; the figure is the mean over the second half of the run."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightloam_amd  # noqa: E402,F401
from lightloam_amd import api, synth  # noqa: E402


def pose7(p3):
    x, y, yaw = p3
    return np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2), x, y, 0.0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=40); ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    cfg = synth.default_cfg(args.rings)
    scans = [synth.scan(cfg, k) for k in range(args.frames)]
    ctx = api.Context(api.default_params(args.rings, batch=args.frames, max_points=max(map(len, scans))))
    for k, s in enumerate(scans):
        ctx.upload_scan(k, s)
    ctx.extract(0, args.frames)
    feats = [ctx.features(k) for k in range(args.frames)]
    cm = api.CubeMap(ctx, args.rings * 120 + 64, 400000, pool_points=1 << 22)
    guesses = []
    for k in range(args.frames):
        g = pose7(synth.pose(cfg, k)); g[4:] += [0.05, -0.03, 0.01]           # what the odometry hands over
        guesses.append(g)
    t_frame = []
    for k in range(args.frames):
        t0 = time.perf_counter()
        pose, ran = cm.process(guesses[k], feats[k]["less_sharp"], feats[k]["less_flat"])
        t_frame.append(time.perf_counter() - t0)
    _, cnt = cm.info()
    half = args.frames // 2
    gpu_ms = 1e3 * float(np.mean(t_frame[half:]))
    out = {"metric": "laserMapping frames/sec (cube map prepare + 2 x (5-NN association + LM) + map update), 64-ring scans, 1 GPU",
           "value": 1e3 / gpu_ms, "unit": "frames/s", "ms_per_frame": gpu_ms, "frames": args.frames,
           "map_points": {"corner_from_map": cnt[0], "surf_from_map": cnt[1], "corner_stack": cnt[2], "surf_stack": cnt[3]},
           "note": "host-driven, one frame at a time (the stage is sequential: every frame's map depends on the previous pose)"}
    if not args.no_cpu_baseline:
        from oracle import orc
        orc.set_nn_mode(1)
        oc = orc.CubeMap()
        tc = []
        for k in range(min(args.frames, 16)):
            t0 = time.perf_counter()
            oc.prepare(guesses[k][4:], feats[k]["less_sharp"], feats[k]["less_flat"])
            q, t, _ = oc.optimize(guesses[k][:4], guesses[k][4:])
            oc.update(q, t)
            tc.append(time.perf_counter() - t0)
        orc.set_nn_mode(0)
        out["cpu_baseline"] = {"value": 1.0 / float(np.mean(tc[len(tc) // 2:])), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"{len(tc)} frames of the same drive, oracle/ll_oracle.c single thread, grid NN"}
    print(json.dumps(out))
    cm.close(); ctx.close()


if __name__ == "__main__":
    main()
