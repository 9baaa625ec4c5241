"""Soak of the two frame loops on long drives, against the oracle:
  odometry   ll_odometry_frames (laserOdometry.cpp:439-832: 3 outer iterations x Ceres LM, vote from frame 6, warm start from the
             previous frame's result) over N consecutive scans: every frame's relative pose within 1e-6 of the oracle's loop, ATE
             against the generator's ground truth within 1 % of the CPU path's (whatever that is on the generated scene)
  mapping    the same frames through the cube map, free running, fed device-to-device from the slots
             (ll_cubemap_process_slot, laserMapping.cpp:1584-2165): every frame's refined pose within 1e-6 of the oracle's
on two data shapes: the synthetic 64-ring drive and the HDL-64E true laser table in KITTI order (ring capacity 4608).
usage: soak_frames.py [frames per shape, default 60] [mapping frames per shape, default min(frames, 40)]"""
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lightloam_amd  # noqa: F401,E402
from lightloam_amd import api, synth, hdl64  # noqa: E402
from oracle import orc  # noqa: E402
from test_gpu_odometry import ate, integrate  # noqa: E402

api.load_library(); orc.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
NM = int(sys.argv[2]) if len(sys.argv) > 2 else min(N, 40)
t00 = time.time()
for name, kind in (("synthetic 64-ring drive", "synth"), ("HDL-64E table, KITTI order", "hdl64")):
    if kind == "synth":
        cfg = synth.default_cfg(64)
        scans = [synth.scan(cfg, k) for k in range(N)]
        gt = np.array([synth.pose(cfg, k) for k in range(N)])
        pose0 = np.array([0, 0, 0, 1.0, 0.9, 0.0, 0.0]); extra = {}
    else:
        scans = [hdl64.hdl64_scan(k, order="kitti") for k in range(N)]
        gt = np.array([hdl64.pose(k) for k in range(N)])
        pose0 = np.array([0, 0, 0, 1.0, 1.0, 0.0, 0.0]); extra = {"max_ring_points": 4608}
    P = orc.params(64)
    ex = [orc.extract(s, P) for s in scans]
    ctx = api.Context(api.default_params(64, batch=N, max_points=max(map(len, scans)) + 7, **extra))
    for k, s in enumerate(scans):
        ctx.upload_scan(k, s)
    ctx.extract(0, N)
    # ---- odometry frame loop
    orc.set_nn_mode(1)
    q = pose0[:4].copy(); t = pose0[4:].copy(); rel_o = []
    for k in range(1, N):
        q, t = orc.odometry_frame(q, t, ex[k], ex[k - 1], vote=k > 5)
        rel_o.append(np.concatenate([q, t]))
    orc.set_nn_mode(0)
    rel_o = np.array(rel_o)
    ctx.set_target_from_slot(0)
    rel_d = ctx.odometry_frames(1, N - 1, pose0=pose0, n_outer=3, first_frame_index=1)
    err = np.abs(rel_d - rel_o).max(axis=1)
    assert err.max() < 1e-6, (name, "odometry", int(err.argmax()) + 1, float(err.max()))
    c, s_ = np.cos(gt[0, 2]), np.sin(gt[0, 2])
    gt_xy = (gt[:, :2] - gt[0, :2]) @ np.array([[c, -s_], [s_, c]])
    ate_o, ate_d = ate(integrate(rel_o), gt_xy), ate(integrate(rel_d), gt_xy)
    travelled = float(np.linalg.norm(np.diff(gt_xy, axis=0), axis=1).sum())
    # (how good the trajectory is depends on the generated scene -- the street of the HDL-64E generator ends after ~150 m -- and is
    # not what this checks: the device's ATE has to be the CPU path's)
    assert abs(ate_d - ate_o) <= 0.01 * ate_o + 1e-9, (name, ate_o, ate_d, travelled)
    print(f"{name}: odometry {N - 1} frames ok, worst |pose - oracle| {err.max():.2e}, ATE {ate_d:.4f} m over {travelled:.1f} m (oracle {ate_o:.4f})", flush=True)
    # ---- mapping, free running: the guess of frame k is the ground truth offset the way odometry drift would hand it over
    dc = api.CubeMap(ctx, 64 * 120 + 64, 200000, pool_points=1 << 21)
    oc = orc.CubeMap()
    worst = 0.0
    for k in range(NM):
        x, y, yaw = gt[k]
        g = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2), x + 0.06, y - 0.04, 0.02])
        if k > 0:
            g[4:] += [-0.11, 0.11, -0.03]
        f = ex[k]
        oc.prepare(g[4:], f["less_sharp"], f["less_flat"])
        qo, to, ran_o = oc.optimize(g[:4], g[4:]); oc.update(qo, to)
        pose, ran_d = dc.process_slot(g, k)
        assert ran_d == ran_o, (name, "mapping", k, ran_d, ran_o)
        e = max(float(np.abs(pose[:4] - qo).max()), float(np.abs(pose[4:] - to).max()))
        assert e < 1e-6, (name, "mapping", k, e)
        worst = max(worst, e)
    dc.close(); oc.close(); ctx.close()
    print(f"{name}: mapping {NM} frames ok, worst |pose - oracle| {worst:.2e}", flush=True)
print(f"frame-loop soak passed: 2 x {N - 1} odometry frames, 2 x {NM} mapping frames ({time.time() - t00:.0f} s)")
