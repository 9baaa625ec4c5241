"""Soak of the stages DOWNSTREAM of the feature clouds through the batch hot path, against the oracle:
N consecutive scans of four data shapes (three settings of the synthetic 64-ring generator, the HDL-64E true laser table in KITTI
order at ring capacity 4608; with LL_SOAK_ALL_SHAPES=1 also 16 / 32 / 128 rings and the HDL-64E table in firing order), every slot with its own randomly perturbed pose guess; for every slot k >= 1 (target = slot k - 1):
  association   (src, a, b[, c]) index tuples of corners and planes        exact
  vote          incompatibility counts, selected set, weights               exact
  H, g, cost    Huber(0.1) normal equations at the guess                    <= 1e-9 (relative to the largest entry)
  GN step       Cholesky solve + manifold Plus -> the slot's pose           <= 1e-9
usage: soak_hot_path.py [scans per shape, default 48] [every, default 1: check every n-th slot]"""
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lightloam_amd  # noqa: F401,E402
from lightloam_amd import api, synth, hdl64  # noqa: E402
from oracle import orc  # noqa: E402

api.load_library(); orc.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 1
TOL = 1e-9


def close(got, want, what):
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    scale = max(1.0, float(np.max(np.abs(want))))
    err = float(np.max(np.abs(got - want))) / scale
    assert err <= TOL, f"{what}: {err:.3e}"
    return err


def rel_guess(a, b, rng):
    """current (pose b) -> previous (pose a) frame transform of planar poses (x, y, yaw), perturbed like a warm start"""
    dyaw = b[2] - a[2]
    c, s = np.cos(a[2]), np.sin(a[2])
    dx, dy = b[0] - a[0], b[1] - a[1]
    t = np.array([c * dx + s * dy, -s * dx + c * dy, 0.0]) * rng.uniform(0.8, 1.1) + rng.normal(0.0, 0.03, 3)
    rv = np.array([rng.normal(0, 0.003), rng.normal(0, 0.003), dyaw * rng.uniform(0.7, 1.2) + rng.normal(0, 0.003)])
    ang = np.linalg.norm(rv)
    q = np.array([0.0, 0.0, 0.0, 1.0]) if ang == 0 else np.concatenate([np.sin(ang / 2) * rv / ang, [np.cos(ang / 2)]])
    return np.concatenate([q, t])


RING_MODEL = {128: dict(ring_model=1, lower_bound=-25.0, up_bound=15.0, minimum_range=0.3)}     # BASELINE config 5's 128 rings: the linear formula of :162
shapes = [("synthetic ring-major", 64, dict(), None),
          ("synthetic azimuth-major, jitter, drops, NaN", 64, dict(order=1, az_jitter_deg=0.4, drop_prob=0.03, emit_nan=1), None),
          ("synthetic ring-major, jitter, 10 % drops", 64, dict(az_jitter_deg=0.7, drop_prob=0.1), None),
          ("HDL-64E table, KITTI order", 64, "kitti", 4608)]
if os.environ.get("LL_SOAK_ALL_SHAPES"):                  # the other ring counts and the raw firing order (the committed long log has them)
    shapes += [("synthetic 16 rings", 16, dict(), None), ("synthetic 32 rings, jitter", 32, dict(az_jitter_deg=0.4), None),
               ("synthetic 128 rings (linear ring model)", 128, dict(), None), ("HDL-64E table, firing order", 64, "firing", 4608)]
rng = np.random.default_rng(int(os.environ.get("LL_SOAK_SEED", "20261002")))   # LL_SOAK_SEED: other guesses, other scans of the streams
total = 0
worst = dict(H=0.0, g=0.0, cost=0.0, pose=0.0)
t00 = time.time()
for name, rings, kw, cap in shapes:
    if isinstance(kw, dict):
        cfg = synth.default_cfg(rings, **kw)
        k0 = int(rng.integers(0, 400))
        scans = [synth.scan(cfg, k0 + k) for k in range(N)]
        poses = [synth.pose(cfg, k0 + k) for k in range(N)]
    else:
        k0 = int(rng.integers(0, 40))
        scans = [hdl64.hdl64_scan(k0 + k, order=kw) for k in range(N)]
        poses = [hdl64.pose(k0 + k) for k in range(N)]
    model = RING_MODEL.get(rings, {})
    P = orc.params(rings, **model)
    extra = dict(model, **({"max_ring_points": cap} if cap else {}))
    ctx = api.Context(api.default_params(rings, batch=N, max_points=max(map(len, scans)) + 7, **extra))
    for k, s in enumerate(scans):
        ctx.upload_scan(k, s)
    guesses = np.zeros((N, 7)); guesses[:, 3] = 1.0
    for k in range(1, N):
        guesses[k] = rel_guess(poses[k - 1], poses[k], rng)
    ctx.extract(0, N)
    ctx.set_target_from_slot(0)                      # slot 0's own target (unused below)
    ctx.set_pose_guess(0, N, guesses)
    ctx.hot_path(1, N - 1, None, vote=True)          # every slot from its stored guess
    ctx.synchronize()
    refs = [orc.extract(s, P) for s in scans]
    orc.set_nn_mode(1)
    checked = 0
    for k in range(1, N, EVERY):
        cur, prev = refs[k], refs[k - 1]
        assert ctx.scan_info(k).status == 0 and cur["rc"] == 0, (name, k, "status")
        q, t = guesses[k, :4], guesses[k, 4:]
        es, ea, eb = orc.associate_corner(q, t, cur["sharp"], prev["less_sharp"])
        ps, pa, pb, pc = orc.associate_plane(q, t, cur["flat"], prev["less_flat"])
        ges, gea, geb = ctx.edge_corr(k); gps, gpa, gpb, gpc = ctx.plane_corr(k)
        for got, want, nm in ((ges, es, "e_src"), (gea, ea, "e_a"), (geb, eb, "e_b"), (gps, ps, "p_src"), (gpa, pa, "p_a"), (gpb, pb, "p_b"), (gpc, pc, "p_c")):
            assert len(got) == len(want) and (got == want).all(), (name, k, nm)
        assert len(es) > 10 and len(ps) > 10, (name, k, "too few correspondences for a meaningful check")
        cnt, sidx, sw = orc.vote(cur["flat"][ps], prev["less_flat"][pa])
        gcnt, gsel, gw = ctx.vote_result(k)
        want_sel = np.zeros(len(cnt), bool); want_sel[sidx] = True
        want_w = np.ones(len(cnt), np.float32); want_w[sidx] = sw
        assert (gcnt == cnt).all() and (gsel == want_sel).all() and (gw[gsel] == want_w[gsel]).all(), (name, k, "vote")
        order = np.sort(sidx)
        Ho, go, co = orc.normal_equations(q, t, cur["sharp"], es, prev["less_sharp"], ea, eb, cur["flat"], ps[order], prev["less_flat"],
                                          pa[order], pb[order], pc[order], want_w[order], 0.1)
        H, g, cost = ctx.normal_equations_result(k)
        worst["H"] = max(worst["H"], close(H, Ho, f"{name} {k} H")); worst["g"] = max(worst["g"], close(g, go, f"{name} {k} g"))
        worst["cost"] = max(worst["cost"], close(cost, co, f"{name} {k} cost"))
        rc, d = orc.gn_solve(Ho, go)
        assert rc == 0, (name, k, "oracle Cholesky")
        qo, to = orc.pose_update(q, t, d)
        worst["pose"] = max(worst["pose"], close(ctx.pose(k), np.concatenate([qo, to]), f"{name} {k} pose"))
        checked += 1
    orc.set_nn_mode(0)
    ctx.close(); total += checked
    print(f"{name}: {checked} scan pairs ok (scans {k0} .. {k0 + N - 1})", flush=True)
print(f"hot-path soak passed: {total} scan pairs; worst relative error H {worst['H']:.2e}  g {worst['g']:.2e}  cost {worst['cost']:.2e}  pose {worst['pose']:.2e}"
      f"  ({time.time() - t00:.0f} s)")
