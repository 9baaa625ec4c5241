"""the reference's own solver (ceres::Solve, LM, max 4 iterations) and frame loop
(3 outer iterations, vote from the 6th frame, warm start) on the device, against the oracle's restatement, and the
absolute trajectory error of both against the synthetic ground truth (north_star: ATE within 1 % of the CPU path)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def qmul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def qrot(q, v):
    u, w = q[:3], q[3]
    uv = 2 * np.cross(u, v)
    return v + w * uv + np.cross(u, uv)


def integrate(rel):
    """t_w += q_w * t ; q_w = q_w * q   (laserOdometry.cpp:830-831)"""
    qw = np.array([0, 0, 0, 1.0]); tw = np.zeros(3); out = [tw.copy()]
    for p in rel:
        tw = tw + qrot(qw, p[4:]); qw = qmul(qw, p[:4]); out.append(tw.copy())
    return np.array(out)


def ate(traj, gt_xy):
    return float(np.sqrt(np.mean(np.sum((traj[:, :2] - gt_xy) ** 2, axis=1))))


@pytest.mark.parametrize("rings,nframes", [(16, 14), (64, 9)])
def test_sequence_matches_oracle_and_ground_truth(api, orc, synth, rings, nframes):
    cfg = synth.default_cfg(rings)
    scans = [synth.scan(cfg, k) for k in range(nframes)]
    P = orc.params(rings)
    ex = [orc.extract(s, P) for s in scans]
    # ---- oracle frame loop
    orc.set_nn_mode(1)
    # the reference warm-starts every frame from the previous motion and real sequences start at rest; the synthetic one
    # starts at full speed, so both paths get the same first guess near the true first motion
    pose0 = np.array([0, 0, 0, 1.0, 0.9, 0.0, 0.0])
    q = pose0[:4].copy(); t = pose0[4:].copy(); rel_o = []
    for k in range(1, nframes):
        q, t = orc.odometry_frame(q, t, ex[k], ex[k - 1], vote=k > 5)
        rel_o.append(np.concatenate([q, t]))
    orc.set_nn_mode(0)
    rel_o = np.array(rel_o)
    # ---- device frame loop: slots 0..n-1 hold the scans, slot 0 is the first target
    ctx = api.Context(api.default_params(rings, batch=nframes, max_points=max(map(len, scans))))
    for k, s in enumerate(scans):
        ctx.upload_scan(k, s)
    ctx.extract(0, nframes)
    ctx.set_target_from_slot(0)
    rel_d = ctx.odometry_frames(1, nframes - 1, pose0=pose0, n_outer=3, first_frame_index=1)
    ctx.close()
    # frame-by-frame agreement (same algorithm; f64 rounding only, unless a discrete decision flips -- none may here)
    assert np.abs(rel_d - rel_o).max() < 1e-6, np.abs(rel_d - rel_o).max(axis=1)
    # trajectories against the synthetic ground truth
    gt = np.array([synth.pose(cfg, k) for k in range(nframes)])
    c, s_ = np.cos(gt[0, 2]), np.sin(gt[0, 2])
    gt_xy = (gt[:, :2] - gt[0, :2]) @ np.array([[c, -s_], [s_, c]])
    ate_o, ate_d = ate(integrate(rel_o), gt_xy), ate(integrate(rel_d), gt_xy)
    travelled = float(np.linalg.norm(np.diff(gt_xy, axis=0), axis=1).sum())
    assert ate_o < 0.02 * travelled, (ate_o, travelled)           # the odometry itself works: < 2 % drift on this stretch
    assert abs(ate_d - ate_o) <= 0.01 * ate_o + 1e-9              # north_star: ATE within 1 % of the CPU path


def test_lm_solve_parity_single_pair(api, orc, synth):
    cfg = synth.default_cfg(16)
    scans = [synth.scan(cfg, k) for k in range(2)]
    P = orc.params(16)
    e0, e1 = orc.extract(scans[0], P), orc.extract(scans[1], P)
    pose = np.array([0.0, 0.0, 0.0, 1.0, 0.5, 0.1, 0.0])           # a poor guess: several LM iterations, some may be rejected
    q, t = pose[:4], pose[4:]
    es, ea, eb = orc.associate_corner(q, t, e1["sharp"], e0["less_sharp"])
    ps, pa, pb, pc = orc.associate_plane(q, t, e1["flat"], e0["less_flat"])
    cnt, sidx, sw = orc.vote(e1["flat"][ps], e0["less_flat"][pa])
    order = np.sort(sidx); wmap = np.ones(len(ps), np.float32); wmap[sidx] = sw
    qo, to, summ = orc.lm_solve(q, t, e1["sharp"], es, e0["less_sharp"], ea, eb, e1["flat"], ps[order], e0["less_flat"],
                                pa[order], pb[order], pc[order], wmap[order])
    ctx = api.Context(api.default_params(16, batch=2, max_points=max(map(len, scans))))
    ctx.upload_scan(0, scans[0]); ctx.upload_scan(1, scans[1])
    ctx.extract(0, 2)
    ctx.set_target_from_slot(0)
    ctx.associate(1, 1, pose)
    ctx.vote(1, 1, True)
    ctx.lm_solve(1, 1)
    got = ctx.pose(1)
    ctx.close()
    assert summ[2] >= 1 and summ[1] < summ[0]
    assert np.abs(got[:4] - qo).max() < 1e-9 and np.abs(got[4:] - to).max() < 1e-9


def test_lm_options_are_the_ceres_defaults(api):
    o = api.LmOptions()
    api.load_library().ll_lm_default_options(__import__("ctypes").byref(o))
    assert (o.max_num_iterations, o.initial_radius, o.min_relative_decrease, o.jacobi_scaling) == (4, 1e4, 1e-3, 1)
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (1e-6, 1e-10, 1e-8)


def test_forty_frame_s64_drive_ate(api, orc, synth):
    """north_star's accuracy clause on a longer KITTI-shape drive: 40 frames (36 m, a 4 deg turn), device odometry vs the
    oracle's, both against the synthetic ground truth."""
    rings, nframes = 64, 40
    cfg = synth.default_cfg(rings)
    scans = [synth.scan(cfg, k) for k in range(nframes)]
    P = orc.params(rings)
    ex = [orc.extract(s, P) for s in scans]
    pose0 = np.array([0, 0, 0, 1.0, 0.9, 0.0, 0.0])
    orc.set_nn_mode(1)
    q = pose0[:4].copy(); t = pose0[4:].copy(); rel_o = []
    for k in range(1, nframes):
        q, t = orc.odometry_frame(q, t, ex[k], ex[k - 1], vote=k > 5)
        rel_o.append(np.concatenate([q, t]))
    orc.set_nn_mode(0)
    ctx = api.Context(api.default_params(rings, batch=nframes, max_points=max(map(len, scans))))
    for k, s in enumerate(scans):
        ctx.upload_scan(k, s)
    ctx.extract(0, nframes)
    ctx.set_target_from_slot(0)
    rel_d = ctx.odometry_frames(1, nframes - 1, pose0=pose0, n_outer=3, first_frame_index=1)
    ctx.close()
    gt = np.array([synth.pose(cfg, k) for k in range(nframes)])
    c, s_ = np.cos(gt[0, 2]), np.sin(gt[0, 2])
    gt_xy = (gt[:, :2] - gt[0, :2]) @ np.array([[c, -s_], [s_, c]])
    ate_o, ate_d = ate(integrate(np.array(rel_o)), gt_xy), ate(integrate(rel_d), gt_xy)
    travelled = float(np.linalg.norm(np.diff(gt_xy, axis=0), axis=1).sum())
    assert travelled > 30 and ate_o < 0.02 * travelled, (ate_o, travelled)
    assert abs(ate_d - ate_o) <= 0.01 * ate_o + 1e-9, (ate_d, ate_o)              # ATE within 1 % of the CPU path
    assert np.abs(rel_d - np.array(rel_o)).max() < 1e-5


def test_features_uploaded_from_host_give_the_same_frames(api, synth):
    """A separate laserOdometry process gets the four feature clouds by topic, not from an extracted slot:
    ll_upload_features + ll_odometry_frames on one slot + ll_set_target_from_slot per frame must reproduce the relative
    poses of the all-on-device sequence bit for bit (same kernels, same clouds)."""
    rings, nframes = 16, 9
    cfg = synth.default_cfg(rings)
    scans = [synth.scan(cfg, k) for k in range(nframes)]
    pose0 = np.array([0, 0, 0, 1.0, 0.9, 0.0, 0.0])
    reg = api.Context(api.default_params(rings, batch=nframes, max_points=max(map(len, scans))))
    for k, s in enumerate(scans):
        reg.upload_scan(k, s)
    reg.extract(0, nframes)
    reg.set_target_from_slot(0)
    rel_a = reg.odometry_frames(1, nframes - 1, pose0=pose0, n_outer=3, first_frame_index=1)
    feats = [reg.features(k) for k in range(nframes)]
    reg.close()
    odo = api.Context(api.default_params(rings, batch=2, max_points=max(map(len, scans))))
    odo.set_target(feats[0]["less_sharp"], feats[0]["less_flat"])          # the first frame only initialises (:426-430)
    guess = pose0.copy(); rel_b = []
    for k in range(1, nframes):
        f = feats[k]
        odo.upload_features(0, f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
        guess = odo.odometry_frames(0, 1, pose0=guess, n_outer=3, first_frame_index=k)[0]      # warm start (:61-65)
        rel_b.append(guess.copy())
        odo.set_target_from_slot(0)                                         # the pointer swap + kd-tree rebuild (:882-896)
    odo.close()
    assert np.array_equal(np.array(rel_b), rel_a)


def test_uploaded_slots_serve_as_targets_of_the_next_slot(api, synth):
    """ll_upload_features into slots 0..k and ONE ll_odometry_frames over the whole range: slot j-1 is the target of slot
    j, so every uploaded slot needs its own search grid and ring tables (built by ll_upload_features itself, not only by
    the extract / set_target paths).  Same relative poses as the all-on-device sequence, bit for bit."""
    rings, nframes = 16, 7
    cfg = synth.default_cfg(rings)
    scans = [synth.scan(cfg, k) for k in range(nframes)]
    pose0 = np.array([0, 0, 0, 1.0, 0.9, 0.0, 0.0])
    reg = api.Context(api.default_params(rings, batch=nframes, max_points=max(map(len, scans))))
    for k, s in enumerate(scans):
        reg.upload_scan(k, s)
    reg.extract(0, nframes)
    reg.set_target_from_slot(0)
    rel_a = reg.odometry_frames(1, nframes - 1, pose0=pose0, n_outer=3, first_frame_index=1)
    feats = [reg.features(k) for k in range(nframes)]
    reg.close()
    odo = api.Context(api.default_params(rings, batch=nframes, max_points=max(map(len, scans))))
    # stale grids on purpose: the slots first hold OTHER clouds (the scans in reverse order), then the real ones
    for k in range(nframes):
        f = feats[nframes - 1 - k]
        odo.upload_features(k, f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
    for k in range(nframes):
        f = feats[k]
        odo.upload_features(k, f["sharp"], f["less_sharp"], f["flat"], f["less_flat"])
    odo.set_target_from_slot(0)
    rel_b = odo.odometry_frames(1, nframes - 1, pose0=pose0, n_outer=3, first_frame_index=1)
    # and the staged path: associate over a range of uploaded slots gives the same correspondences as on extracted slots
    odo.close()
    assert np.array_equal(rel_b, rel_a)
