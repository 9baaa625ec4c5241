"""tools/ate.py: the trajectory-error script of SURVEY 8f #4 on trajectories with a known error."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ate", os.path.join(ROOT, "tools", "ate.py"))
ate = importlib.util.module_from_spec(spec); spec.loader.exec_module(ate)


def _traj(n, seed=0):
    rng = np.random.default_rng(seed)
    T = np.tile(np.eye(4), (n, 1, 1))
    yaw = np.cumsum(rng.normal(0.01, 0.003, n))
    T[:, 0, 0] = np.cos(yaw); T[:, 0, 1] = -np.sin(yaw); T[:, 1, 0] = np.sin(yaw); T[:, 1, 1] = np.cos(yaw)
    step = np.stack([np.cos(yaw), np.sin(yaw), np.zeros(n)], 1)
    T[:, :3, 3] = np.cumsum(step, axis=0)
    return T


def _write(path, T):
    np.savetxt(path, T[:, :3, :].reshape(len(T), 12), fmt="%.6e")


def test_ate_of_identical_and_offset_trajectories(tmp_path):
    gt = _traj(200)
    r = ate.ate(gt, gt)
    assert r["ate_rmse_m"] == 0.0 and abs(r["path_length_m"] - 199.0) < 1e-9
    # a constant world-frame shift vanishes under both alignments, not under "none"
    est = gt.copy(); est[:, :3, 3] += [3.0, -2.0, 0.5]
    assert ate.ate(est, gt, "none")["ate_rmse_m"] > 3.0
    assert ate.ate(est, gt, "first")["ate_rmse_m"] < 1e-12
    assert ate.ate(est, gt, "rigid")["ate_rmse_m"] < 1e-9
    # a rigid motion of the whole estimate (rotation about z + shift) is removed by "rigid" and by "first"
    c, s = np.cos(0.7), np.sin(0.7)
    G = np.eye(4); G[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]; G[:3, 3] = [10, 5, -1]
    est = G[None] @ gt
    assert ate.ate(est, gt, "rigid")["ate_rmse_m"] < 1e-9 and ate.ate(est, gt, "first")["ate_rmse_m"] < 1e-9
    # a known per-frame error: every position 0.3 m off along y after the first frame
    est = gt.copy(); est[1:, 1, 3] += 0.3
    r = ate.ate(est, gt, "first")
    assert abs(r["ate_max_m"] - 0.3) < 1e-12 and abs(r["ate_rmse_m"] - 0.3 * np.sqrt(199 / 200)) < 1e-12
    assert abs(r["ate_rmse_over_path"] - r["ate_rmse_m"] / 199.0) < 1e-15


def test_ate_command_line_reads_the_trajectory_file_format(tmp_path):
    gt = _traj(50, 1); est = gt.copy(); est[:, 0, 3] *= 1.01
    _write(tmp_path / "gt.txt", gt); _write(tmp_path / "est.txt", est)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ate.py"), str(tmp_path / "est.txt"), str(tmp_path / "gt.txt")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout)
    assert r["frames"] == 50 and 0 < r["ate_rmse_over_path"] < 0.02
