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


def _counter_csv(path, rows):
    import csv
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Start_Timestamp", "End_Timestamp"])
        for r in rows:
            w.writerow(r)


def test_counter_tools_sum_the_tier_launches_and_stamp_their_output(tmp_path):
    """tools/sq_issue.py / tools/pmc_traffic.py: the kernels a step launches several times (k_ring_pick6 / 8 / 12, k_ring_features<9, true> /
    <12, true>: the tiers of long rings) count as ONE kernel per step, set-up launches of one scan are left out, and the output carries the
    stamp bench.py matches (source digest, ring count, workload, batch)."""
    full = 64 * 8192
    d = tmp_path / "sq"; d.mkdir()
    rows = []
    for _ in range(3):                                                          # three steps
        rows += [["k_ring_pick6(LLView, int, int, int, int)", "SQ_INSTS_VALU", 1000, full, 0, 10],
                 ["k_ring_pick8(LLView, int, int, int, int)", "SQ_INSTS_VALU", 200, full, 0, 10],
                 ["void k_ring_features<9, true>(LLView, int, int, int, int)", "SQ_INSTS_VALU", 3000, 256 * full, 0, 10],
                 ["void k_ring_features<12, true>(LLView, int, int, int, int)", "SQ_INSTS_VALU", 500, 256 * full, 0, 10],
                 ["k_ring_pick6(LLView, int, int, int, int)", "SQ_INSTS_SALU", 700, full, 0, 10]]
    rows.append(["k_ring_pick6(LLView, int, int, int, int)", "SQ_INSTS_VALU", 7, 64, 0, 1])      # the carry scan's launch
    _counter_csv(d / "x_counter_collection.csv", rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sq_issue.py"), str(d), "--batch", "8192", "--workload", "hdl64", "--source-digest", "abc"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    T = json.loads(out.stdout)
    assert (T["batch"], T["rings"], T["workload"], T["source_digest"]) == (8192, 64, "hdl64", "abc")
    assert T["kernels"]["k_ring_pick"]["valu"] == 1200 and T["kernels"]["k_ring_pick"]["salu"] == 700
    assert T["kernels"]["k_ring_features"]["valu"] == 3500
    # the traffic tool on the same shape of table (+ the calibration launch it scales by)
    f = tmp_path / "fetch"; w = tmp_path / "write"; f.mkdir(); w.mkdir()
    kib = 1 << 20                                                               # 1 GiB in KiB
    _counter_csv(f / "a_counter_collection.csv", [["k_calib_copy(float4 const*, float4*, unsigned long)", "FETCH_SIZE", kib / 2, 1024, 0, 1],
                                                  ["void k_ring_features<9, true>(LLView, int, int, int, int)", "FETCH_SIZE", 4000, full, 0, 1],
                                                  ["void k_ring_features<12, true>(LLView, int, int, int, int)", "FETCH_SIZE", 1000, full, 0, 1]])
    _counter_csv(w / "a_counter_collection.csv", [["k_calib_copy(float4 const*, float4*, unsigned long)", "WRITE_SIZE", kib, 1024, 0, 1],
                                                  ["void k_ring_features<9, true>(LLView, int, int, int, int)", "WRITE_SIZE", 800, full, 0, 1]])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), str(f), str(w), "--batch", "8192", "--workload", "hdl64", "--source-digest", "abc"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    P = json.loads(out.stdout)
    k = P["kernels"]["k_ring_features"]
    assert abs(P["calibration"]["fetch_reported_per_true"] - 0.5) < 1e-12      # the gfx950 FETCH_SIZE caveat, corrected by the known-byte launch
    assert abs(k["hbm_read_bytes_per_launch"] - 2 * 5000 * 1024) < 1e-6 and abs(k["hbm_write_bytes_per_launch"] - 800 * 1024) < 1e-6


def test_bench_uses_a_counter_file_only_on_a_matching_stamp(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    args = type("A", (), dict(rings=64, batch=16384, workload="synthetic"))()
    good = dict(source_digest=bench.source_digest(), rings=64, batch=16384, workload="synthetic",
                kernels={"k_ring_pick": dict(valu=1.0, salu=2.0, lds=3.0), "k_ring_features": dict(valu=4.0, salu=5.0, lds=6.0)})
    p = tmp_path / "sq_issue.json"
    p.write_text(json.dumps(good))
    got, why = bench.issue_from_profile(args, ["k_ring_pick", "k_ring_features"], str(p))
    assert got is not None and got["k_ring_features"]["valu"] == 4.0
    for key, val in (("source_digest", "0" * 16), ("batch", 8192), ("workload", "hdl64"), ("rings", 128)):
        p.write_text(json.dumps(dict(good, **{key: val})))
        got, why = bench.issue_from_profile(args, ["k_ring_pick"], str(p))
        assert got is None and key in why
