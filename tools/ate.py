#!/usr/bin/env python3
"""Absolute trajectory error between two KITTI-format pose files (one line per frame: the 12 values of the 3x4 matrix
[R | t], row-major -- what laserMapping.cpp:2306-2325 and lightloam::TrajectoryWriter write, and the format of the KITTI
odometry ground truth).

    python tools/ate.py <estimate.txt> <ground_truth.txt> [--align first|rigid|none]

  first  (default) both trajectories are re-expressed relative to their first pose (the reference's result file already is)
  rigid  least-squares rotation + translation of the estimate onto the ground truth (Horn / Umeyama without scale)
  none   compare as written
Prints one JSON line: frames, path length of the ground truth, ATE rmse / mean / max in metres and rmse as a fraction of
the path length -- the number BASELINE.json's "ATE within 1 % of reference" clause compares between two estimators.
"""
import argparse
import json
import sys

import numpy as np


def load(path):
    a = np.loadtxt(path, ndmin=2)
    if a.shape[1] != 12:
        raise SystemExit(f"{path}: expected 12 values per line, got {a.shape[1]}")
    T = np.tile(np.eye(4), (len(a), 1, 1))
    T[:, :3, :] = a.reshape(-1, 3, 4)
    return T


def relative_to_first(T):
    return np.linalg.inv(T[0])[None] @ T


def rigid_align(est, ref):
    """R, t minimising sum |R est_i + t - ref_i|^2 (no scale)."""
    mu_e, mu_r = est.mean(axis=0), ref.mean(axis=0)
    H = (est - mu_e).T @ (ref - mu_r)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    return R, mu_r - R @ mu_e


def ate(est_T, ref_T, align="first"):
    n = min(len(est_T), len(ref_T))
    est_T, ref_T = est_T[:n], ref_T[:n]
    if align == "first":
        est_T, ref_T = relative_to_first(est_T), relative_to_first(ref_T)
    e, r = est_T[:, :3, 3], ref_T[:, :3, 3]
    if align == "rigid":
        R, t = rigid_align(e, r)
        e = e @ R.T + t
    d = np.linalg.norm(e - r, axis=1)
    path = float(np.linalg.norm(np.diff(r, axis=0), axis=1).sum()) if n > 1 else 0.0
    rmse = float(np.sqrt(np.mean(d * d)))
    return dict(frames=int(n), path_length_m=path, ate_rmse_m=rmse, ate_mean_m=float(d.mean()), ate_max_m=float(d.max()),
                ate_rmse_over_path=(rmse / path if path > 0 else None), align=align)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("estimate"); ap.add_argument("ground_truth")
    ap.add_argument("--align", choices=["first", "rigid", "none"], default="first")
    a = ap.parse_args(argv)
    print(json.dumps(ate(load(a.estimate), load(a.ground_truth), a.align)))


if __name__ == "__main__":
    sys.exit(main())
