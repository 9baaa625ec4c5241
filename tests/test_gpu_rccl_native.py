"""north_star: "Host code stays C++/ROS ... RCCL all-reduce of the 6x6 / 6x1 normal equations over xGMI" -- from C++, with no torch
in the process.  tests/native/rccl_normal_equations.cpp links librccl and liblightloam_hip, brings one
communicator up per visible device (ncclCommInitAll: world 1 on the one-GPU pool, N wherever N devices are visible) and runs

  * the row-parallel Levenberg-Marquardt of laserMapping (ncclAllReduce of ll_map_evaluate_dev's 44-double record on the library's
    stream, lightloam::map_optimize_row_parallel) against the one-rank ll_map_optimize, and
  * lightloam::LaserMapping::process_tile_parallel with ncclAllGather as its all_gather against the unsplit LaserMapping::process."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _to_world(pts, pose3):
    x, y, yaw = pose3
    c, s = np.cos(yaw), np.sin(yaw)
    out = pts.astype(np.float64).copy()
    out[:, 0] = c * pts[:, 0] - s * pts[:, 1] + x
    out[:, 1] = s * pts[:, 0] + c * pts[:, 1] + y
    return out.astype(np.float32)


def _pose7(pose3, off=(0.0, 0.0, 0.0)):
    x, y, yaw = pose3
    return np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2), x + off[0], y + off[1], off[2]])


def test_cpp_rccl_row_parallel_and_tile_parallel(tmp_path, orc, synth, api):
    from lightloam_amd import build
    lib_dir = os.path.dirname(build.lib_path())
    exe = str(tmp_path / "rccl_normal_equations")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "native", "rccl_normal_equations.cpp"), "-o", exe,
                           "-L", lib_dir, "-llightloam_hip", "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    cfg = synth.default_cfg(16)
    P = orc.params(16)
    n_hist, n_frames = 5, 4
    feats = [orc.extract(synth.scan(cfg, k), P) for k in range(n_hist + 1)]
    poses = [synth.pose(cfg, k) for k in range(n_hist + 1)]
    corner_map = orc.voxel_grid(np.concatenate([_to_world(f["less_sharp"], p) for f, p in zip(feats[:-1], poses[:-1])]), 0.4)
    surf_map = orc.voxel_grid(np.concatenate([_to_world(f["less_flat"], p) for f, p in zip(feats[:-1], poses[:-1])]), 0.8)
    corner_stack = orc.voxel_grid(feats[-1]["less_sharp"], 0.4)
    surf_stack = orc.voxel_grid(feats[-1]["less_flat"], 0.8)
    guess = _pose7(poses[-1], (0.15, -0.1, 0.03))
    for name, a in (("map_corner", corner_map), ("map_surf", surf_map), ("stack_corner", corner_stack), ("stack_surf", surf_stack)):
        np.ascontiguousarray(a, "<f4").tofile(tmp_path / f"{name}.bin")
    guess.astype("<f8").tofile(tmp_path / "pose.bin")
    for k in range(n_frames):
        np.ascontiguousarray(feats[k]["less_sharp"], "<f4").tofile(tmp_path / f"corner_{k}.bin")
        np.ascontiguousarray(feats[k]["less_flat"], "<f4").tofile(tmp_path / f"surf_{k}.bin")
        _pose7(poses[k], (0.05 * k, -0.03, 0.01)).astype("<f8").tofile(tmp_path / f"odom_{k}.bin")
    env = dict(os.environ)
    out = subprocess.run([exe, str(tmp_path), str(n_frames)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rccl world" in out.stdout

    row = np.fromfile(tmp_path / "out_row.bin", "<f8")
    world = int(row[0])
    assert world >= 1
    ranks = row[1:1 + 7 * world].reshape(world, 7); ref = row[1 + 7 * world:8 + 7 * world]; n_red = row[8 + 7 * world:]
    assert (n_red == 2 * (1 + 4)).all(), n_red                         # per outer iteration: lm_begin + four accepts
    assert np.isfinite(ranks).all()
    for r in range(1, world):
        assert ranks[r].tobytes() == ranks[0].tobytes(), "ranks disagree on the pose"
    if world == 1:
        assert ranks[0].tobytes() == ref.tobytes()                     # the same kernels, the all-reduce of one rank is the identity
    else:
        assert np.abs(ranks[0] - ref).max() <= 1e-7                    # f64 summation order
    gt = _pose7(poses[-1])
    assert np.abs(ranks[0][4:] - gt[4:]).max() < 0.05                  # and it is the right answer

    tile = np.fromfile(tmp_path / "out_tile.bin", "<f8").reshape(n_frames, world + 1, 7)
    for k in range(n_frames):
        for r in range(world):
            assert tile[k, r].tobytes() == tile[k, world].tobytes(), (k, r)   # the split search is exact: bit-identical to the unsplit map


def test_cpp_tile_parallel_ranks_leave_a_failing_frame_together(tmp_path, orc, synth, api):
    """include/lightloam_host.hpp, LaserMapping::process_tile_parallel: no exception between two collectives.  Three ranks as three
    host threads (the all_gather = a barrier + memcpy between them); one rank's ll_cubemap_prepare fails -> all three come back
    with an exception, nobody waits in a gather (tests/native/tile_parallel_exits.cpp has a 60 s watchdog)."""
    from lightloam_amd import build
    lib_dir = os.path.dirname(build.lib_path())
    exe = str(tmp_path / "tile_parallel_exits")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "tile_parallel_exits.cpp"), "-o", exe,
                           "-L", lib_dir, "-llightloam_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    cfg = synth.default_cfg(16)
    P = orc.params(16)
    for k in range(2):
        f = orc.extract(synth.scan(cfg, k), P)
        np.ascontiguousarray(f["less_sharp"], "<f4").tofile(tmp_path / f"corner_{k}.bin")
        np.ascontiguousarray(f["less_flat"], "<f4").tofile(tmp_path / f"surf_{k}.bin")
        _pose7(synth.pose(cfg, k), (0.05 * k, -0.03, 0.01)).astype("<f8").tofile(tmp_path / f"odom_{k}.bin")
    out = subprocess.run([exe, str(tmp_path), "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "left the failing frame together" in out.stdout

