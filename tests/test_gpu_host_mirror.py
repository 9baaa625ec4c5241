"""The C++ host mirror (include/lightloam_host.hpp) driven like the reference nodes would: a g++-built program links
the HIP library, reads two KITTI-format .bin scans, and its outputs are checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_mirror_pipeline(tmp_path, orc, synth, api):
    from lightloam_amd import build
    lib_dir = os.path.dirname(build.lib_path())
    exe = str(tmp_path / "host_pipeline")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "host_pipeline.cpp"), "-o", exe,
                           "-L", lib_dir, "-llightloam_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    cfg = synth.default_cfg(16)
    scans = [synth.scan(cfg, k) for k in range(2)]
    for k, s in enumerate(scans):
        s.astype("<f4").tofile(tmp_path / f"scan{k}.bin")
    out = subprocess.run([exe, str(tmp_path / "scan0.bin"), str(tmp_path / "scan1.bin"), str(tmp_path / "o"), "16"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    P = orc.params(16)
    e0, e1 = orc.extract(scans[0], P), orc.extract(scans[1], P)
    rd = lambda name, dt=np.float32: np.fromfile(tmp_path / f"o_{name}.bin", dtype=dt)
    for name, key in (("cloud1", "cloud"), ("sharp1", "sharp"), ("lsharp1", "less_sharp"), ("flat1", "flat"), ("lflat1", "less_flat")):
        assert rd(name).tobytes() == e1[key].tobytes(), name
    q = np.array([0, 0, 0, 1.0]); t = np.array([0.9, 0, 0])
    es, ea, eb = orc.associate_corner(q, t, e1["sharp"], e0["less_sharp"])
    ps, pa, pb, pc = orc.associate_plane(q, t, e1["flat"], e0["less_flat"])
    cnt, sidx, sw = orc.vote(e1["flat"][ps], e0["less_flat"][pa])
    order = np.sort(sidx); wmap = np.ones(len(ps), np.float32); wmap[sidx] = sw
    H, g, _ = orc.normal_equations(q, t, e1["sharp"], es, e0["less_sharp"], ea, eb, e1["flat"], ps[order], e0["less_flat"],
                                   pa[order], pb[order], pc[order], wmap[order], 0.1)
    rc, d = orc.gn_solve(H, g)
    qo, to = orc.pose_update(q, t, d)
    pose = rd("pose", np.float64)
    assert np.allclose(pose[:4], qo, atol=1e-9) and np.allclose(pose[4:], to, atol=1e-9)
    # the free function: same selected set and weights as the oracle, in the reference's low-count-first order per region
    sel = rd("selected").reshape(-1, 2)
    got_idx = sel[:, 0].astype(int)
    assert sorted(got_idx.tolist()) == sorted(sidx.tolist())
    assert all(wmap[i] == w for i, w in zip(got_idx, sel[:, 1]))
    n = len(ps); chunk = n // 10
    region = np.minimum(got_idx // chunk, 9) if chunk else np.full(len(got_idx), 9)
    assert (np.diff(region) >= 0).all()
    for r in range(10):
        c = cnt[got_idx[region == r]]
        assert (np.diff(c) >= 0).all()


def test_kitti_directory_tool_writes_the_reference_trajectory_format(tmp_path, synth, api):
    """tools/ll_odometry_kitti.cpp: *.bin directory in, laserMapping.cpp:2306-2325 text format out."""
    from lightloam_amd import build
    lib_dir = os.path.dirname(build.lib_path())
    exe = str(tmp_path / "ll_odometry_kitti")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "ll_odometry_kitti.cpp"), "-o", exe,
                           "-L", lib_dir, "-llightloam_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    cfg = synth.default_cfg(16)
    n = 10
    scans = [synth.scan(cfg, k) for k in range(n)]
    d = tmp_path / "velodyne"; d.mkdir()
    for k, s in enumerate(scans):
        s.astype("<f4").tofile(d / f"{k:06d}.bin")
    res = tmp_path / "traj.txt"
    out = subprocess.run([exe, str(d), str(res), "16", "0.9"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    T = np.loadtxt(res)
    assert T.shape == (n, 12)
    assert np.allclose(T[0], [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0])           # H_init^-1 * H_init
    line = open(res).readline().split()
    assert all("e" in v and len(v.split("e")[0].split(".")[1]) == 6 for v in line)   # scientific, precision 6
    # same numbers as the Python-side integration of the device's relative poses
    ctx = api.Context(api.default_params(16, batch=n, max_points=max(map(len, scans))))
    for k, s in enumerate(scans):
        ctx.upload_scan(k, s)
    ctx.extract(0, n); ctx.set_target_from_slot(0)
    rel = ctx.odometry_frames(1, n - 1, pose0=[0, 0, 0, 1, 0.9, 0, 0])
    ctx.close()
    qw = np.array([0, 0, 0, 1.0]); tw = np.zeros(3)
    for k, p in enumerate(rel):
        u, w = qw[:3], qw[3]; uv = 2 * np.cross(u, p[4:]); tw = tw + p[4:] + w * uv + np.cross(u, uv)
        ax, ay, az, aw = qw; bx, by, bz, bw = p[:4]
        qw = np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                       aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])
        assert np.allclose(T[k + 1, [3, 7, 11]], tw, atol=2e-6 * max(1, np.abs(tw).max()))
    gt = synth.pose(cfg, n - 1)
    assert abs(T[-1, 3] - gt[0]) < 0.2 and abs(T[-1, 7] - gt[1]) < 0.2


def test_kitti_tool_with_the_mapping_node(tmp_path, synth):
    """the three-node chain (scanRegistration -> laserOdometry -> laserMapping) through the C++ host mirror: the written
    trajectory is the mapped one (laserMapping.cpp:2284-2325) and stays on the synthetic ground truth."""
    from lightloam_amd import build
    lib_dir = os.path.dirname(build.lib_path())
    exe = str(tmp_path / "ll_odometry_kitti")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-pthread", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "ll_odometry_kitti.cpp"), "-o", exe,
                           "-L", lib_dir, "-llightloam_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    cfg = synth.default_cfg(16)
    n = 12
    d = tmp_path / "velodyne"; d.mkdir()
    for k in range(n):
        synth.scan(cfg, k).astype("<f4").tofile(d / f"{k:06d}.bin")
    odo, mapped, tiled = tmp_path / "odo.txt", tmp_path / "map.txt", tmp_path / "map3.txt"
    for res, flag in ((odo, "0"), (mapped, "1"), (tiled, "3")):
        out = subprocess.run([exe, str(d), str(res), "16", "0.9", flag], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
    # the map split over three ranks (LaserMapping::process_tile_parallel) writes the same file
    assert tiled.read_bytes() == mapped.read_bytes()
    To, Tm = np.loadtxt(odo), np.loadtxt(mapped)
    assert To.shape == Tm.shape == (n, 12)
    assert np.allclose(Tm[0], [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0])
    gt = np.array([synth.pose(cfg, k) for k in range(n)])
    c, s_ = np.cos(gt[0, 2]), np.sin(gt[0, 2])
    gt_xy = (gt[:, :2] - gt[0, :2]) @ np.array([[c, -s_], [s_, c]])
    ate = lambda T: float(np.sqrt(np.mean(np.sum((T[:, [3, 7]] - gt_xy) ** 2, axis=1))))
    travelled = float(np.linalg.norm(np.diff(gt_xy, axis=0), axis=1).sum())
    assert ate(To) < 0.02 * travelled and ate(Tm) < 0.02 * travelled, (ate(To), ate(Tm), travelled)
    assert np.abs(Tm[:, [3, 7]] - To[:, [3, 7]]).max() < 0.25                  # a refinement of the odometry, not a different path
