"""The ROS1 node wrappers (ros/lightloam_*_node.cpp) compiled as they are against DECLARED TEST DOUBLES of the
roscpp / sensor_msgs / nav_msgs / tf classes they use (tests/native/ros_double -- this image has no ROS) and driven like roscpp would:
parameters -> main() -> the subscription's callback with a PointCloud2 -> what was published on which topic."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from lightloam_amd import build
    lib_dir = os.path.dirname(build.lib_path())
    exe = str(tmp_path / "ros_node_double")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "tests", "native", "ros_double"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "ros_node_double.cpp"),
                           "-o", exe, "-L", lib_dir, "-llightloam_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cloud2_layout_and_topic_surface_without_a_device(tmp_path, api):
    """pcl::toROSMsg's PointXYZI wire layout (x@0 y@4 z@8 intensity@16, 32-byte points), the field-name based parser on a
    22-byte velodyne-style message, and main()'s early exit on an unsupported scan_line (scanRegistration.cpp:447-451)."""
    out = subprocess.run([_build(tmp_path), "layout"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "fields x@0:7x1 y@4:7x1 z@8:7x1 intensity@16:7x1"
    assert lines[1] == "step 32 row 96 h 1 w 3 dense 1 be 0 bytes 96"
    assert lines[2:] == ["roundtrip 1", "odd_layout 1", "missing_field_rejected 1", "organised 1", "short_data_rejected 1",
                         "bigendian_rejected 1", "empty 1", "bad_scan_line_exit 0 advertised 0"]


@pytest.mark.parametrize("driver", ["ros_odometry_double", "ros_mapping_double"])
def test_odometry_and_mapping_nodes_compile_against_the_doubles(tmp_path, api, driver):
    """no GPU: ros/lightloam_laser_odometry_node.cpp and ros/lightloam_laser_mapping_node.cpp as they are, against the
    declared doubles, linked with the C-ABI library"""
    from lightloam_amd import build
    lib_dir = os.path.dirname(build.lib_path())
    subprocess.check_call(["g++", "-O0", "-std=c++14", "-pthread", "-I", os.path.join(ROOT, "tests", "native", "ros_double"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", driver + ".cpp"),
                           "-o", str(tmp_path / driver), "-L", lib_dir, "-llightloam_hip", "-Wl,-rpath," + lib_dir,
                           "-Wl,-rpath,/opt/rocm/lib"])


@pytest.mark.gpu
def test_node_publishes_the_five_clouds_of_the_reference(tmp_path, orc, synth, api):
    """One 64-ring scan through main() + laserCloudHandler: the reference's subscription and six advertisements with
    their queue sizes (:453-465), five publications stamped and framed like the input (:382-410), and the published
    points bit-identical to the oracle's clouds."""
    cfg = synth.default_cfg(64)
    scan = synth.scan(cfg, 2)
    path = tmp_path / "scan.bin"
    scan.astype("<f4").tofile(path)
    out = subprocess.run([_build(tmp_path), "run", str(path), "64", "5.0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "subscribed /rslidar_points:100"
    assert lines[1] == ("advertised /laser_cloud_flat:100 /laser_cloud_less_flat:100 /laser_cloud_less_sharp:100 "
                        "/laser_cloud_sharp:100 /laser_remove_points:100 /velodyne_cloud_2:100")
    ref = orc.extract(scan, orc.params(64))
    topics = {"/velodyne_cloud_2": "cloud", "/laser_cloud_sharp": "sharp", "/laser_cloud_less_sharp": "less_sharp",
              "/laser_cloud_flat": "flat", "/laser_cloud_less_flat": "less_flat"}
    seen = {}
    for ln in lines[2:]:
        topic, _, n, _, npub, _, stamp, _, frame = ln.split()
        seen[topic] = int(n)
        assert npub == "1" and stamp == "1234.5678" and frame == "rslidar"
    assert set(seen) == set(topics)                             # /laser_remove_points is advertised, never published
    for topic, key in topics.items():
        got = np.fromfile(str(path) + "." + topic.replace("/", "_") + ".f4", dtype=np.float32)
        assert got.tobytes() == ref[key].tobytes(), topic


@pytest.mark.gpu
def test_odometry_node_frame_loop_and_topics(tmp_path, synth, api):
    """ros/lightloam_laser_odometry_node.cpp on an 8-frame 16-ring drive: five subscriptions and five advertisements with
    the reference's names and queue sizes (laserOdometry.cpp:354-372), one odometry + path message per frame (the first
    frame only initialises: identity), the clouds every mapping_skip_frame-th frame in "/camera", and the published
    world poses = the device frame loop's relative poses composed as :830-831 does."""
    from lightloam_amd import build
    from test_gpu_odometry import integrate, qmul
    rings, nframes = 16, 8
    cfg = synth.default_cfg(rings)
    scans = [synth.scan(cfg, k) for k in range(nframes)]
    reg = api.Context(api.default_params(rings, batch=nframes, max_points=max(map(len, scans))))
    for k, s in enumerate(scans):
        reg.upload_scan(k, s)
    reg.extract(0, nframes)
    reg.set_target_from_slot(0)
    rel = reg.odometry_frames(1, nframes - 1, pose0=None, n_outer=3, first_frame_index=1)
    for k in range(nframes):
        f = reg.features(k)
        for name in ("sharp", "less_sharp", "flat", "less_flat"):
            np.ascontiguousarray(f[name], "<f4").tofile(tmp_path / f"{k}.{name}.f4")
        np.ascontiguousarray(reg.cloud(k)[0], "<f4").tofile(tmp_path / f"{k}.cloud.f4")
    reg.close()
    lib_dir = os.path.dirname(build.lib_path())
    exe = str(tmp_path / "ros_odometry_double")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "tests", "native", "ros_double"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "ros_odometry_double.cpp"),
                           "-o", exe, "-L", lib_dir, "-llightloam_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(tmp_path), str(nframes), str(rings)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "rc 0"
    assert lines[1] == ("subscribed /laser_cloud_flat:100 /laser_cloud_less_flat:100 /laser_cloud_less_sharp:100 "
                        "/laser_cloud_sharp:100 /velodyne_cloud_2:100")
    assert lines[2] == ("advertised /laser_cloud_corner_last:100 /laser_cloud_surf_last:100 /laser_odom_path:100 "
                        "/laser_odom_to_init:100 /velodyne_cloud_3:100")
    assert lines[3] == (f"published /laser_cloud_corner_last:{nframes // 2} /laser_cloud_surf_last:{nframes // 2} "
                        f"/laser_odom_path:{nframes} /laser_odom_to_init:{nframes} /velodyne_cloud_3:{nframes // 2}")
    assert lines[4] == f"path_poses {nframes} path_frame rslidar cloud_frame /camera cloud_stamp {100 + nframes - 2}"
    rows = [ln.split() for ln in open(tmp_path / "odom.txt").read().strip().splitlines()]
    assert len(rows) == nframes and all(r[8] == "rslidar" and r[9] == "/laser_odom" for r in rows)
    assert [int(r[7]) for r in rows] == [100 + k for k in range(nframes)]
    got = np.array([[float(v) for v in r[:7]] for r in rows])
    assert np.array_equal(got[0], [0, 0, 0, 1, 0, 0, 0])                 # the initialisation frame
    tw = integrate(rel)                                                   # t_w after every frame (:830)
    qw = np.array([0, 0, 0, 1.0]); qs = [qw]
    for p in rel:
        qw = qmul(qw, p[:4]); qs.append(qw)                               # q_w (:831)
    assert np.allclose(got[:, 4:], tw, rtol=0, atol=1e-12) and np.allclose(got[:, :4], np.array(qs), rtol=0, atol=1e-12)


def _qmul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def _qrot(q, v):
    u, w = np.asarray(q[:3]), q[3]
    uv = 2 * np.cross(u, v)
    return v + w * uv + np.cross(u, uv)


@pytest.mark.gpu
def test_mapping_node_frame_loop_and_topics(tmp_path, synth, api):
    """ros/lightloam_laser_mapping_node.cpp on a 7-frame 16-ring drive: four subscriptions and six advertisements
    (laserMapping.cpp:2369-2387), per frame one /aft_mapped_to_init + path + registered cloud + tf + RESULT_PATH line, the
    surround cloud on frame 0 and 5, the map cloud on frame 0, the high-frequency republish with the reference's field
    permutation; mapped poses = transformAssociateToMap -> ll_cubemap_process -> transformUpdate driven from Python."""
    from lightloam_amd import build
    from test_gpu_odometry import integrate
    rings, nframes = 16, 7
    cfg = synth.default_cfg(rings)
    scans = [synth.scan(cfg, k) for k in range(nframes)]
    reg = api.Context(api.default_params(rings, batch=nframes, max_points=max(map(len, scans))))
    for k, s in enumerate(scans):
        reg.upload_scan(k, s)
    reg.extract(0, nframes)
    reg.set_target_from_slot(0)
    rel = reg.odometry_frames(1, nframes - 1, pose0=np.array([0, 0, 0, 1.0, 0.9, 0, 0]), n_outer=3, first_frame_index=1)
    tw = integrate(rel)
    qw = [np.array([0, 0, 0, 1.0])]
    for p in rel:
        qw.append(_qmul(qw[-1], p[:4]))
    feats = []
    for k in range(nframes):
        f = reg.features(k); cloud = reg.cloud(k)[0]
        feats.append((f["less_sharp"].copy(), f["less_flat"].copy(), cloud.copy()))
        for name, arr in (("less_sharp", f["less_sharp"]), ("less_flat", f["less_flat"]), ("cloud", cloud)):
            np.ascontiguousarray(arr, "<f4").tofile(tmp_path / f"{k}.{name}.f4")
        with open(tmp_path / f"{k}.odom.txt", "w") as fo:
            fo.write(" ".join(repr(float(v)) for v in list(qw[k]) + list(tw[k])))
    # ---- the same frames through the Python binding
    cm = api.CubeMap(reg, 20000, 200000, pool_points=1 << 22)
    q_wmap, t_wmap = np.array([0, 0, 0, 1.0]), np.zeros(3)
    expect, expect_high = [], []
    for k in range(nframes):
        qh = _qmul(q_wmap, qw[k]); th = _qrot(q_wmap, tw[k]) + t_wmap                      # laserOdometryHandler (:168-247)
        expect_high.append(th)
        guess = np.concatenate([qh, th])                                                    # transformAssociateToMap (:113-117)
        pose, _ = cm.process(guess, feats[k][0], feats[k][1])
        expect.append(pose)
        qinv = np.array([-qw[k][0], -qw[k][1], -qw[k][2], qw[k][3]]) / np.dot(qw[k], qw[k])
        q_wmap = _qmul(pose[:4], qinv); t_wmap = pose[4:] - _qrot(q_wmap, tw[k])            # transformUpdate (:119-123)
    cen, _ = cm.info()
    cm.close(); reg.close()
    lib_dir = os.path.dirname(build.lib_path())
    exe = str(tmp_path / "ros_mapping_double")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-pthread", "-I", os.path.join(ROOT, "tests", "native", "ros_double"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "ros_mapping_double.cpp"),
                           "-o", exe, "-L", lib_dir, "-llightloam_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(tmp_path), str(nframes)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == f"rc 0 frames {nframes}"
    assert lines[1] == "subscribed /laser_cloud_corner_last:100 /laser_cloud_surf_last:100 /laser_odom_to_init:100 /velodyne_cloud_3:100"
    assert lines[2] == ("advertised /aft_mapped_path:100 /aft_mapped_to_init:100 /aft_mapped_to_init_high_frec:100 "
                        "/laser_cloud_map:100 /laser_cloud_surround:100 /velodyne_cloud_registered:100")
    assert lines[3] == (f"published /aft_mapped_path:{nframes} /aft_mapped_to_init:{nframes} /aft_mapped_to_init_high_frec:{nframes} "
                        f"/laser_cloud_map:1 /laser_cloud_surround:2 /velodyne_cloud_registered:{nframes}")
    words = lines[4].split()
    assert words[1] == str(nframes) and int(words[3]) > 1000 and int(words[5]) > 0 and int(words[7]) == len(feats[-1][2]) and words[9] == "rslidar"
    assert lines[5] == f"tf {nframes} rslidar /aft_mapped"
    rows = [ln.split() for ln in open(tmp_path / "mapped.txt").read().strip().splitlines()]
    got = np.array([[float(v) for v in r[:7]] for r in rows])
    assert len(rows) == nframes and all(r[8] == "rslidar" and r[9] == "/aft_mapped" for r in rows)
    assert np.allclose(got, np.array(expect), rtol=0, atol=1e-9)
    tfrow = np.array([float(v) for v in open(tmp_path / "tf.txt").read().split()])
    assert np.array_equal(tfrow, got[-1])
    # RESULT_PATH: one KITTI line per frame, the first one the identity (H_init^-1 H)
    res = [[float(v) for v in ln.split()] for ln in open(tmp_path / "result.txt").read().strip().splitlines()]
    assert len(res) == nframes and np.allclose(res[0], [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], atol=1e-6)
    # the full-resolution scan in map coordinates: q_w_curr * p + t_w_curr in f64, stored f32 (:125-133)
    regd = np.fromfile(tmp_path / "registered.f4", dtype=np.float32).reshape(-1, 4)
    src = feats[-1][2].astype(np.float64)
    want = np.array([_qrot(expect[-1][:4], p[:3]) + expect[-1][4:] for p in src[:500]])
    assert np.allclose(regd[:500, :3], want.astype(np.float32), atol=1e-5) and np.array_equal(regd[:, 3], feats[-1][2][:, 3])
    # high-frequency republish: position = the odometry moved by the last correction; orientation permuted as :237-240
    hrows = [[float(v) for v in ln.split()[:7]] for ln in open(tmp_path / "high.txt").read().strip().splitlines()]
    assert np.allclose(np.array(hrows)[:, 4:], np.array(expect_high), atol=1e-9)
    x, y, z, w = hrows[0][:4]                                        # identity pose: roll = yaw = pi/2 -> q_after = (.5, .5, .5, .5)
    assert np.allclose([x, y, z, w], [0.5, -0.5, 0.5, -0.5], atol=1e-12)
