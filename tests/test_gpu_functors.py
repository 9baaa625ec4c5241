"""The lidarFactor.hpp functors on caller-supplied residual blocks (ll_factor_blocks_*, ll_functors.hip) and the
drop-in header include/lightloam_lidarFactor.hpp that keeps the reference's Create(...) call sites
(laserOdometry.cpp:615, :783; laserMapping.cpp:1918, :2033), against the oracle's restatement of the functors
(validated there by forward-mode duals and central differences).  Tolerance: 1e-12 relative on f64 residuals and
Jacobians (the device uses the same closed forms; differences are f64 rounding of a few operations)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL = 1e-12


def _blocks(seed, n_e, n_p, n_n):
    rng = np.random.default_rng(seed)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)              # the nodes build Vector3d from float points
    curr = lambda n: f32(rng.uniform(-40, 40, (n, 3)))
    edge = np.hstack([curr(n_e), f32(rng.uniform(-40, 40, (n_e, 3))), f32(rng.uniform(-40, 40, (n_e, 3)))])
    j = f32(rng.uniform(-40, 40, (n_p, 3)))
    plane = np.hstack([curr(n_p), j, f32(j + rng.uniform(-2, 2, (n_p, 3))), f32(j + rng.uniform(-2, 2, (n_p, 3))),
                       rng.choice([1.0, 5.0], (n_p, 1))])
    nrm = rng.normal(size=(n_n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    pnorm = np.hstack([curr(n_n), nrm, rng.uniform(-3, 3, (n_n, 1))])   # mapping's normals are f64 (QR solve)
    return edge, plane, pnorm


def _poses():
    out = [np.array([0, 0, 0, 1.0, 0, 0, 0])]
    for ax, ang, t in (((0, 0, 1), 0.03, (0.9, -0.1, 0.02)), ((0.3, -0.5, 0.8), 0.6, (-3.0, 2.0, 0.5))):
        ax = np.array(ax, float); ax /= np.linalg.norm(ax)
        out.append(np.concatenate([np.sin(ang / 2) * ax, [np.cos(ang / 2)], t]))
    return out


def _oracle_rows(orc, edge, plane, pnorm, pose):
    q, t = pose[:4], pose[4:]
    r, Jq, Jt = [], [], []
    for e in edge:
        a = orc.edge_factor(q, t, e[0:3], e[3:6], e[6:9]); r.append(a[0]); Jq.append(a[1]); Jt.append(a[2])
    for p in plane:
        a = orc.plane_factor_modify(q, t, p[0:3], p[3:6], p[6:9], p[9:12], 1.0, p[12]); r.append(a[0]); Jq.append(a[1]); Jt.append(a[2])
    for p in pnorm:
        a = orc.plane_norm_factor(q, t, p[0:3], p[3:6], p[6]); r.append(a[0]); Jq.append(a[1]); Jt.append(a[2])
    return np.concatenate(r), np.vstack(Jq), np.vstack(Jt)


def _close(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, what
    scale = max(1.0, np.abs(b).max())
    assert np.abs(a - b).max() <= REL * scale * 50, (what, np.abs(a - b).max(), scale)


@pytest.mark.parametrize("n_e,n_p,n_n", [(37, 91, 53), (0, 300, 0), (5, 0, 0), (0, 0, 1)])
def test_blocks_match_the_oracle_functors(api, orc, n_e, n_p, n_n):
    ctx = api.Context(api.default_params(16, batch=1, max_points=4096))
    edge, plane, pnorm = _blocks(n_e + 7 * n_p + n_n, n_e, n_p, n_n)
    ctx.factor_blocks_set(edge, plane, pnorm)
    for pose in _poses():
        r, Jq, Jt = ctx.factor_blocks_evaluate(pose[:4], pose[4:])
        ro, Jqo, Jto = _oracle_rows(orc, edge, plane, pnorm, pose)
        _close(r, ro, "residuals"); _close(Jq, Jqo, "d r / d q"); _close(Jt, Jto, "d r / d t")
    # a second problem on the same context replaces the first (smaller, then larger: the buffers grow)
    edge2, plane2, pnorm2 = _blocks(5, 3, 2, 1)
    ctx.factor_blocks_set(edge2, plane2, pnorm2)
    r, _, _ = ctx.factor_blocks_evaluate(*np.split(_poses()[1], [4]))
    assert len(r) == 3 * 3 + 2 + 1
    _close(r, _oracle_rows(orc, edge2, plane2, pnorm2, _poses()[1])[0], "second problem")
    big = _blocks(9, 3000, 6000, 2000)
    ctx.factor_blocks_set(*big)
    r, Jq, Jt = ctx.factor_blocks_evaluate(*np.split(_poses()[2], [4]))
    idx = np.r_[0:9, 3 * 3000 + 5990:3 * 3000 + 6000, 3 * 3000 + 6000 + 1990:3 * 3000 + 8000]
    ro, Jqo, Jto = _oracle_rows(orc, big[0][:3], big[1][5990:], big[2][1990:], _poses()[2])
    _close(r[idx], ro, "large problem"); _close(Jq[idx], Jqo, "large problem Jq")
    ctx.close()


def test_block_entry_points_validate_their_arguments(api):
    import ctypes as C
    ctx = api.Context(api.default_params(16, batch=1, max_points=4096))
    lib = ctx.lib
    e = np.zeros((2, 9))
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.ll_factor_blocks_set(ctx.h, 2, None, 0, None, 0, None) == -2           # blocks announced, no data
    assert lib.ll_factor_blocks_set(ctx.h, -1, P(e), 0, None, 0, None) == -2
    assert lib.ll_factor_blocks_set(ctx.h, 2, P(e), 0, None, 0, None) == 0
    q = np.array([0, 0, 0, 1.0]); t = np.zeros(3); r = np.zeros(6)
    assert lib.ll_factor_blocks_evaluate(ctx.h, P(q), P(t), P(r), None, None, 5) == -4  # 6 rows do not fit in 5
    assert lib.ll_factor_blocks_evaluate(ctx.h, None, P(t), P(r), None, None, 6) == -2
    assert lib.ll_factor_blocks_evaluate(ctx.h, P(q), P(t), P(r), None, None, 6) == 0   # residuals only
    ctx.close()


def test_blocks_with_per_block_s_match_the_oracle_functors(api, orc):
    """DISTORTION 1: the functors' s_ per block (ll_factor_blocks_set_s) through Identity.slerp(s, q), s * t (lidarFactor.hpp:25-27)"""
    ctx = api.Context(api.default_params(16, batch=1, max_points=4096))
    edge, plane, pnorm = _blocks(31, 40, 90, 11)
    rng = np.random.default_rng(7)
    es, ps = rng.uniform(0.0, 1.0, len(edge)), rng.uniform(0.0, 1.0, len(plane))
    es[:3] = [0.0, 1.0, 0.5]; ps[:2] = [1.0, 0.0]                                   # the ends of the sweep too
    ctx.factor_blocks_set(edge, plane, pnorm)
    ctx.factor_blocks_set_s(es, ps)
    for pose in _poses():
        q, t = pose[:4], pose[4:]
        r, Jq, Jt = ctx.factor_blocks_evaluate(q, t)
        ro, Jqo, Jto = [], [], []
        for e, s_ in zip(edge, es):
            a = orc.edge_factor(q, t, e[0:3], e[3:6], e[6:9], s_); ro.append(a[0]); Jqo.append(a[1]); Jto.append(a[2])
        for p, s_ in zip(plane, ps):
            a = orc.plane_factor_modify(q, t, p[0:3], p[3:6], p[6:9], p[9:12], s_, p[12]); ro.append(a[0]); Jqo.append(a[1]); Jto.append(a[2])
        for p in pnorm:
            a = orc.plane_norm_factor(q, t, p[0:3], p[3:6], p[6]); ro.append(a[0]); Jqo.append(a[1]); Jto.append(a[2])
        _close(r, np.concatenate(ro), "residuals"); _close(Jq, np.vstack(Jqo), "d r / d q"); _close(Jt, np.vstack(Jto), "d r / d t")
    # a new set of blocks starts at s = 1 again
    ctx.factor_blocks_set(edge, plane, pnorm)
    r, Jq, Jt = ctx.factor_blocks_evaluate(*np.split(_poses()[0], [4]))
    ro, Jqo, Jto = _oracle_rows(orc, edge, plane, pnorm, _poses()[0])
    _close(r, ro, "s reset"); _close(Jq, Jqo, "s reset Jq")
    ctx.close()


@pytest.mark.parametrize("with_s", [False, True])
def test_lidar_factor_header_keeps_the_reference_call_sites(tmp_path, api, orc, with_s):
    from lightloam_amd import build
    lib_dir = os.path.dirname(build.lib_path())
    exe = str(tmp_path / "lidar_factor_adapter")
    subprocess.check_call(["g++", "-O2", "-std=c++14", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "lidar_factor_adapter.cpp"), "-o", exe,
                           "-L", lib_dir, "-llightloam_hip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    n = (23, 40, 31)
    edge, plane, pnorm = _blocks(77, *n)
    poses = _poses()[1:]
    with open(tmp_path / "blocks.bin", "wb") as f:
        f.write(np.array(n, np.int32).tobytes()); f.write(edge.tobytes()); f.write(plane.tobytes()); f.write(pnorm.tobytes())
        f.write(np.concatenate(poses).tobytes())
    out = subprocess.run([exe, str(tmp_path / "blocks.bin"), str(tmp_path / "out.bin")] + (["s"] if with_s else []), capture_output=True, text=True, timeout=300)
    bs = (lambda i: 0.25 + 0.5 * ((i % 7) / 7.0)) if with_s else (lambda i: 1.0)      # the s_ the program passes to Create()
    assert out.returncode == 0, out.stdout + out.stderr
    assert "2 device evaluations" in out.stdout                                        # 94 cost functions x 3 passes, two launches
    got = np.fromfile(tmp_path / "out.bin", np.float64)
    order = []                                                                         # the program's creation order
    for i in range(max(n)):
        order += [(k, i) for k in range(3) if i < n[k]]
    at = 0
    for pass_, pose in ((0, poses[0]), (1, poses[0]), (2, poses[1])):
        q, t = pose[:4], pose[4:]
        for k, i in order:
            if k == 0:
                ro, Jqo, Jto = orc.edge_factor(q, t, edge[i, 0:3], edge[i, 3:6], edge[i, 6:9], bs(i))
            elif k == 1:
                p = plane[i]; ro, Jqo, Jto = orc.plane_factor_modify(q, t, p[0:3], p[3:6], p[6:9], p[9:12], bs(i), p[12])
            else:
                p = pnorm[i]; ro, Jqo, Jto = orc.plane_norm_factor(q, t, p[0:3], p[3:6], p[6])
            rows = len(ro)
            _close(got[at:at + rows], ro, (pass_, k, i)); at += rows
            if pass_ != 1:
                _close(got[at:at + 4 * rows].reshape(rows, 4), Jqo, (pass_, k, i, "Jq")); at += 4 * rows
                _close(got[at:at + 3 * rows].reshape(rows, 3), Jto, (pass_, k, i, "Jt")); at += 3 * rows
    assert at == len(got)
