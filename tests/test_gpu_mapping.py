"""SURVEY.md section 8f #2, first stage: laserMapping's scan-to-submap optimisation (laserMapping.cpp:1822-2095) on the
device against the oracle's restatement: same residual blocks (exact), line points / plane parameters, normal equations
and the optimised pose within f64 rounding, and the optimised pose against the synthetic ground truth."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-9


def _to_world(pts, pose3):
    x, y, yaw = pose3
    c, s = np.cos(yaw), np.sin(yaw)
    out = pts.astype(np.float64).copy()
    out[:, 0] = c * pts[:, 0] - s * pts[:, 1] + x
    out[:, 1] = s * pts[:, 0] + c * pts[:, 1] + y
    return out.astype(np.float32)


def _pose7(pose3):
    x, y, yaw = pose3
    return np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2), x, y, 0.0])


@pytest.fixture(scope="module", params=[16, 64])
def scene(request, orc, synth):
    rings = request.param
    n_hist = 5 if rings == 16 else 3
    cfg = synth.default_cfg(rings)
    P = orc.params(rings)
    feats = [orc.extract(synth.scan(cfg, k), P) for k in range(n_hist + 1)]
    poses = [synth.pose(cfg, k) for k in range(n_hist + 1)]
    # the cube map's content: earlier scans' features in the world frame, down-sized like :2151-2165 (0.4 / 0.8 m)
    corner_map = orc.voxel_grid(np.concatenate([_to_world(f["less_sharp"], p) for f, p in zip(feats[:-1], poses[:-1])]), 0.4)
    surf_map = orc.voxel_grid(np.concatenate([_to_world(f["less_flat"], p) for f, p in zip(feats[:-1], poses[:-1])]), 0.8)
    corner_stack = orc.voxel_grid(feats[-1]["less_sharp"], 0.4)              # :1813-1821
    surf_stack = orc.voxel_grid(feats[-1]["less_flat"], 0.8)
    gt = _pose7(poses[-1])
    guess = gt.copy(); guess[4:] += [0.15, -0.1, 0.03]
    dq = np.array([0.002, -0.001, 0.004, 1.0]); dq /= np.linalg.norm(dq)
    ax, ay, az, aw = dq; bx, by, bz, bw = guess[:4]
    guess[:4] = [aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                 aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz]
    return dict(rings=rings, corner_map=corner_map, surf_map=surf_map, corner_stack=corner_stack, surf_stack=surf_stack, gt=gt, guess=guess)


@pytest.fixture(scope="module")
def dev(scene, api):
    ctx = api.Context(api.default_params(scene["rings"], batch=1, max_points=4096))
    m = api.Map(ctx, len(scene["corner_map"]) + 8, len(scene["surf_map"]) + 8, len(scene["corner_stack"]) + 8, len(scene["surf_stack"]) + 8)
    m.set_map(scene["corner_map"], scene["surf_map"])
    m.set_scan(scene["corner_stack"], scene["surf_stack"])
    yield m
    m.close(); ctx.close()


def close(a, b, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, what
    scale = max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    assert (float(np.abs(a - b).max()) if a.size else 0.0) <= REL * scale, what


def test_blocks_match_oracle(scene, dev, orc):
    q, t = scene["guess"][:4], scene["guess"][4:]
    e_src, e_a, e_b, p_src, p_n, p_d = orc.map_associate(q, t, scene["corner_stack"], scene["corner_map"], scene["surf_stack"], scene["surf_map"])
    assert len(e_src) > 20 and len(p_src) > 100, (len(e_src), len(p_src))
    dev.associate(scene["guess"])
    ne, npl = dev.counts()
    assert (ne, npl) == (len(e_src), len(p_src))
    src, a, b = dev.edges()
    assert (src == e_src).all()
    # the eigenvector's sign is a convention: (a, b) may come out swapped, the residual block is the same
    same = np.abs(a - e_a).max(axis=1) <= 1e-9 * np.maximum(1, np.abs(e_a).max(axis=1))
    swapped = np.abs(a - e_b).max(axis=1) <= 1e-9 * np.maximum(1, np.abs(e_b).max(axis=1))
    assert (same | swapped).all()
    assert (np.where(same[:, None], b - e_b, b - e_a).__abs__().max() <= 1e-9 * max(1, np.abs(e_b).max()))
    src, n, d = dev.planes()
    assert (src == p_src).all()
    close(n, p_n, "plane normals"); close(d, p_d, "negative_OA_dot_norm")
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-12)


def test_normal_equations_match_oracle(scene, dev, orc):
    q, t = scene["guess"][:4], scene["guess"][4:]
    blocks = orc.map_associate(q, t, scene["corner_stack"], scene["corner_map"], scene["surf_stack"], scene["surf_map"])
    Ho, go, co = orc.map_normal_equations(q, t, scene["corner_stack"], blocks[0], blocks[1], blocks[2], scene["surf_stack"], blocks[3], blocks[4], blocks[5])
    dev.associate(scene["guess"])
    H, g, cost = dev.normal_equations(scene["guess"])
    close(H, Ho, "H"); close(g, go, "g"); close(cost, co, "cost")
    # and at another pose with the same blocks (what the LM iterations evaluate)
    other = scene["guess"].copy(); other[4:] += [0.01, 0.02, -0.01]
    Ho, go, co = orc.map_normal_equations(other[:4], other[4:], scene["corner_stack"], blocks[0], blocks[1], blocks[2], scene["surf_stack"], blocks[3], blocks[4], blocks[5])
    H, g, cost = dev.normal_equations(other)
    close(H, Ho, "H'"); close(g, go, "g'"); close(cost, co, "cost'")


def test_residual_jacobian_rows_match_the_functors(scene, dev, orc):
    """what ceres::CostFunction::Evaluate would return per block (LidarEdgeFactor, LidarPlaneNormFactor), loss not applied"""
    q, t = scene["guess"][:4], scene["guess"][4:]
    e_src, e_a, e_b, p_src, p_n, p_d = orc.map_associate(q, t, scene["corner_stack"], scene["corner_map"], scene["surf_stack"], scene["surf_map"])
    dev.associate(scene["guess"])
    r, Jq, Jt = dev.residual_jacobian(scene["guess"])
    ne = len(e_src)
    assert len(r) == 3 * ne + len(p_src)
    src, a, b = dev.edges()                                       # the device's own line points (a / b may be swapped vs the oracle)
    for i in range(0, ne, max(1, ne // 50)):
        ro, Jqo, Jto = orc.edge_factor(q, t, scene["corner_stack"][src[i], :3], a[i], b[i])
        close(r[3 * i:3 * i + 3], ro, "edge r"); close(Jq[3 * i:3 * i + 3], Jqo, "edge Jq"); close(Jt[3 * i:3 * i + 3], Jto, "edge Jt")
    for i in range(0, len(p_src), max(1, len(p_src) // 50)):
        ro, Jqo, Jto = orc.plane_norm_factor(q, t, scene["surf_stack"][p_src[i], :3], p_n[i], p_d[i])
        row = 3 * ne + i
        close(r[row:row + 1], ro, "plane r"); close(Jq[row:row + 1], Jqo, "plane Jq"); close(Jt[row:row + 1], Jto, "plane Jt")


def test_optimize_matches_oracle_and_ground_truth(scene, dev, orc):
    q, t, ran = orc.map_optimize(scene["guess"][:4], scene["guess"][4:], scene["corner_stack"], scene["corner_map"], scene["surf_stack"], scene["surf_map"])
    assert ran
    pose, ran_d = dev.optimize(scene["guess"])
    assert ran_d
    assert np.abs(pose[:4] - q).max() < 1e-7 and np.abs(pose[4:] - t).max() < 1e-7, (pose, q, t)
    gt = scene["gt"]
    assert np.abs(pose[4:6] - gt[4:6]).max() < 0.05, (pose, gt)                   # pulled back onto the map
    assert np.abs(pose[4:6] - gt[4:6]).max() < np.abs(scene["guess"][4:6] - gt[4:6]).max()
    assert abs(abs(np.dot(pose[:4], gt[:4])) - 1.0) < 1e-4


def test_small_map_is_left_alone(scene, api):
    """:1822 -- the optimisation only runs with more than 10 corner and 50 surf map points"""
    ctx = api.Context(api.default_params(scene["rings"], batch=1, max_points=4096))
    m = api.Map(ctx, 64, 64, len(scene["corner_stack"]) + 8, len(scene["surf_stack"]) + 8)
    m.set_map(scene["corner_map"][:10], scene["surf_map"][:60])
    m.set_scan(scene["corner_stack"], scene["surf_stack"])
    pose, ran = m.optimize(scene["guess"])
    assert not ran and (pose == scene["guess"]).all()
    with pytest.raises(api.LightLoamError) as e:
        m.set_map(scene["corner_map"][:65], scene["surf_map"][:60])
    assert e.value.code == -4
    m.close(); ctx.close()


# ---- row-parallel over ranks (SURVEY 8e / BASELINE config 4): two processes share the box's one GPU, gloo carries the
# all-reduce (RCCL needs one device per rank; the driver's multi-GPU runs use it through the same code path)
def _row_parallel_worker(rank, world, port, rings, out):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import torch.distributed as dist
        import lightloam_amd  # noqa: F401
        from lightloam_amd import api, parallel, synth
        from oracle import orc

        class Req:
            param = rings
        sc = scene.__wrapped__(Req, orc, synth)
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        ctx = api.Context(api.default_params(rings, batch=1, max_points=4096))
        cs, ss = sc["corner_stack"][rank::world], sc["surf_stack"][rank::world]          # this rank's share of the rows
        m = api.Map(ctx, len(sc["corner_map"]) + 8, len(sc["surf_map"]) + 8, len(cs) + 8, len(ss) + 8)
        m.set_map(sc["corner_map"], sc["surf_map"])
        m.set_scan(cs, ss)
        pose = parallel.map_optimize_row_parallel(m, sc["guess"])
        m.close()
        if rank == 0:
            full = api.Map(ctx, len(sc["corner_map"]) + 8, len(sc["surf_map"]) + 8, len(sc["corner_stack"]) + 8, len(sc["surf_stack"]) + 8)
            full.set_map(sc["corner_map"], sc["surf_map"]); full.set_scan(sc["corner_stack"], sc["surf_stack"])
            ref, ran = full.optimize(sc["guess"])
            full.close()
            assert ran and np.abs(pose - ref).max() < 1e-7, (pose, ref)
        import torch
        got = [torch.zeros(7, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(got, torch.from_numpy(pose))
        assert all((got[0] == g).all() for g in got)                                     # identical LM state everywhere
        ctx.close()
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        out.put((rank, repr(e) + traceback.format_exc()))
    finally:
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass


def test_row_parallel_two_ranks_match_one():
    import multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    out = mpc.Queue()
    procs = [mpc.Process(target=_row_parallel_worker, args=(r, 2, port, 16, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_an_undefined_pose_is_an_error_not_a_result(scene, api):
    """Round-3 advice: a solve that ends with a NaN pose (k_map_compact's lost-predecessor path poisons the pose; a NaN guess does the
    same from outside) used to hand that pose back with LL_OK.  Now ll_map_optimize / ll_map_solve / ll_map_get_pose return
    LL_ERR_STATE, the hand-over words are repaired, and the next solve with a proper guess gives the usual answer."""
    ctx = api.Context(api.default_params(scene["rings"], batch=1, max_points=4096))
    m = api.Map(ctx, len(scene["corner_map"]) + 8, len(scene["surf_map"]) + 8, len(scene["corner_stack"]) + 8, len(scene["surf_stack"]) + 8)
    m.set_map(scene["corner_map"], scene["surf_map"]); m.set_scan(scene["corner_stack"], scene["surf_stack"])
    good, ran = m.optimize(scene["guess"])
    assert ran and np.isfinite(good).all()
    bad = scene["guess"].copy(); bad[5] = np.nan
    with pytest.raises(api.LightLoamError) as e:
        m.optimize(bad)
    assert e.value.code == -7, e.value                                           # LL_ERR_STATE
    again, ran = m.optimize(scene["guess"])
    assert ran and again.tobytes() == good.tobytes()
    m.close(); ctx.close()


@pytest.mark.parametrize("when", ["begin", "accept"])
def test_a_peers_nan_record_reaches_the_healthy_ranks_pose(scene, api, when):
    """Round-5 advice: in the row-parallel solve a failing rank contributes a record of NaNs to the all-reduce (lightloam_rccl.hpp,
    parallel.map_optimize_row_parallel).  A HEALTHY rank that steps its LM state with the NaN sum must end with an error, not with a
    finite un-optimised pose and LL_OK: ll_map_lm_begin / _accept carry an explicit status (ll_lm_step.h, state slot 71), every later
    step leaves a NaN pose, and ll_map_get_pose reports LL_ERR_STATE.  The next frame (set_pose + a clean sequence) works again."""
    ctx = api.Context(api.default_params(scene["rings"], batch=1, max_points=4096))
    m = api.Map(ctx, len(scene["corner_map"]) + 8, len(scene["surf_map"]) + 8, len(scene["corner_stack"]) + 8, len(scene["surf_stack"]) + 8)
    m.set_map(scene["corner_map"], scene["surf_map"]); m.set_scan(scene["corner_stack"], scene["surf_stack"])
    nan_rec = np.full(44, np.nan)

    def sequence(poison):
        m.set_pose(scene["guess"]); m.associate(None)
        rec = m.evaluate()
        m.lm_begin(nan_rec if poison == "begin" else rec)
        for k in range(4):
            m.lm_propose()
            rec = m.evaluate()
            m.lm_accept(nan_rec if (poison == "accept" and k == 1) else rec)
        return m.pose()

    with pytest.raises(api.LightLoamError) as e:
        sequence(when)
    assert e.value.code == -7, e.value                                           # LL_ERR_STATE on the healthy rank too
    good = sequence(None)
    assert np.isfinite(good).all()
    ref, ran = m.optimize(scene["guess"], n_outer=1)
    assert ran and np.abs(good - ref).max() < 1e-7
    m.close(); ctx.close()
