"""DISTORTION 1 -- the reference's other compile-time path (laserOdometry.cpp:23 is 0 in its build): every point's interpolation
ratio s = (intensity - int(intensity)) / SCAN_PERIOD in TransformToStart (:81-88) and in LidarEdgeFactor / LidarPlaneFactor_modify
(:570-571, :740-741; lidarFactor.hpp:25-27: Identity.slerp(s, q), s * t).  ll_params.distortion = 1 against the oracle with
orc.set_distortion(1): association indices exact, residual / Jacobian rows, normal equations, GN step and the LM solve <= 1e-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL_TOL = 1e-9


def close(a, b, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    err = float(np.abs(a - b).max()) if a.size else 0.0
    assert err <= REL_TOL * scale, f"{what}: max abs err {err:g} vs scale {scale:g}"


@pytest.fixture(scope="module")
def dist(api, orc, synth):
    cfg = synth.default_cfg(16)
    scans = [synth.scan(cfg, k) for k in range(3)]
    P = orc.params(16)
    ctx = api.Context(api.default_params(16, batch=3, max_points=max(map(len, scans)) + 7, distortion=1))
    for k, s in enumerate(scans):
        ctx.upload_scan(k, s)
    ctx.extract(0, 3)
    ref = [orc.extract(s, P) for s in scans]
    # a rotation well away from identity, so that slerp(s, q) differs from q for the points in mid-sweep
    q = np.array([0.02, -0.035, 0.06, 1.0]); q /= np.linalg.norm(q)
    t = np.array([0.8, 0.05, -0.03])
    pose = np.concatenate([q, t])
    orc.set_distortion(1)
    ctx.set_target_from_slot(0)
    ctx.associate(1, 2, pose); ctx.vote(1, 2, True); ctx.normal_equations(1, 2); ctx.synchronize()
    out = []
    for k in (1, 2):
        cur, prev = ref[k], ref[k - 1]
        es, ea, eb = orc.associate_corner(q, t, cur["sharp"], prev["less_sharp"])
        ps, pa, pb, pc = orc.associate_plane(q, t, cur["flat"], prev["less_flat"])
        cnt, sidx, sw = orc.vote(cur["flat"][ps], prev["less_flat"][pa])
        out.append(dict(es=es, ea=ea, eb=eb, ps=ps, pa=pa, pb=pb, pc=pc, cnt=cnt, sidx=sidx, sw=sw, cur=cur, prev=prev))
    yield dict(ctx=ctx, q=q, t=t, pose=pose, ref=out, orc=orc)
    orc.set_distortion(0)
    ctx.close()


def test_interpolation_ratios_spread_over_the_sweep(dist):
    """the synthetic scans carry relTime in the intensity's fraction: s must really differ from 1 for this file to test anything"""
    orc = dist["orc"]
    s = np.array([orc.point_s(p) for p in dist["ref"][0]["cur"]["sharp"][::7]])
    assert s.min() < 0.2 and s.max() > 0.8 and (np.abs(s - 1.0) > 1e-3).mean() > 0.9


def test_association_indices_exact_with_distortion(dist):
    for i, k in enumerate((1, 2)):
        r = dist["ref"][i]
        es, ea, eb = dist["ctx"].edge_corr(k)
        ps, pa, pb, pc = dist["ctx"].plane_corr(k)
        assert len(r["es"]) > 10 and len(r["ps"]) > 10
        for got, want, nm in ((es, r["es"], "e_src"), (ea, r["ea"], "e_a"), (eb, r["eb"], "e_b"),
                              (ps, r["ps"], "p_src"), (pa, r["pa"], "p_a"), (pb, r["pb"], "p_b"), (pc, r["pc"], "p_c")):
            assert len(got) == len(want) and (got == want).all(), f"scan {k} {nm}"


def _oracle_neq(dist, r):
    orc = dist["orc"]
    order = np.sort(r["sidx"])
    wmap = np.ones(len(r["ps"]), np.float32); wmap[r["sidx"]] = r["sw"]
    return orc.normal_equations(dist["q"], dist["t"], r["cur"]["sharp"], r["es"], r["prev"]["less_sharp"], r["ea"], r["eb"],
                                r["cur"]["flat"], r["ps"][order], r["prev"]["less_flat"], r["pa"][order], r["pb"][order],
                                r["pc"][order], wmap[order], 0.1)


def test_rows_with_distortion(dist):
    """r, d r / d q, d r / d t of every block against the oracle's Jets THROUGH the slerp (lidarFactor.hpp:25-27)"""
    orc = dist["orc"]
    k, r = 1, dist["ref"][0]
    rr, Jq, Jt = dist["ctx"].residual_jacobian(k, dist["pose"])
    ne = len(r["es"])
    assert len(rr) == 3 * ne + len(r["sidx"])
    for i in range(0, ne, max(1, ne // 60)):
        c = r["cur"]["sharp"][r["es"][i]]
        a = r["prev"]["less_sharp"][r["ea"][i], :3]; b = r["prev"]["less_sharp"][r["eb"][i], :3]
        ro, Jqo, Jto = orc.edge_factor(dist["q"], dist["t"], c[:3], a, b, orc.point_s(c))
        close(rr[3 * i:3 * i + 3], ro, "edge r"); close(Jq[3 * i:3 * i + 3], Jqo, "edge Jq"); close(Jt[3 * i:3 * i + 3], Jto, "edge Jt")
    order = np.sort(r["sidx"]); wmap = np.ones(len(r["ps"]), np.float32); wmap[r["sidx"]] = r["sw"]
    for i in order[::max(1, len(order) // 60)]:
        row = 3 * ne + int(np.searchsorted(order, i))
        c = r["cur"]["flat"][r["ps"][i]]
        pj = r["prev"]["less_flat"][r["pa"][i], :3]; pl = r["prev"]["less_flat"][r["pb"][i], :3]; pm = r["prev"]["less_flat"][r["pc"][i], :3]
        ro, Jqo, Jto = orc.plane_factor_modify(dist["q"], dist["t"], c[:3], pj, pl, pm, orc.point_s(c), float(wmap[i]))
        close(rr[row], ro[0], "plane r"); close(Jq[row], Jqo[0], "plane Jq"); close(Jt[row], Jto[0], "plane Jt")


def test_normal_equations_and_gn_step_with_distortion(dist):
    orc, ctx = dist["orc"], dist["ctx"]
    for i, k in enumerate((1, 2)):
        H, g, cost = ctx.normal_equations_result(k)
        Ho, go, co = _oracle_neq(dist, dist["ref"][i])
        close(H, Ho, "H"); close(g, go, "g"); close(cost, co, "cost")
    ctx.gn_step(1, 2)
    for i, k in enumerate((1, 2)):
        Ho, go, _ = _oracle_neq(dist, dist["ref"][i])
        rc, d = orc.gn_solve(Ho, go)
        assert rc == 0
        qo, to = orc.pose_update(dist["q"], dist["t"], d)
        p = ctx.pose(k)
        close(p[:4], qo, "q"); close(p[4:], to, "t")


def test_distortion_off_is_unchanged(api, orc, synth):
    """distortion = 0 (the default, the reference's build) gives bit for bit the poses of a context that never heard of it"""
    cfg = synth.default_cfg(16)
    scans = [synth.scan(cfg, k) for k in range(3)]
    poses = []
    for kw in ({}, {"distortion": 0}):
        ctx = api.Context(api.default_params(16, batch=3, max_points=max(map(len, scans)) + 7, **kw))
        for k, s in enumerate(scans):
            ctx.upload_scan(k, s)
        ctx.extract(0, 3); ctx.set_target_from_slot(0)
        ctx.hot_path(0, 3, np.array([0.001, -0.002, 0.004, 1.0, 0.8, 0.02, -0.01]), vote=True); ctx.synchronize()
        poses.append(np.stack([ctx.pose(k) for k in range(3)]))
        ctx.close()
    assert poses[0].tobytes() == poses[1].tobytes()


def test_frame_loop_with_distortion(api, orc, synth):
    """the reference's frame loop (3 outer iterations x LM <= 4, vote from the 6th frame) with DISTORTION 1 on both sides"""
    rings, nframes = 16, 9
    cfg = synth.default_cfg(rings)
    scans = [synth.scan(cfg, k) for k in range(nframes)]
    P = orc.params(rings)
    ex = [orc.extract(s, P) for s in scans]
    pose0 = np.array([0, 0, 0, 1.0, 0.9, 0.0, 0.0])
    orc.set_nn_mode(1); orc.set_distortion(1)
    try:
        q = pose0[:4].copy(); t = pose0[4:].copy(); rel_o = []
        for k in range(1, nframes):
            q, t = orc.odometry_frame(q, t, ex[k], ex[k - 1], vote=k > 5)
            rel_o.append(np.concatenate([q, t]))
    finally:
        orc.set_nn_mode(0); orc.set_distortion(0)
    rel_o = np.array(rel_o)
    ctx = api.Context(api.default_params(rings, batch=nframes, max_points=max(map(len, scans)), distortion=1))
    for k, s in enumerate(scans):
        ctx.upload_scan(k, s)
    ctx.extract(0, nframes)
    ctx.set_target_from_slot(0)
    rel_d = ctx.odometry_frames(1, nframes - 1, pose0=pose0, n_outer=3, first_frame_index=1)
    ctx.close()
    assert np.abs(rel_d - rel_o).max() < 1e-6, np.abs(rel_d - rel_o).max(axis=1)
    # and it is a different computation from the DISTORTION 0 loop
    ctx = api.Context(api.default_params(rings, batch=nframes, max_points=max(map(len, scans))))
    for k, s in enumerate(scans):
        ctx.upload_scan(k, s)
    ctx.extract(0, nframes); ctx.set_target_from_slot(0)
    rel_0 = ctx.odometry_frames(1, nframes - 1, pose0=pose0, n_outer=3, first_frame_index=1)
    ctx.close()
    assert np.abs(rel_d - rel_0).max() > 1e-4
