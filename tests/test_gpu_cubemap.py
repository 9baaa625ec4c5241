"""SURVEY.md section 8f #2, second stage: laserMapping's cube map on the device (laserMapping.cpp:1584-1821, :2101-2165)
against the oracle's restatement.  With the SAME pose handed to both sides the map state must stay bit-identical frame
after frame (cube contents, the clouds gathered from the map, the down-sized scan); the free-running loop (each side
using its own optimised pose) must agree to f64 rounding."""
import numpy as np
import pytest

from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu


def _pose7(pose3, offset=(0.0, 0.0, 0.0)):
    x, y, yaw = pose3
    return np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2), x + offset[0], y + offset[1], offset[2]])


@pytest.fixture(scope="module")
def frames(orc, synth):
    cfg = synth.default_cfg(16)
    P = orc.params(16)
    out = []
    for k in range(7):
        f = orc.extract(synth.scan(cfg, k), P)
        out.append(dict(corner=f["less_sharp"], surf=f["less_flat"], pose3=synth.pose(cfg, k)))
    return out


def _nonempty(oc):
    return [(s, i) for s in (0, 1) for i in range(4851) if len(oc.cube(s, i))]


@pytest.mark.parametrize("offset", [(0.0, 0.0, 0.0), (-431.0, 512.5, 30.0)])
def test_same_poses_give_the_same_map(api, orc, frames, offset):
    """offset (-431, 512.5, 30) starts the vehicle far from the origin: the first prepare shifts the cube window along
    all three axes (:1595-1778), and the negative coordinates take the `+ 25.0 < 0` branches (:1588-1593, :2112-2117)."""
    ctx = api.Context(api.default_params(16, batch=1, max_points=4096))
    dc = api.CubeMap(ctx, 4096, 32768, pool_points=1 << 18)
    oc = orc.CubeMap()
    for k, f in enumerate(frames):
        pose = _pose7(f["pose3"], offset)
        pose[4:] += [0.01 * k, -0.02, 0.005]                                   # any pose will do, as long as both sides get it
        oc.prepare(pose[4:], f["corner"], f["surf"]); dc.prepare(pose[4:], f["corner"], f["surf"])
        cen, cnt = dc.info()
        assert cen == oc.center()
        for which in range(4):
            assert_bit_equal(dc.cloud(which), oc.cloud(which), f"frame {k} cloud {which}")
        assert cnt == tuple(len(oc.cloud(w)) for w in range(4))
        oc.update(pose[:4], pose[4:]); dc.update(pose)
        cubes = _nonempty(oc)
        assert len(cubes) >= 2
        for s, i in cubes:
            assert_bit_equal(dc.cube(s, i), oc.cube(s, i), f"frame {k} cube {('corner', 'surf')[s]} {i}")
        total_d = sum(len(dc.cube(s, i, cap=1 << 16)) for s in (0, 1) for i in range(0, 4851, 97))   # spot check of empty cubes
        total_o = sum(len(oc.cube(s, i)) for s in (0, 1) for i in range(0, 4851, 97))
        assert total_d == total_o
    if offset != (0.0, 0.0, 0.0):
        assert oc.center() != (10, 10, 5)
    dc.close(); ctx.close(); oc.close()


def test_pool_compaction_keeps_the_map(api, orc, frames):
    """a pool just large enough to force compactions between frames"""
    ctx = api.Context(api.default_params(16, batch=1, max_points=4096))
    dc = api.CubeMap(ctx, 4096, 32768, pool_points=40000)
    oc = orc.CubeMap()
    for k in range(12):
        f = frames[k % len(frames)]
        pose = _pose7(frames[k % len(frames)]["pose3"]); pose[4] += 0.3 * (k // len(frames))
        oc.prepare(pose[4:], f["corner"], f["surf"]); dc.prepare(pose[4:], f["corner"], f["surf"])
        oc.update(pose[:4], pose[4:]); dc.update(pose)
    for s, i in _nonempty(oc):
        assert_bit_equal(dc.cube(s, i), oc.cube(s, i), f"cube {s} {i}")
    dc.close(); ctx.close(); oc.close()


def test_free_running_mapping_matches_oracle(api, orc, frames):
    ctx = api.Context(api.default_params(16, batch=1, max_points=4096))
    dc = api.CubeMap(ctx, 4096, 32768, pool_points=1 << 18)
    oc = orc.CubeMap()
    for k, f in enumerate(frames):
        guess = _pose7(f["pose3"]); guess[4:] += [0.08, -0.05, 0.02]           # what odometry would hand over
        oc.prepare(guess[4:], f["corner"], f["surf"])
        q, t, ran_o = oc.optimize(guess[:4], guess[4:]); oc.update(q, t)
        pose, ran_d = dc.process(guess, f["corner"], f["surf"])
        assert ran_d == ran_o == (k > 0)
        assert np.abs(pose[:4] - q).max() < 1e-6 and np.abs(pose[4:] - t).max() < 1e-6, (k, pose, q, t)
    # the maps were built from poses that agree to ~1e-9: same cubes, same sizes up to a rare voxel-boundary flip
    for s, i in _nonempty(oc):
        a, b = dc.cube(s, i), oc.cube(s, i)
        assert abs(len(a) - len(b)) <= max(2, len(b) // 500)
    dc.close(); ctx.close(); oc.close()


def test_process_slot_equals_process_with_downloaded_clouds(api, synth):
    """the device-to-device feed (slot -> map stage) gives exactly what the host round trip gives"""
    cfg = synth.default_cfg(16)
    n = 5
    scans = [synth.scan(cfg, k) for k in range(n)]
    ctx = api.Context(api.default_params(16, batch=n, max_points=max(map(len, scans))))
    for k, s in enumerate(scans):
        ctx.upload_scan(k, s)
    ctx.extract(0, n)
    a = api.CubeMap(ctx, 4096, 32768, pool_points=1 << 18)
    b = api.CubeMap(ctx, 4096, 32768, pool_points=1 << 18)
    for k in range(n):
        guess = _pose7(synth.pose(cfg, k)); guess[4:] += [0.05, 0.02, -0.01]
        f = ctx.features(k)
        pa, ra = a.process(guess, f["less_sharp"], f["less_flat"])
        pb, rb = b.process_slot(guess, k)
        assert ra == rb and (pa == pb).all()
    assert a.info() == b.info()
    for which in range(4):
        assert_bit_equal(a.cloud(which), b.cloud(which), f"cloud {which}")
    a.close(); b.close(); ctx.close()


def test_a_pool_too_small_for_the_surface_cloud_only_leaves_the_map_as_it_was(api, orc, synth):
    """ll_cubemap_update decides the pool capacity of BOTH cloud types before it commits either (round-3 advice).  A pool just large
    enough for one scan fills up as the vehicle moves on (every frame 60 m further: new cubes each time); the surface cloud -- three
    times the corner cloud -- runs out first.  That update fails with LL_ERR_CAPACITY, nothing has changed (the cubes filled so far
    still hold what they held, corner AND surface), the map is not marked broken, and the same call fails the same way again."""
    cfg = synth.default_cfg(64)
    fe = orc.extract(synth.scan(cfg, 0), orc.params(64))
    corner, surf = fe["less_sharp"], fe["less_flat"]
    assert len(surf) > 2 * len(corner)
    ctx = api.Context(api.default_params(64, batch=1, max_points=4096))
    cap_s = len(surf) + 64
    dc = api.CubeMap(ctx, len(corner) + 64, cap_s, pool_points=cap_s)
    x0, y0, yaw = synth.pose(cfg, 0)
    failed_at = None
    before = None
    for k in range(12):
        pose = _pose7((x0 + 60.0 * k, y0, yaw))
        dc.prepare(pose[4:], corner, surf)
        snapshot = [dc.cube(s, i, cap=1 << 17).copy() for s in (0, 1) for i in range(0, 4851, 3)]
        try:
            dc.update(pose)
        except api.LightLoamError as e:
            assert e.code == -4, e                                              # LL_ERR_CAPACITY
            failed_at, before = k, snapshot
            break
    assert failed_at is not None and failed_at >= 1, "the pool never filled up"
    assert sum(len(c) for c in before) > 0
    with pytest.raises(api.LightLoamError) as e2:
        dc.update(pose)
    assert e2.value.code == -4, e2.value                                        # again LL_ERR_CAPACITY, not -7 (LL_ERR_STATE)
    after = [dc.cube(s, i, cap=1 << 17) for s in (0, 1) for i in range(0, 4851, 3)]
    assert all(a.tobytes() == b_.tobytes() for a, b_ in zip(after, before))     # neither cloud type was touched
    dc.prepare(pose[4:], corner, surf)                                          # still usable
    dc.close(); ctx.close()
