"""Error behaviour through the C ABI on the GPU: per-slot failures must not disturb the rest of a batch."""
import numpy as np
import pytest

from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scans(synth):
    cfg = synth.default_cfg(16)
    return [synth.scan(cfg, k) for k in range(3)]


@pytest.fixture(params=["tiles", "walk"])
def org_path(request):
    """both organise paths of ll_organize.hip: the tile-parallel kernels of small calls, the one-pass walk of a batch"""
    from conftest import set_org_path
    set_org_path(request.param)
    yield request.param
    set_org_path("tiles")


def test_empty_and_all_nan_slots_inside_a_batch(api, orc, scans, org_path):
    """slot 1: no points at all; slot 2: only NaNs and points inside minimum_range -> LL_ERR_EMPTY for both (the reference
    would read points[0] of an empty cloud); slots 0 and 3 must come out exactly as if they were alone."""
    P = orc.params(16)
    bad = np.full((500, 4), np.nan, np.float32); bad[::2] = [0.05, 0.05, 0.0, 0.0]
    ctx = api.Context(api.default_params(16, batch=4, max_points=max(map(len, scans))))
    ctx.upload_scan(0, scans[0]); ctx.upload_scan(1, np.zeros((0, 4), np.float32)); ctx.upload_scan(2, bad); ctx.upload_scan(3, scans[1])
    ctx.extract(0, 4)
    assert [ctx.scan_info(k).status for k in range(4)] == [0, -5, -5, 0]
    assert ctx.scan_info(1).n == 0 and ctx.scan_info(2).n_sharp == 0
    for slot, s in ((0, scans[0]), (3, scans[1])):
        ref = orc.extract(s, P)
        f = ctx.features(slot)
        for name in ("sharp", "less_sharp", "flat", "less_flat"):
            assert_bit_equal(f[name], ref[name], name)
    # the whole hot path over the batch: failed slots produce no correspondences, and a slot whose TARGET failed
    # (slot 3, target = slot 2) has nothing to match against
    ctx.set_target_from_slot(0)
    ctx.hot_path(0, 4, np.array([0, 0, 0, 1, 0, 0, 0.0]), vote=True)
    ctx.synchronize()
    assert ctx.pair_info(1).n_edge == 0 and ctx.pair_info(2).n_plane == 0 and ctx.pair_info(3).n_plane == 0
    assert ctx.pair_info(0).n_plane > 50                    # slot 0 against the carry (itself): dense correspondences
    assert np.isfinite(ctx.pose(0)).all() and (ctx.pose(3) == [0, 0, 0, 1, 0, 0, 0]).all()   # singular system: pose untouched
    ctx.close()


def test_ring_longer_than_capacity_is_reported(api, scans, org_path):
    ctx = api.Context(api.default_params(16, batch=1, max_points=len(scans[0]), max_ring_points=1024))
    ctx.upload_scan(0, scans[0])                               # 1800 azimuths per ring > 1024
    ctx.extract(0, 1)
    info = ctx.scan_info(0)
    assert info.status == -4 and info.max_ring > 1024 and info.n_sharp == 0
    ctx.close()


def test_upload_larger_than_max_points_is_refused(api, scans):
    ctx = api.Context(api.default_params(16, batch=1, max_points=1000))
    with pytest.raises(api.LightLoamError) as e:
        ctx.upload_scan(0, scans[0])
    assert e.value.code == -4
    with pytest.raises(api.LightLoamError) as e:
        ctx.extract(0, 2)                                      # slot range out of bounds
    assert e.value.code == -2
    ctx.close()


def test_download_capacity_checks(api, scans):
    import ctypes as C
    ctx = api.Context(api.default_params(16, batch=1, max_points=len(scans[0])))
    ctx.upload_scan(0, scans[0]); ctx.extract(0, 1)
    small = np.zeros((10, 4), np.float32)
    rc = ctx.lib.ll_download_cloud(ctx.h, 0, small.ctypes.data_as(C.c_void_p), 10, None, None)
    assert rc == -4
    ctx.close()


def test_mapping_objects_validate_their_arguments(api, scans):
    ctx = api.Context(api.default_params(16, batch=1, max_points=4096))
    with pytest.raises(api.LightLoamError) as e:
        api.Map(ctx, 0, 10, 10, 10)                                              # capacities must be positive
    assert e.value.code == -2
    m = api.Map(ctx, 100, 100, 50, 50)
    pts = np.zeros((60, 4), np.float32)
    with pytest.raises(api.LightLoamError) as e:
        m.set_scan(pts, pts)                                                     # 60 > 50
    assert e.value.code == -4
    # an optimisation with nothing in the map: not run, pose returned unchanged, no residual blocks
    m.set_map(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32))
    m.set_scan(pts[:10], pts[:10])
    pose, ran = m.optimize([0, 0, 0, 1, 1, 2, 3.0])
    assert not ran and (pose == [0, 0, 0, 1, 1, 2, 3.0]).all()
    m.associate([0, 0, 0, 1, 0, 0, 0.0])
    assert m.counts() == (0, 0)
    H, g, cost = m.normal_equations()
    assert not H.any() and not g.any() and cost == 0.0
    m.close()
    with pytest.raises(api.LightLoamError) as e:
        api.CubeMap(ctx, 100, 100, pool_points=10)                               # pool too small to be a pool
    assert e.value.code == -2
    cm = api.CubeMap(ctx, 4096, 16384, pool_points=1 << 16)
    with pytest.raises(api.LightLoamError) as e:
        cm.prepare([0, 0, 0.0], np.zeros((5000, 4), np.float32), pts)           # 5000 > 4096
    assert e.value.code == -4
    # an empty scan goes through: nothing is added, nothing optimised
    pose, ran = cm.process([0, 0, 0, 1, 0, 0, 0.0], np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32))
    assert not ran and cm.info() == ((10, 10, 5), (0, 0, 0, 0))
    cm.close(); ctx.close()


def test_tile_parallel_entry_points_validate_their_arguments(api):
    import ctypes as C
    ctx = api.Context(api.default_params(16, batch=1, max_points=4096))
    m = api.Map(ctx, 100, 100, 50, 50)
    pts = np.random.default_rng(0).uniform(-1, 1, (40, 4)).astype(np.float32)
    m.set_map(pts, pts); m.set_scan(pts[:10], pts[:10])
    lib = ctx.lib
    pose = np.array([0, 0, 0, 1, 0, 0, 0.0])
    P = pose.ctypes.data_as(C.c_void_p)
    assert lib.ll_map_knn_partial(m.h, P, None, None, None, None) == -2          # stack points but nowhere to put the candidates
    cn, ci, sn, si = m.knn_partial(pose)
    for parts in (0, 65):
        assert lib.ll_map_associate_merged(m.h, P, parts, cn.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p),
                                           sn.ctypes.data_as(C.c_void_p), si.ctypes.data_as(C.c_void_p)) == -2
    with pytest.raises(api.LightLoamError) as e:
        m.set_map_ids(np.arange(40, dtype=np.int32), None)                       # ids of one cloud only
    assert e.value.code == -2
    # without ids a Map numbers its points by position: one part == the plain association
    m.associate(pose); want = m.counts(), m.edges(), m.planes()
    m.associate_merged(cn[None], ci[None], sn[None], si[None], pose)
    assert m.counts() == want[0]
    for rank, world in ((0, 0), (2, 2), (-1, 2), (0, 65)):
        with pytest.raises(api.LightLoamError) as e:
            m.set_row_shard(rank, world)
        assert e.value.code == -2
    # an empty stack needs no buffers at all
    m.set_scan(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32))
    assert lib.ll_map_knn_partial(m.h, P, None, None, None, None) == 0
    assert lib.ll_map_associate_merged(m.h, P, 2, None, None, None, None) == 0 and m.counts() == (0, 0)
    m.close()
    cm = api.CubeMap(ctx, 4096, 16384, pool_points=1 << 16)
    for rank, world in ((0, 0), (2, 2), (-1, 2), (0, 65)):
        with pytest.raises(api.LightLoamError) as e:
            cm.set_shard(rank, world)
        assert e.value.code == -2
    cm.set_shard(1, 2); cm.set_shard(0, 1)                                       # allowed while the map is empty; (0, 1) = unsplit again
    pose2, ran = cm.process(pose, np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32))
    assert not ran
    cm.close(); ctx.close()
