"""Two configurations of the extract stage that must not change a bit of its output:

* ll_params.voxel_sort_ranks = 1 -- the VoxelGrid sort's match-any ranking (ll_features.hip), which ll_create selects by itself for a
  device that fails its LDS lane-order check (round-5 advice: such a device used to be refused).  Forced here, against the oracle
  and the committed fixtures, through both organise paths and every capacity tier.
* ll_params.input_stride_floats = 3 -- the resident raw scan as packed (x, y, z): scanRegistration.cpp:105-106 converts the message to
  PointXYZ and never sees a 4th float.  Synchronous upload (any caller stride), the asynchronous single-scan and strided uploads.
"""
import glob
import os

import numpy as np
import pytest

from conftest import assert_bit_equal, set_org_path

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))

SHAPES = {
    "S64": dict(rings=64),
    "S16": dict(rings=16),
    "S32": dict(rings=32),
    "S64_azmajor_jitter_nan": dict(rings=64, order=1, az_jitter_deg=0.4, drop_prob=0.03, emit_nan=1),
    "S64_ringmajor_jitter": dict(rings=64, az_jitter_deg=0.4),
    "S128_linear_model": dict(rings=128),
    "HDL64E_table_kitti_order": dict(rings=64, gen="hdl64", order="kitti"),
    "HDL64E_table_firing_order": dict(rings=64, gen="hdl64", order="firing"),
}
RING_MODEL = {128: dict(ring_model=1, lower_bound=-25.0, up_bound=15.0, minimum_range=0.3)}


def _scans(shape, synth):
    kw = dict(SHAPES[shape]); rings = kw.pop("rings")
    extra_prm = {}
    if kw.pop("gen", None) == "hdl64":
        import scangen
        scans = [scangen.hdl64_scan(k, **kw) for k in range(2)]
        extra_prm = dict(max_ring_points=4608)
    else:
        cfg = synth.default_cfg(rings, **kw)
        scans = [synth.scan(cfg, k) for k in range(2)]
    return rings, scans, dict(RING_MODEL.get(rings, {})), extra_prm


def _check_against_oracle(ctx, ref, what):
    for k, r in enumerate(ref):
        assert ctx.scan_info(k).status == 0
        cloud, ss, se = ctx.cloud(k)
        assert_bit_equal(cloud, r["cloud"], f"{what} scan {k} laserCloud")
        assert (ss == r["scan_start"]).all() and (se == r["scan_end"]).all()
        f = ctx.features(k)
        for name in ("sharp", "less_sharp", "flat", "less_flat"):
            assert_bit_equal(f[name], r[name], f"{what} scan {k} {name}")


@pytest.mark.parametrize("org", ["tiles", "walk"])
@pytest.mark.parametrize("shape", list(SHAPES))
def test_match_any_sort_is_bit_exact(shape, org, api, orc, synth):
    rings, scans, extra, extra_prm = _scans(shape, synth)
    ref = [orc.extract(s, orc.params(rings, **extra)) for s in scans]
    set_org_path(org)
    try:
        ctx = api.Context(api.default_params(rings, batch=2, max_points=max(map(len, scans)) + 7, voxel_sort_ranks=1, **extra, **extra_prm))
        for k, s in enumerate(scans):
            ctx.upload_scan(k, s)
        ctx.extract(0, 2)
        _check_against_oracle(ctx, ref, f"{shape}-{org} match-any sort")
        ctx.close()
    finally:
        set_org_path("tiles")


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
@pytest.mark.parametrize("variant", [dict(voxel_sort_ranks=1), dict(input_stride_floats=3), dict(voxel_sort_ranks=1, input_stride_floats=3)],
                         ids=["match_any", "xyz12", "both"])
def test_variants_reproduce_golden(path, variant, api):
    G = np.load(path)
    rings = int(G["rings"])
    ctx = api.Context(api.default_params(rings, batch=2, max_points=max(len(G["scan0"]), len(G["scan1"])), **variant))
    ctx.upload_scan(0, G["scan0"]); ctx.upload_scan(1, G["scan1"])
    ctx.extract(0, 2)
    for k in (0, 1):
        cloud, ss, se = ctx.cloud(k)
        assert_bit_equal(cloud, G[f"s{k}_cloud"], f"cloud[{k}]")
        f = ctx.features(k)
        for key in ("sharp", "less_sharp", "flat", "less_flat"):
            assert_bit_equal(f[key], G[f"s{k}_{key}"], f"{key}[{k}]")
    ctx.set_target_from_slot(0)
    ctx.hot_path(1, 1, G["pose"], vote=True)
    es, ea, eb = ctx.edge_corr(1); ps, pa, pb, pc = ctx.plane_corr(1)
    for got, key in ((es, "e_src"), (ea, "e_a"), (eb, "e_b"), (ps, "p_src"), (pa, "p_a"), (pb, "p_b"), (pc, "p_c")):
        assert (got == G[key]).all(), key
    ctx.close()


@pytest.mark.parametrize("org", ["tiles", "walk"])
@pytest.mark.parametrize("shape", list(SHAPES))
def test_packed_xyz_input_is_bit_exact(shape, org, api, orc, synth):
    """every shape through both organise paths with the raw scan resident as 12-byte points; the caller hands over 4-float and
    3-float arrays (ll_upload_scan repacks either into the resident layout)"""
    rings, scans, extra, extra_prm = _scans(shape, synth)
    ref = [orc.extract(s, orc.params(rings, **extra)) for s in scans]
    set_org_path(org)
    try:
        ctx = api.Context(api.default_params(rings, batch=2, max_points=max(map(len, scans)) + 7, input_stride_floats=3, **extra, **extra_prm))
        ctx.upload_scan(0, scans[0])                                       # (n, 4): repacked on the host
        ctx.upload_scan(1, np.ascontiguousarray(scans[1][:, :3]))          # (n, 3): copied as it is
        ctx.extract(0, 2)
        _check_against_oracle(ctx, ref, f"{shape}-{org} packed xyz")
        ctx.close()
    finally:
        set_org_path("tiles")


@pytest.mark.parametrize("stride", [3, 4])
def test_async_uploads_in_the_resident_layout(stride, api, orc, synth):
    """ll_upload_scan_async / ll_upload_scans_async_strided take their buffers in the context's resident layout (16- or 12-byte points);
    the streamed slots extract to the same clouds as synchronously uploaded ones, and the hot path to the same poses"""
    rings = 64
    cfg = synth.default_cfg(rings)
    scans = [synth.scan(cfg, k) for k in range(5)]
    mp = max(map(len, scans))
    ctx = api.Context(api.default_params(rings, batch=5, max_points=mp, input_stride_floats=stride))
    ref = api.Context(api.default_params(rings, batch=5, max_points=mp))
    for k, s in enumerate(scans):
        ref.upload_scan(k, s)
    staging = api.PinnedStaging(3, (mp + 63) // 64 * 64, stride)
    for i in range(3):
        staging.put(i, scans[2 + i])
    pinned = [api.PinnedScan(scans[k], stride) for k in range(2)]
    ctx.upload_scan_async(0, pinned[0]); ctx.upload_scans_async(1, pinned[1:2])
    ctx.upload_staging_async(2, staging, 0, 3)
    ctx.stream_record(1, 0); ctx.stream_wait(0, 0)
    pose = np.array([0, 0, 0, 1, 0.9, 0.0, 0.0])
    for c in (ctx, ref):
        c.extract(0, 1); c.set_target_from_slot(0)
        c.hot_path(1, 4, np.tile(pose, (4, 1)), vote=True)
        c.synchronize()
    for k in range(1, 5):
        a, b = ctx.features(k), ref.features(k)
        for name in ("sharp", "less_sharp", "flat", "less_flat"):
            assert_bit_equal(a[name], b[name], f"slot {k} {name}")
        assert ctx.pose(k).tobytes() == ref.pose(k).tobytes()
        assert ctx.pair_info(k).n_plane_selected > 0
    for p in pinned:
        p.close()
    staging.close(); ctx.close(); ref.close()


def test_two_stream_association_stage_changes_no_result(api, synth):
    """ll_hot_path_batch over >= 512 scans builds the target grids of one quarter of the range beside the search of the quarter before it
    (two HIP streams, events); ll_set_two_stream(0) runs kernel after kernel.  Same correspondences and poses bit for bit, also across
    repeated calls (the events are reused) and with the per-kernel profiler on (the stage is then one interval)."""
    rings, B = 16, 1056                                   # not a multiple of the piece size: ragged last quarter
    cfg = synth.default_cfg(rings)
    scans = [synth.scan(cfg, k) for k in range(9)]
    ctx = api.Context(api.default_params(rings, batch=B + 1, max_points=max(map(len, scans))))
    ctx.upload_scan(B, scans[0]); ctx.extract(B, 1); ctx.set_target_from_slot(B)
    for i in range(B):
        ctx.upload_scan(i, scans[(i + 1) % 9])
    pose = np.array([0, 0, 0, 1, 0.9, 0.0, 0.0])
    ctx.set_pose_guess(0, B, np.tile(pose, (B, 1)))

    def run():
        ctx.hot_path(0, B, None, vote=True); ctx.synchronize()
        return (np.stack([ctx.pose(i) for i in range(0, B, 7)]).tobytes(),
                [tuple(a.tobytes() for a in ctx.plane_corr(i)) for i in (0, 1, 263, 264, 527, 528, 791, 792, B - 1)])

    ctx.set_two_stream(False); one = run()
    ctx.set_two_stream(True); two = run(); again = run()
    assert one == two == again
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    prof_run = run()
    prof = ctx.profile_read(reset=True); ctx.profile_enable(False)
    assert prof_run == one
    assert prof["k_build_grid||k_associate"][1] == 1 and prof["k_build_grid"][1] == 0 and prof["k_associate"][1] == 0
    assert all(ctx.pair_info(i).n_plane_selected > 0 for i in range(0, B, 97))
    ctx.close()
