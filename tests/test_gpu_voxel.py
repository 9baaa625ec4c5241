"""pcl::VoxelGrid on whole clouds (laserMapping's downSizeFilterCorner / downSizeFilterSurf): ll_voxel_grid against the
oracle's restatement, bit for bit (voxel membership and order exact, f32 centroid sums in input order)."""
import numpy as np
import pytest

from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(api):
    c = api.Context(api.default_params(16, batch=1, max_points=4096))
    yield c
    c.close()


def _cloud(rng, n, extent, lattice=None):
    p = np.zeros((n, 4), np.float32)
    p[:, :3] = rng.uniform(-1, 1, (n, 3)) * extent
    if lattice:
        p[:, :3] = np.round(p[:, :3] / lattice) * lattice           # points exactly on voxel boundaries, duplicates
    p[:, 3] = rng.uniform(0, 64, n)
    return p


@pytest.mark.parametrize("n,extent,leaf,lattice", [
    (1, (1, 1, 1), 0.4, None), (7, (0.1, 0.1, 0.1), 0.4, None), (5000, (30, 30, 3), 0.4, None),
    (60000, (60, 60, 8), 0.8, None), (300000, (120, 120, 10), 0.8, None), (20000, (20, 20, 2), 0.4, 0.2),
    (4097, (10, 10, 10), 0.2, 0.1),
])
def test_matches_oracle(ctx, orc, n, extent, leaf, lattice):
    rng = np.random.default_rng(n)
    pts = _cloud(rng, n, np.array(extent), lattice)
    want = orc.voxel_grid(pts, leaf)
    got = ctx.voxel_grid(pts, leaf)
    assert len(got) == len(want)
    assert_bit_equal(got, want, f"voxel grid n={n} leaf={leaf}")


def test_feature_clouds_of_a_scan(ctx, orc, synth):
    cfg = synth.default_cfg(64)
    f = orc.extract(synth.scan(cfg, 2), orc.params(64))
    for name, leaf in (("less_sharp", 0.4), ("less_flat", 0.8)):                   # laserMapping.cpp:2363-2369
        assert_bit_equal(ctx.voxel_grid(f[name], leaf), orc.voxel_grid(f[name], leaf), name)


def test_leaf_too_small_passes_the_cloud_through(ctx, orc):
    """more than INT_MAX voxels: PCL warns and returns the input unchanged"""
    pts = _cloud(np.random.default_rng(5), 1000, np.array([400, 400, 400]))
    pts[:, :3] = np.abs(pts[:, :3]) + 0.5                                             # no negative zero to lose its sign
    want = orc.voxel_grid(pts, 0.2)
    assert len(want) == len(pts)
    assert_bit_equal(ctx.voxel_grid(pts, 0.2), want, "too small")


def test_empty_and_capacity(ctx, api):
    assert len(ctx.voxel_grid(np.zeros((0, 4), np.float32), 0.4)) == 0
    with pytest.raises(api.LightLoamError):
        ctx.voxel_grid(np.zeros((4, 4), np.float32), 0.0)
