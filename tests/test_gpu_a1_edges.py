"""a1 on the DEVICE at the places where a libm bit can change an integer result (scanRegistration.cpp:139-168, :177-208).

Every ring of the synthetic scans sits at a bin centre, so the regular parity tests never feed k_classify a point whose
t = z / sqrt(x^2 + y^2) is near a ring threshold, nor an azimuth next to a wrap / halfPassed boundary.  Here the device gets
  * its bit-exact atanf / atan2f / "/ M_PI" / z / sqrtf(..) restatements compared with the host glibc on the GPU box,
  * points placed AT every ring threshold and +-1..3 float steps around it, for the 16 / 32 / 64-ring models and the linear model,
  * >= 10^6 points with elevations spread over the whole range (and beyond), in sweep order and in random order,
  * azimuths packed around every boundary of the wrap / halfPassed logic,
and laserCloud (ring order + intensity bits) must equal the oracle's, which calls glibc exactly where the reference does.
"""
import numpy as np
import pytest

import scangen
from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["tiles", "walk"])
def org_path(request):
    """Every test of this file runs through both organise paths of ll_organize.hip (tile-parallel kernels of small calls,
    and k_organize, the one-pass walk of a batch)."""
    from conftest import set_org_path
    set_org_path(request.param)
    yield request.param
    set_org_path("tiles")

MODELS = {
    "VLP16": dict(rings=16, prm=dict(minimum_range=0.3)),
    "HDL32": dict(rings=32, prm=dict(minimum_range=0.3)),
    "HDL64": dict(rings=64, prm=dict()),
    "linear128": dict(rings=128, prm=dict(ring_model=1, lower_bound=-25.0, up_bound=15.0, minimum_range=0.3)),
    "linear40": dict(rings=40, prm=dict(ring_model=1, lower_bound=-16.0, up_bound=7.0, minimum_range=0.3)),
}


def _same_bits(a, b):
    a = np.asarray(a); b = np.asarray(b)
    if a.dtype.kind == "f":
        return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    return a == b


def _specials():
    v = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.17549435e-38, 3.4028235e38, -3.4028235e38,
         0.4375, 0.6875, 1.1875, 2.4375, 2.0 ** -29, 2.0 ** 25, 2.0 ** 60, 2.0 ** -60, 5.0, 8.0, 3.0e-5]
    return np.array(v, np.float32)


def test_device_libm_restatements_equal_glibc_on_this_box(api, orc):
    """ll_atanf / ll_atan2f / ll_atan2f_finite / ll_div_pi_f32 and the correctly rounded z / sqrtf(x*x + y*y) as the GPU
    evaluates them (hipcc, -ffp-contract=off, -fhip-fp32-correctly-rounded-divide-sqrt) against the host libm."""
    rng = np.random.default_rng(2)
    ctx = api.Context(api.default_params(64, batch=1, max_points=1024))
    sp = _specials()
    # atanf: every exponent, both signs; dense around the reduction boundaries and the tiny / huge cut-offs
    bits = rng.integers(0, 1 << 32, 3_000_000, dtype=np.uint64).astype(np.uint32)
    a = bits.view(np.float32)
    for c in (0.4375, 0.6875, 1.1875, 2.4375, 2.0 ** -29, 2.0 ** 25, 1.0):
        k = np.float32(c).view(np.uint32).astype(np.int64) + np.arange(-4096, 4096)
        a = np.concatenate([a, k.astype(np.uint32).view(np.float32), -k.astype(np.uint32).view(np.float32)])
    a = np.concatenate([a, sp])
    bad = ~_same_bits(ctx.exact_math(0, a), orc.libm(0, a))
    assert not bad.any(), f"atanf: {int(bad.sum())} of {len(a)} differ, first {a[bad][:4]!r}"
    bad = ~_same_bits(ctx.exact_math(3, a), orc.libm(2, a))
    assert not bad.any(), f"(float)((double)a / M_PI): {int(bad.sum())} differ, first {a[bad][:4]!r}"
    # atan2f: lidar-like magnitudes, every exponent combination, the special-value grid, near-axis points
    n = 2_000_000
    y = (rng.standard_normal(n) * 10 ** rng.uniform(-3, 2.2, n)).astype(np.float32)
    x = (rng.standard_normal(n) * 10 ** rng.uniform(-3, 2.2, n)).astype(np.float32)
    yb = rng.integers(0, 1 << 32, 1_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    xb = rng.integers(0, 1 << 32, 1_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    gy, gx = np.meshgrid(sp, sp)
    ax_y = (rng.standard_normal(200_000) * 1e-6).astype(np.float32); ax_x = (rng.standard_normal(200_000) * 30).astype(np.float32)
    y = np.concatenate([y, yb, gy.ravel(), ax_y, ax_x]); x = np.concatenate([x, xb, gx.ravel(), ax_x, ax_y])
    want = orc.libm(1, y, x)
    for op, name in ((1, "ll_atan2f"), (2, "ll_atan2f_finite")):
        bad = ~_same_bits(ctx.exact_math(op, y, x), want)
        assert not bad.any(), f"{name}: {int(bad.sum())} of {len(y)} differ, first (y, x) = {list(zip(y[bad][:3], x[bad][:3]))}"
    # z / sqrtf(x*x + y*y): lidar-like and extreme operands (correctly rounded sqrt and division, no contraction)
    z = np.concatenate([(rng.standard_normal(n) * 10 ** rng.uniform(-2, 2, n)).astype(np.float32), gy.ravel()])
    px = np.concatenate([(rng.standard_normal(n) * 10 ** rng.uniform(-2, 2, n)).astype(np.float32), gx.ravel()])
    py = np.concatenate([(rng.standard_normal(n) * 10 ** rng.uniform(-2, 2, n)).astype(np.float32), gx.ravel()[::-1]])
    bad = ~_same_bits(ctx.exact_math(4, z, px, py), orc.libm(3, z, px, py))
    assert not bad.any(), f"z / sqrtf(x*x + y*y): {int(bad.sum())} of {len(z)} differ"
    ctx.close()


def _threshold_points(orc, P, rng, reach=3):
    """points whose t = z / sqrtf(x*x + y*y) is a ring threshold of P or 1..reach float steps from it, in several azimuth
    directions and at several ranges; returns (xyz1 [n, 4] float32, keys of their t, threshold keys)"""
    thr = scangen.ring_thresholds(orc, P)
    finite = thr[(thr > scangen.float_key(np.float32(-np.inf))) & (thr < scangen.float_key(np.float32(np.inf)))]
    pts = []
    steps = np.arange(-reach, reach + 1)
    tk = (finite[:, None] + steps[None, :]).ravel()
    t = scangen.key_float(tk).astype(np.float64)
    # exact directions: sqrt(x^2 + y^2) is a power of two, so z = t * rho is exact and t comes back bit for bit
    for rho, (cx, cy) in ((8.0, (1, 0)), (16.0, (0, 1)), (32.0, (-1, 0)), (8.0, (0, -1))):
        pts.append(np.stack([np.full_like(t, rho * cx), np.full_like(t, rho * cy), t * rho], axis=1))
    # inexact directions: rho is not a float; take the z candidates around t * rho and keep those whose t lands in the window
    for ang, rho in ((0.3, 11.0), (1.9, 23.7), (-2.4, 6.3), (-0.77, 47.1), (3.0, 9.9)):
        x = np.float32(rho * np.cos(ang)); y = np.float32(rho * np.sin(ang))
        r32 = np.sqrt(np.float32(x * x) + np.float32(y * y), dtype=np.float32)
        z0 = (t * np.float64(r32)).astype(np.float32)
        for dz in range(-2, 3):
            z = (z0.view(np.int32) + dz).view(np.float32)
            pts.append(np.stack([np.full(len(z), x, np.float64), np.full(len(z), y, np.float64), z.astype(np.float64)], axis=1))
    p = np.concatenate(pts).astype(np.float32)
    p = p[np.isfinite(p).all(axis=1)]
    p = p[rng.permutation(len(p))]
    keys = scangen.float_key(orc.libm(3, p[:, 2], p[:, 0], p[:, 1]))
    return np.concatenate([p, np.zeros((len(p), 1), np.float32)], axis=1), keys, finite


@pytest.mark.parametrize("model", list(MODELS))
def test_ring_ids_at_every_threshold(api, orc, model):
    """t placed on every threshold of the model and +-1..3 float steps around it: the device's threshold search (op 5, what
    k_classify runs), the device's direct evaluation of the formula chain (op 6) and the oracle (glibc) give the same ring;
    then the same points as ONE scan through ll_extract_batch: laserCloud bit-exact."""
    rings, prm = MODELS[model]["rings"], MODELS[model]["prm"]
    P = orc.params(rings, **prm)
    rng = np.random.default_rng(rings)
    pts, keys, thr = _threshold_points(orc, P, rng)
    # the construction must really sit on the edges: every finite threshold is hit exactly and from both sides
    d = keys[:, None] - thr[None, :]
    assert ((d == 0).any(axis=0)).all() and ((d == -1).any(axis=0)).all() and ((d == 1).any(axis=0)).all(), "threshold coverage"
    want = orc.scan_ids(pts, P)
    assert len(np.unique(want[want >= 0])) == rings and (want < 0).any()
    ctx = api.Context(api.default_params(rings, batch=1, max_points=len(pts) + 8, max_ring_points=8192, **prm))
    got5 = ctx.exact_math(5, pts[:, 2], pts[:, 0], pts[:, 1])
    got6 = ctx.exact_math(6, pts[:, 2], pts[:, 0], pts[:, 1])
    assert (got5 == want).all(), f"{model}: threshold search differs from the oracle at {int((got5 != want).sum())} edge points"
    assert (got6 == want).all(), f"{model}: device formula chain differs from the oracle at {int((got6 != want).sum())} edge points"
    rc, cloud, ss, se = orc.organize(pts, P)
    assert rc == 0
    ctx.upload_scan(0, pts)
    ctx.extract(0, 1)
    assert ctx.scan_info(0).status == 0
    got, gss, gse = ctx.cloud(0)
    assert_bit_equal(got, cloud, f"{model} laserCloud of the edge points")
    assert (gss == ss).all() and (gse == se).all()
    ctx.close()


@pytest.mark.parametrize("model,n", [("HDL64", 300_000), ("linear128", 390_000), ("HDL32", 140_000), ("VLP16", 70_000)])
def test_elevations_spread_over_the_whole_range(api, orc, model, n):
    """4 scans per model, 3.6 million points in total (ring 0 is two bins wide -- int() truncates towards zero -- and must fit 8192): elevations anywhere (3 deg beyond both ends: rejected rings), in
    sweep order and in random order (the halfPassed flag flips at an arbitrary point), NaN / inf returns, points inside
    minimum_range.  laserCloud bit-exact; for the 64-ring model also the four feature clouds (4.7 k points per ring: the
    32-row instantiation of the feature kernel)."""
    rings, prm = MODELS[model]["rings"], MODELS[model]["prm"]
    P = orc.params(rings, **prm)
    lo = P.lower_bound if (rings == 64 or P.ring_model == 1) else (-15.0 if rings == 16 else -92.0 / 3.0)
    hi = P.up_bound if (rings == 64 or P.ring_model == 1) else (15.0 if rings == 16 else 92.0 / 3.0 - 20.0)
    rng = np.random.default_rng(1000 + rings)
    scans = [scangen.spread_scan(rng, n, lo - 3.0, hi + 3.0, sweep=(k % 2 == 0)) for k in range(4)]
    ctx = api.Context(api.default_params(rings, batch=4, max_points=n, max_ring_points=8192, **prm))
    for k, s in enumerate(scans):
        ctx.upload_scan(k, s)
    ctx.extract(0, 4)
    for k, s in enumerate(scans):
        assert ctx.scan_info(k).status == 0, (model, k, ctx.scan_info(k).status, ctx.scan_info(k).max_ring)
        if model == "HDL64":
            ref = orc.extract(s, P)
            f = ctx.features(k)
            for name in ("sharp", "less_sharp", "flat", "less_flat"):
                assert_bit_equal(f[name], ref[name], f"{model} scan {k} {name}")
            cloud, ss, se = ref["cloud"], ref["scan_start"], ref["scan_end"]
        else:
            rc, cloud, ss, se = orc.organize(s, P)
            assert rc == 0
        got, gss, gse = ctx.cloud(k)
        assert len(got) == len(cloud) and len(cloud) > 0.5 * n
        assert_bit_equal(got, cloud, f"{model} scan {k} laserCloud")
        assert (gss == ss).all() and (gse == se).all()
    ctx.close()


@pytest.mark.parametrize("s0,last_gap", [(0.3, 0.01), (np.pi / 2, 0.2), (3.1, 0.05), (-3.1, 1.0), (-1.2, 3.3), (0.0, 2 * np.pi - 0.3)])
def test_azimuths_at_the_wrap_and_half_sweep_boundaries(api, orc, s0, last_gap):
    """~100 000 points per scan within +-2.5e-6 rad (~20 float steps) of startOri - pi/2, startOri + 3 pi/2, the halfPassed
    threshold startOri + pi and the endOri windows, on both sides of the flip; start azimuths next to +-pi, and sweeps that end
    well short of / beyond a revolution (the 3 pi / pi fix of endOri, :120-126).  The intensity bits carry relTime, so
    laserCloud bit-exact means every compare of :181-203 went the reference's way."""
    rng = np.random.default_rng(int(abs(s0) * 1000) + 7)
    scan = scangen.wrap_scan(rng, s0, last_gap)
    P = orc.params(16)
    # the crowding is real: raw ori values on both sides of (double)startOri - pi/2 within two float steps, or the image of it
    ori = -orc.libm(1, scan[:, 1], scan[:, 0]).astype(np.float64)
    so = float(np.float32(ori[0]))
    for b in (so - np.pi / 2, so + 3 * np.pi / 2, so + np.pi - 2 * np.pi, so + np.pi):
        if -np.pi < b < np.pi:
            tol = max(3 * float(np.spacing(np.float32(abs(b)))), 2e-7)
            assert ((ori < b) & (ori > b - tol)).any() and ((ori > b) & (ori < b + tol)).any(), f"no points next to {b}"
    rc, cloud, ss, se = orc.organize(scan, P)
    assert rc == 0
    ctx = api.Context(api.default_params(16, batch=1, max_points=len(scan), max_ring_points=8192))
    ctx.upload_scan(0, scan)
    ctx.extract(0, 1)
    assert ctx.scan_info(0).status == 0
    got, gss, gse = ctx.cloud(0)
    assert_bit_equal(got, cloud, f"wrap scan s0 = {s0}")
    assert (gss == ss).all() and (gse == se).all()
    ctx.close()
