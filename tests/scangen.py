"""numpy scan generators for the a1 edge-case tests and the KITTI-shaped (config 3) stand-in.

The synthetic generator of the product tree (light-loam_amd/host/ll_synth.c) puts every ring at the exact bin centre of
scanRegistration.cpp:162.  Real sensors do not: the HDL-64E's lasers sit at two different angular spacings, are mounted
at different heights and fire with per-laser azimuth offsets, so the elevation the reference computes from (x, y, z)
falls anywhere inside a ring's bin -- and sometimes across its edge.  These generators reproduce that, in the point
orders a KITTI .bin file / a raw firing sequence have.  Test infrastructure only.
"""
import numpy as np

import lightloam_amd  # noqa: F401  (import shim of the hyphenated package directory)

from lightloam_amd.hdl64 import HDL64_ELEV_DEG, hdl64_scan, _street_scene, _raycast, pose as hdl64_pose  # noqa: E402,F401


def float_key(f):
    """order-preserving int64 key of float32 values (-inf .. +inf), as ll_float_key"""
    i = np.asarray(f, np.float32).view(np.int32).astype(np.int64)
    return np.where(i >= 0, i, i ^ 0x7fffffff)


def key_float(k):
    k = np.asarray(k, np.int64)
    i = np.where(k >= 0, k, k ^ 0x7fffffff).astype(np.int64)
    return (i & 0xffffffff).astype(np.uint32).view(np.float32)


def ring_thresholds(orc, P):
    """thr[k], k = 0 .. R: the smallest float32 t whose ring by the ORACLE's formula chain (host libm) is >= k, found by
    bisection over the float keys with x = 1, y = 0, z = t (so that z / sqrt(x^2 + y^2) is t itself); +inf where no float
    reaches ring k.  Independent of the product's ll_ring_thresholds."""
    R = P.n_scans

    def ring_unclamped_ge(t, k):                      # is ring(t) >= k, treating "above the last ring" as >= every k
        ids = orc.scan_ids(np.stack([np.ones_like(t), np.zeros_like(t), t], axis=1), P)
        # rejected points are either below ring 0 or above ring R-1: decide by comparing with a point known to be inside
        below = (ids < 0) & (t < t_mid)
        above = (ids < 0) & ~below
        eff = np.where(below, -1, np.where(above, R, ids))
        return eff >= k

    # a t that certainly lies inside the ring range: the middle of the sensor's elevation interval
    probe = np.tan(np.deg2rad(np.linspace(-60, 60, 2001))).astype(np.float32)
    pid = orc.scan_ids(np.stack([np.ones_like(probe), np.zeros_like(probe), probe], axis=1), P)
    inside = probe[pid >= 0]
    t_mid = inside[len(inside) // 2]
    ks = np.arange(R + 1)
    lo = np.full(R + 1, float_key(np.float32(-np.inf)), np.int64); hi = np.full(R + 1, float_key(np.float32(np.inf)), np.int64)
    while (lo < hi).any():
        mid = (lo + hi) >> 1
        ge = ring_unclamped_ge(key_float(mid), ks)
        hi = np.where(ge, mid, hi); lo = np.where(ge, lo, mid + 1)
    return lo                                          # keys


def spread_scan(rng, n, lo_deg, hi_deg, sweep, nan_frac=0.002):
    """n points with elevations uniform over [lo_deg, hi_deg], ranges log-uniform 1.5 .. 110 m (some inside minimum_range),
    azimuths as one clockwise sweep with jitter (sweep=True) or in random order, a few NaN / inf returns"""
    el = np.deg2rad(rng.uniform(lo_deg, hi_deg, n))
    az = -2 * np.pi * (np.arange(n) + rng.uniform(-0.4, 0.4, n)) / n if sweep else rng.uniform(-np.pi, np.pi, n)
    r = 10 ** rng.uniform(np.log10(1.5), np.log10(110.0), n)
    p = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el), rng.random(n)], axis=1).astype(np.float32)
    bad = rng.random(n) < nan_frac
    p[bad, rng.integers(0, 3, int(bad.sum()))] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), int(bad.sum()))
    return p


def wrap_scan(rng, s0, last_gap, per_boundary=1900, window=2.5e-6):
    """A 16-ring scan whose first point has ori = s0 and whose azimuths crowd around every constant the wrap / halfPassed
    logic compares against (scanRegistration.cpp:177-205), before AND after the half-way flip."""
    two_pi = 2 * np.pi
    e0 = s0 + two_pi - last_gap                                   # ori of the last point + 2 pi, before the 3 pi / pi fix (:115-126)
    consts = [s0 - np.pi / 2, s0 + 3 * np.pi / 2, s0 + np.pi, s0 - np.pi, e0 - 3 * np.pi / 2, e0 + np.pi / 2, e0 - two_pi, s0]
    fold = lambda a: (a + np.pi) % two_pi - np.pi                 # raw ori lives in (-pi, pi]
    groups = []
    for c in consts:
        for img in (c, c - two_pi, c + two_pi):
            groups.append(fold(img) + rng.uniform(-window, window, per_boundary))
    near = np.concatenate(groups)
    fill = rng.uniform(-np.pi, np.pi, 3000)

    def block():
        a = np.concatenate([near, fill]); return a[rng.permutation(len(a))]

    flip = fold(s0 + np.pi + 0.5)                                 # clearly past the half: sets halfPassed for what follows
    ori = np.concatenate([[fold(s0)], block(), [flip], block(), [fold(e0)]])
    ring = rng.integers(0, 16, len(ori))
    el = np.deg2rad(-15.0 + 2.0 * ring)
    r = rng.uniform(6.0, 30.0, len(ori))
    az = -ori                                                     # ori = -atan2(y, x)
    return np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el), np.zeros(len(ori))], axis=1).astype(np.float32)
