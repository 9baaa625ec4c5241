"""HDL-64E / KITTI-shaped workload generator (BASELINE config 3 stand-in; KITTI itself is not in the image).

The C generator (host/ll_synth.c) puts every ring at the exact bin centre of scanRegistration.cpp:162.  Real sensors do
not: the HDL-64E's lasers sit at two different angular spacings, are mounted at different heights and fire with per-laser
azimuth offsets, so the elevation the reference computes from (x, y, z) falls anywhere inside a ring's bin -- sometimes
across its edge, and some bins of the linear 64-ring model hold two lasers (ring capacity 4608).  numpy only; used by
tests/scangen.py and by `bench.py --workload hdl64`.  Workload plumbing, not part of the product path.
"""
import numpy as np

# Velodyne HDL-64E S2 vertical angles, the table behind /root/reference/paramter_configuration_for_benchmarks.txt:21-29
# ("angle >= -8.83: scanID = int((2 - angle) * 3 + 0.5), else 32 + int((-8.83 - angle) * 2 + 0.5)"): an upper block of 32
# lasers 1/3 deg apart from +2 deg, a lower block of 32 lasers 1/2 deg apart from -8.83 deg.
HDL64_ELEV_DEG = np.concatenate([2.0 - np.arange(32) / 3.0, -8.83 - np.arange(32) / 2.0])


def _street_scene(rng):
    """axis-aligned boxes (cx, cy, hx, hy, z0, z1) and vertical cylinders (cx, cy, r, z0, z1) of a street canyon"""
    boxes = [(-20.0 + 9.0 * i + rng.uniform(-1, 1), s * rng.uniform(3.0, 6.0), 2.1, 0.9, -1.73, -0.2) for i in range(8) for s in (-1, 1)]
    poles = [(-24.0 + 7.0 * i + rng.uniform(-1, 1), s * 8.0, 0.15, -1.73, 6.0) for i in range(9) for s in (-1, 1)]
    return np.array(boxes), np.array(poles)


def _raycast(org, d, boxes, poles, half_len=45.0, half_wid=11.0, ground=-1.73, max_range=80.0):
    """first hit of rays org + t d (N x 3 each, world frame) with the ground, the two building fronts y = +-half_wid, the
    street's ends x = +-half_len, the boxes and the poles; inf = no return"""
    with np.errstate(divide="ignore", invalid="ignore"):
        best = np.full(len(d), np.inf)
        t = (ground - org[:, 2]) / d[:, 2]
        best = np.where((d[:, 2] < -1e-9) & (t > 0), np.minimum(best, t), best)
        for axis, lim in ((1, half_wid), (0, half_len)):
            for sgn in (-1.0, 1.0):
                t = (sgn * lim - org[:, axis]) / d[:, axis]
                z = org[:, 2] + t * d[:, 2]
                ok = (t > 1e-6) & (z < 14.0) & (d[:, axis] * sgn > 1e-9)
                best = np.where(ok, np.minimum(best, t), best)
        for cx, cy, hx, hy, z0, z1 in boxes:
            lo = np.array([cx - hx, cy - hy, z0]); hi = np.array([cx + hx, cy + hy, z1])
            ta = (lo - org) / d; tb = (hi - org) / d
            tn = np.nanmax(np.minimum(ta, tb), axis=1); tf = np.nanmin(np.maximum(ta, tb), axis=1)
            ok = (tn <= tf) & (tn > 1e-6)
            best = np.where(ok, np.minimum(best, tn), best)
        for cx, cy, r, z0, z1 in poles:
            ox, oy = org[:, 0] - cx, org[:, 1] - cy
            a = d[:, 0] ** 2 + d[:, 1] ** 2; b = ox * d[:, 0] + oy * d[:, 1]; c = ox * ox + oy * oy - r * r
            disc = b * b - a * c
            t = (-b - np.sqrt(np.maximum(disc, 0.0))) / a
            z = org[:, 2] + t * d[:, 2]
            ok = (disc >= 0) & (a > 1e-12) & (t > 1e-6) & (z >= z0) & (z <= z1)
            best = np.where(ok, np.minimum(best, t), best)
    return np.where(best <= max_range, best, np.inf)


def hdl64_scan(k, order="kitti", n_az=1900, seed=64, step=1.0, mount_spread=0.10, rot_spread_deg=2.5, range_sigma=0.02):
    """Scan k of an HDL-64E driving down the street (pose: x = k * step, yaw = 0.004 k).  Every laser has its true elevation,
    its own mounting height (+-mount_spread m) and a rotational offset (alternating +-rot_spread_deg), so the apparent
    elevation atan(z / sqrt(x^2 + y^2)) of a return depends on its range.
    order "kitti": laser by laser from the top, each a full revolution (how a KITTI .bin lists the points);
    order "firing": column by column, the 64 lasers of a column in firing order (upper / lower blocks interleaved).
    Returns an (n, 4) float32 array (x, y, z, reflectance) in the sensor frame, ~120 k points."""
    rng = np.random.default_rng(seed)
    boxes, poles = _street_scene(rng)
    el = np.deg2rad(HDL64_ELEV_DEG)
    z0 = np.linspace(mount_spread, -mount_spread, 64)
    rot = np.deg2rad(rot_spread_deg) * np.where(np.arange(64) % 2 == 0, 1.0, -1.0) * (0.4 + 0.6 * rng.random(64))
    yaw = 0.004 * k
    pos = np.array([k * step, 0.15 * np.sin(0.7 * k), 0.0])
    rk = np.random.default_rng(seed * 1000003 + k)
    j = np.arange(n_az)
    az = -2.0 * np.pi * (j[None, :] + rk.uniform(-0.2, 0.2, (64, n_az))) / n_az + rot[:, None]       # clockwise sweep, jittered
    ce, se = np.cos(el)[:, None], np.sin(el)[:, None]
    ds = np.stack([ce * np.cos(az), ce * np.sin(az), np.broadcast_to(se, az.shape)], axis=-1)         # sensor frame
    cy, sy = np.cos(yaw), np.sin(yaw)
    dw = np.stack([cy * ds[..., 0] - sy * ds[..., 1], sy * ds[..., 0] + cy * ds[..., 1], ds[..., 2]], axis=-1)
    org_s = np.zeros((64, n_az, 3)); org_s[..., 2] = z0[:, None]
    org_w = org_s + pos
    t = _raycast(org_w.reshape(-1, 3), dw.reshape(-1, 3), boxes, poles).reshape(64, n_az)
    t = t + range_sigma * rk.standard_normal(t.shape)
    p = org_s + t[..., None] * ds
    keep = np.isfinite(t) & (rk.random(t.shape) > 0.02)                                              # 2 % dropped returns
    refl = rk.random(t.shape)
    pts = np.concatenate([p, refl[..., None]], axis=-1).astype(np.float32)
    if order == "kitti":
        sel = [pts[r][keep[r]] for r in range(64)]
    elif order == "firing":
        fire = np.stack([np.arange(32), 32 + np.arange(32)], axis=1).reshape(-1)                     # upper / lower block alternate
        cols = pts[fire].transpose(1, 0, 2); kc = keep[fire].T
        sel = [cols[c][kc[c]] for c in range(n_az)]
    else:
        raise ValueError(order)
    return np.ascontiguousarray(np.concatenate(sel), dtype=np.float32)


def pose(k, step=1.0):
    """(x, y, yaw) of scan k in the world frame: the trajectory hdl64_scan drives."""
    return np.array([k * step, 0.15 * np.sin(0.7 * k), 0.004 * k])
