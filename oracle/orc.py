"""ctypes binding of the CPU oracle (oracle/ll_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (light-loam_amd/) never imports this module.  PARITY UNPINNED -- see ll_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libll_oracle.so")

POINT = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("intensity", "f4")])


class Params(C.Structure):
    _fields_ = [("n_scans", C.c_int), ("ring_model", C.c_int), ("minimum_range", C.c_double),
                ("lower_bound", C.c_float), ("up_bound", C.c_float)]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("ll_oracle.c", "ll_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        for name in ("orc_organize", "orc_pick", "orc_voxel_grid", "orc_extract", "orc_associate_corner",
                     "orc_associate_plane", "orc_gn_solve"):
            getattr(_lib, name).restype = C.c_int
    return _lib


def params(n_scans=64, minimum_range=None, lower_bound=-24.9, up_bound=2.0, ring_model=0):
    if minimum_range is None:
        minimum_range = 5.0 if n_scans == 64 else 0.3     # launch/aloam_velodyne_HDL_64.launch:8, VLP_16.launch
    return Params(n_scans, ring_model, float(minimum_range), lower_bound, up_bound)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f4(points):
    """(n,4) float32 view of a point array (structured POINT or plain)."""
    a = np.ascontiguousarray(points)
    if a.dtype == POINT:
        a = a.view(np.float32).reshape(-1, 4)
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4)


def organize(xyz, P):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n_in, stride = xyz.shape
    cloud = np.zeros((max(n_in, 1), 4), np.float32)
    n = C.c_int(0)
    ss = np.zeros(P.n_scans, np.int32)
    se = np.zeros(P.n_scans, np.int32)
    rc = lib().orc_organize(_p(xyz), stride, n_in, C.byref(P), _p(cloud), C.byref(n), _p(ss), _p(se))
    return rc, cloud[:n.value].copy(), ss, se


def scan_ids(xyz, P):
    """ring id of every point by scanRegistration.cpp:139-168 (-1 = rejected), no filtering"""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    ids = np.zeros(len(xyz), np.int32)
    lib().orc_scan_ids(_p(xyz), xyz.shape[1], len(xyz), C.byref(P), _p(ids))
    return ids


def libm(op, a, b=None, c=None):
    """the host libm over float32 arrays: op 0 atanf(a), 1 atan2f(a, b), 2 (float)((double)a / M_PI), 3 a / sqrtf(b*b + c*c), 4 expf(a)"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(a if b is None else b, np.float32); c = np.ascontiguousarray(a if c is None else c, np.float32)
    out = np.zeros(len(a), np.float32)
    lib().orc_libm_batch(int(op), _p(a), _p(b), _p(c), len(a), _p(out))
    return out


def curvature(cloud):
    cloud = _f4(cloud)
    curv = np.zeros(len(cloud), np.float32)
    lib().orc_curvature(_p(cloud), len(cloud), _p(curv))
    return curv


def voxel_grid(points, leaf=0.2):
    points = _f4(points)
    out = np.zeros((max(len(points), 1), 4), np.float32)
    n = C.c_int(0)
    lib().orc_voxel_grid(_p(points), len(points), C.c_float(leaf), _p(out), C.byref(n))
    return out[:n.value].copy()


def extract(xyz, P):
    """Whole laserCloudHandler.  Returns dict of arrays."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n_in, stride = xyz.shape
    cap = max(n_in, 1)
    cloud = np.zeros((cap, 4), np.float32)
    curv = np.zeros(cap, np.float32)
    label = np.zeros(cap, np.int32)
    ss = np.zeros(P.n_scans, np.int32)
    se = np.zeros(P.n_scans, np.int32)
    sharp = np.zeros((cap, 4), np.float32)
    less_sharp = np.zeros((cap, 4), np.float32)
    flat = np.zeros((cap, 4), np.float32)
    less_flat = np.zeros((cap, 4), np.float32)
    n = C.c_int(0); ns = C.c_int(0); nls = C.c_int(0); nf = C.c_int(0); nlf = C.c_int(0)
    rc = lib().orc_extract(_p(xyz), stride, n_in, C.byref(P), _p(cloud), C.byref(n), _p(ss), _p(se), _p(curv), _p(label),
                           _p(sharp), C.byref(ns), _p(less_sharp), C.byref(nls), _p(flat), C.byref(nf),
                           _p(less_flat), C.byref(nlf))
    return dict(rc=rc, cloud=cloud[:n.value].copy(), curv=curv[:n.value].copy(), label=label[:n.value].copy(),
                scan_start=ss, scan_end=se, sharp=sharp[:ns.value].copy(), less_sharp=less_sharp[:nls.value].copy(),
                flat=flat[:nf.value].copy(), less_flat=less_flat[:nlf.value].copy())


def set_distortion(on):
    """DISTORTION of laserOdometry.cpp:23 for TransformToStart and the odometry factors of normal_equations / lm_solve /
    odometry_frame: 0 = the reference's build (default), 1 = per-point s = (intensity - int(intensity)) / SCAN_PERIOD."""
    lib().orc_set_distortion(int(bool(on)))


def point_s(point):
    lib().orc_point_s.restype = C.c_double
    p = np.ascontiguousarray(point, dtype=np.float32).reshape(4)
    return float(lib().orc_point_s(_p(p)))


def set_nn_mode(mode):
    lib().orc_set_nn_mode(int(mode))


def transform_to_start(q, t, pts):
    pts = _f4(pts)
    out = np.zeros_like(pts)
    q = np.ascontiguousarray(q, np.float64); t = np.ascontiguousarray(t, np.float64)
    f = lib().orc_transform_to_start
    for i in range(len(pts)):
        f(_p(q), _p(t), C.c_void_p(pts.ctypes.data + 16 * i), C.c_void_p(out.ctypes.data + 16 * i))
    return out


def associate_corner(q, t, sharp, corner_last):
    sharp = _f4(sharp); corner_last = _f4(corner_last)
    q = np.ascontiguousarray(q, np.float64); t = np.ascontiguousarray(t, np.float64)
    cap = max(len(sharp), 1)
    src = np.zeros(cap, np.int32); a = np.zeros(cap, np.int32); b = np.zeros(cap, np.int32)
    n = C.c_int(0)
    lib().orc_associate_corner(_p(q), _p(t), _p(sharp), len(sharp), _p(corner_last), len(corner_last),
                               _p(src), _p(a), _p(b), C.byref(n))
    k = n.value
    return src[:k].copy(), a[:k].copy(), b[:k].copy()


def associate_plane(q, t, flat, surf_last):
    flat = _f4(flat); surf_last = _f4(surf_last)
    q = np.ascontiguousarray(q, np.float64); t = np.ascontiguousarray(t, np.float64)
    cap = max(len(flat), 1)
    src = np.zeros(cap, np.int32); a = np.zeros(cap, np.int32); b = np.zeros(cap, np.int32); c = np.zeros(cap, np.int32)
    n = C.c_int(0)
    lib().orc_associate_plane(_p(q), _p(t), _p(flat), len(flat), _p(surf_last), len(surf_last),
                              _p(src), _p(a), _p(b), _p(c), C.byref(n))
    k = n.value
    return src[:k].copy(), a[:k].copy(), b[:k].copy(), c[:k].copy()


def vote(src, tgt, corner_case=False):
    src = _f4(src); tgt = _f4(tgt)
    n = len(src)
    counts = np.zeros(max(n, 1), np.int32); idx = np.zeros(max(n, 1), np.int32); w = np.zeros(max(n, 1), np.float32)
    ns = C.c_int(0)
    lib().orc_vote(_p(src), _p(tgt), n, int(bool(corner_case)), _p(counts), _p(idx), _p(w), C.byref(ns))
    return counts[:n].copy(), idx[:ns.value].copy(), w[:ns.value].copy()


def _d(a, n):
    a = np.ascontiguousarray(a, np.float64)
    assert a.size == n
    return a


def edge_factor(q, t, cp, a, b, s=1.0):
    r = np.zeros(3); Jq = np.zeros((3, 4)); Jt = np.zeros((3, 3))
    q = _d(q, 4); t = _d(t, 3); cp = _d(cp, 3); a = _d(a, 3); b = _d(b, 3)
    lib().orc_edge_factor(_p(q), _p(t), _p(cp), _p(a), _p(b), C.c_double(s), _p(r), _p(Jq), _p(Jt))
    return r, Jq, Jt


def plane_factor_modify(q, t, cp, j, l, m, s=1.0, weight=1.0):
    r = np.zeros(1); Jq = np.zeros((1, 4)); Jt = np.zeros((1, 3))
    q = _d(q, 4); t = _d(t, 3); cp = _d(cp, 3); j = _d(j, 3); l = _d(l, 3); m = _d(m, 3)
    lib().orc_plane_factor_modify(_p(q), _p(t), _p(cp), _p(j), _p(l), _p(m), C.c_double(s), C.c_double(weight),
                                  _p(r), _p(Jq), _p(Jt))
    return r, Jq, Jt


def plane_norm_factor(q, t, cp, n, d):
    r = np.zeros(1); Jq = np.zeros((1, 4)); Jt = np.zeros((1, 3))
    q = _d(q, 4); t = _d(t, 3); cp = _d(cp, 3); n = _d(n, 3)
    lib().orc_plane_norm_factor(_p(q), _p(t), _p(cp), _p(n), C.c_double(d), _p(r), _p(Jq), _p(Jt))
    return r, Jq, Jt


def quat_plus_jacobian(q):
    P = np.zeros((4, 3)); q = _d(q, 4)
    lib().orc_quat_plus_jacobian(_p(q), _p(P))
    return P


def quat_plus(q, d):
    out = np.zeros(4); q = _d(q, 4); d = _d(d, 3)
    lib().orc_quat_plus(_p(q), _p(d), _p(out))
    return out


def normal_equations(q, t, sharp, e_src, corner_last, e_a, e_b, flat, p_src, surf_last, p_a, p_b, p_c, p_w,
                     huber_delta=0.1):
    q = _d(q, 4); t = _d(t, 3)
    sharp = _f4(sharp); corner_last = _f4(corner_last); flat = _f4(flat); surf_last = _f4(surf_last)
    i32 = lambda a: np.ascontiguousarray(a, np.int32)
    e_src, e_a, e_b, p_src, p_a, p_b, p_c = map(i32, (e_src, e_a, e_b, p_src, p_a, p_b, p_c))
    pw = None if p_w is None else np.ascontiguousarray(p_w, np.float32)
    H = np.zeros((6, 6)); g = np.zeros(6); cost = C.c_double(0)
    lib().orc_normal_equations(_p(q), _p(t), _p(sharp), _p(e_src), _p(corner_last), _p(e_a), _p(e_b), len(e_src),
                               _p(flat), _p(p_src), _p(surf_last), _p(p_a), _p(p_b), _p(p_c),
                               None if pw is None else _p(pw), len(p_src), C.c_double(huber_delta),
                               _p(H), _p(g), C.byref(cost))
    return H, g, cost.value


def gn_solve(H, g):
    H = np.ascontiguousarray(H, np.float64); g = _d(g, 6); d = np.zeros(6)
    rc = lib().orc_gn_solve(_p(H), _p(g), _p(d))
    return rc, d


def pose_update(q, t, delta):
    q = _d(q, 4).copy(); t = _d(t, 3).copy(); delta = _d(delta, 6)
    lib().orc_pose_update(_p(q), _p(t), _p(delta))
    return q, t


class LmOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("initial_radius", C.c_double), ("max_radius", C.c_double),
                ("min_radius", C.c_double), ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double), ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("jacobi_scaling", C.c_int)]


def lm_options():
    o = LmOptions()
    lib().orc_lm_default(C.byref(o))
    return o


def lm_solve(q, t, sharp, e_src, corner_last, e_a, e_b, flat, p_src, surf_last, p_a, p_b, p_c, p_w, huber_delta=0.1, opt=None):
    q = _d(q, 4).copy(); t = _d(t, 3).copy()
    sharp = _f4(sharp); corner_last = _f4(corner_last); flat = _f4(flat); surf_last = _f4(surf_last)
    i32 = lambda a: np.ascontiguousarray(a, np.int32)
    e_src, e_a, e_b, p_src, p_a, p_b, p_c = map(i32, (e_src, e_a, e_b, p_src, p_a, p_b, p_c))
    pw = None if p_w is None else np.ascontiguousarray(p_w, np.float32)
    opt = opt or lm_options()
    summary = np.zeros(4)
    lib().orc_lm_solve(_p(q), _p(t), _p(sharp), _p(e_src), _p(corner_last), _p(e_a), _p(e_b), len(e_src),
                       _p(flat), _p(p_src), _p(surf_last), _p(p_a), _p(p_b), _p(p_c), None if pw is None else _p(pw), len(p_src),
                       C.c_double(huber_delta), C.byref(opt), _p(summary))
    return q, t, summary


def odometry_frame(q, t, cur, last, vote, n_outer=3, huber_delta=0.1, opt=None):
    """cur / last: dicts from extract() (cur: sharp, flat; last: less_sharp, less_flat).  Returns updated (q, t)."""
    q = _d(q, 4).copy(); t = _d(t, 3).copy()
    sharp = _f4(cur["sharp"]); flat = _f4(cur["flat"]); cl = _f4(last["less_sharp"]); sl = _f4(last["less_flat"])
    opt = opt or lm_options()
    lib().orc_odometry_frame(_p(q), _p(t), _p(sharp), len(sharp), _p(flat), len(flat), _p(cl), len(cl), _p(sl), len(sl),
                             int(bool(vote)), n_outer, C.c_double(huber_delta), C.byref(opt))
    return q, t


# ---- f2: laserMapping scan-to-submap (laserMapping.cpp:1822-2095)
def sym_eig3(A):
    A = np.ascontiguousarray(A, np.float64).reshape(9); w = np.zeros(3); V = np.zeros((3, 3))
    lib().orc_sym_eig3(_p(A), _p(w), _p(V))
    return w, V


def qr_solve_5x3(A, b):
    A = np.ascontiguousarray(A, np.float64).reshape(15); b = _d(b, 5); x = np.zeros(3)
    lib().orc_qr_solve_5x3(_p(A), _p(b), _p(x))
    return x


def point_associate_to_map(q, t, pts):
    pts = _f4(pts); out = np.zeros_like(pts); q = _d(q, 4); t = _d(t, 3)
    f = lib().orc_point_associate_to_map
    for i in range(len(pts)):
        f(_p(q), _p(t), C.c_void_p(pts.ctypes.data + 16 * i), C.c_void_p(out.ctypes.data + 16 * i))
    return out


def map_associate(q, t, corner_stack, corner_map, surf_stack, surf_map):
    """-> (e_src, e_a[n,3], e_b[n,3], p_src, p_n[n,3], p_d[n])"""
    q = _d(q, 4); t = _d(t, 3)
    cs, cm, ss, sm = _f4(corner_stack), _f4(corner_map), _f4(surf_stack), _f4(surf_map)
    e_src = np.zeros(len(cs) + 1, np.int32); e_a = np.zeros((len(cs) + 1, 3)); e_b = np.zeros((len(cs) + 1, 3))
    p_src = np.zeros(len(ss) + 1, np.int32); p_n = np.zeros((len(ss) + 1, 3)); p_d = np.zeros(len(ss) + 1)
    ne = C.c_int(0); npl = C.c_int(0)
    lib().orc_map_associate(_p(q), _p(t), _p(cs), len(cs), _p(cm), len(cm), _p(ss), len(ss), _p(sm), len(sm),
                            _p(e_src), _p(e_a), _p(e_b), C.byref(ne), _p(p_src), _p(p_n), _p(p_d), C.byref(npl))
    return (e_src[:ne.value].copy(), e_a[:ne.value].copy(), e_b[:ne.value].copy(),
            p_src[:npl.value].copy(), p_n[:npl.value].copy(), p_d[:npl.value].copy())


def map_normal_equations(q, t, corner_stack, e_src, e_a, e_b, surf_stack, p_src, p_n, p_d, huber_delta=0.1):
    q = _d(q, 4); t = _d(t, 3); cs, ss = _f4(corner_stack), _f4(surf_stack)
    e_src = np.ascontiguousarray(e_src, np.int32); p_src = np.ascontiguousarray(p_src, np.int32)
    e_a = np.ascontiguousarray(e_a, np.float64); e_b = np.ascontiguousarray(e_b, np.float64)
    p_n = np.ascontiguousarray(p_n, np.float64); p_d = np.ascontiguousarray(p_d, np.float64)
    H = np.zeros((6, 6)); g = np.zeros(6); cost = C.c_double(0)
    lib().orc_map_normal_equations(_p(q), _p(t), _p(cs), _p(e_src), _p(e_a), _p(e_b), len(e_src),
                                   _p(ss), _p(p_src), _p(p_n), _p(p_d), len(p_src), C.c_double(huber_delta), _p(H), _p(g), C.byref(cost))
    return H, g, cost.value


def map_optimize(q, t, corner_stack, corner_map, surf_stack, surf_map, n_outer=2, huber_delta=0.1, opt=None):
    q = _d(q, 4).copy(); t = _d(t, 3).copy()
    cs, cm, ss, sm = _f4(corner_stack), _f4(corner_map), _f4(surf_stack), _f4(surf_map)
    opt = opt or lm_options()
    ran = lib().orc_map_optimize(_p(q), _p(t), _p(cs), len(cs), _p(cm), len(cm), _p(ss), len(ss), _p(sm), len(sm),
                                 n_outer, C.c_double(huber_delta), C.byref(opt))
    return q, t, bool(ran)


class CubeMap:
    """laserMapping's cube map (laserMapping.cpp:1584-1821, :2101-2165), restated"""

    def __init__(self, line_res=0.4, plane_res=0.8):
        L = lib()
        L.orc_cubemap_create.restype = C.c_void_p
        L.orc_cubemap_create.argtypes = [C.c_float, C.c_float]
        for f in ("orc_cubemap_destroy", "orc_cubemap_prepare", "orc_cubemap_update", "orc_cubemap_get", "orc_cubemap_cube", "orc_cubemap_center"):
            getattr(L, f).restype = None
        L.orc_cubemap_destroy.argtypes = [C.c_void_p]
        L.orc_cubemap_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_cubemap_optimize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]
        L.orc_cubemap_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_cubemap_get.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_cubemap_cube.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_cubemap_center.argtypes = [C.c_void_p, C.c_void_p]
        self.L = L
        self.h = L.orc_cubemap_create(line_res, plane_res)

    def close(self):
        if self.h:
            self.L.orc_cubemap_destroy(self.h); self.h = None

    def prepare(self, t_w, corner_last, surf_last):
        t = _d(t_w, 3); c, s_ = _f4(corner_last), _f4(surf_last)
        self.L.orc_cubemap_prepare(self.h, _p(t), _p(c), len(c), _p(s_), len(s_))

    def optimize(self, q, t, n_outer=2, huber_delta=0.1, opt=None):
        q = _d(q, 4).copy(); t = _d(t, 3).copy(); opt = opt or lm_options()
        ran = self.L.orc_cubemap_optimize(self.h, _p(q), _p(t), n_outer, huber_delta, C.byref(opt))
        return q, t, bool(ran)

    def update(self, q, t):
        q = _d(q, 4); t = _d(t, 3)
        self.L.orc_cubemap_update(self.h, _p(q), _p(t))

    def _view(self, fn, *args):
        p = C.c_void_p(); n = C.c_int(0)
        fn(self.h, *args, C.byref(p), C.byref(n))
        if n.value == 0:
            return np.zeros((0, 4), np.float32)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n.value, 4)).copy()

    def cloud(self, which):
        """0 corner from map, 1 surf from map, 2 corner stack, 3 surf stack"""
        return self._view(self.L.orc_cubemap_get, which)

    def cube(self, surf, index):
        return self._view(self.L.orc_cubemap_cube, int(surf), int(index))

    def center(self):
        c = (C.c_int * 3)()
        self.L.orc_cubemap_center(self.h, c)
        return tuple(c)
