"""ctypes binding of the C ABI (include/lightloam_hip.h) -- plumbing for tests, smoke and bench.

Fails loudly when the HIP library is missing or no gfx950 device is usable: there is no CPU fallback.
"""
import ctypes as C
import os
import weakref

import numpy as np

from .build import lib_path

POINT = np.dtype([("x", "f4"), ("y", "f4"), ("z", "f4"), ("intensity", "f4")])

LL_OK = 0
STATUS = {0: "LL_OK", -1: "LL_ERR_DEVICE", -2: "LL_ERR_ARG", -3: "LL_ERR_BAD_RINGS", -4: "LL_ERR_CAPACITY",
          -5: "LL_ERR_EMPTY", -6: "LL_ERR_HIP", -7: "LL_ERR_STATE"}

EXPORTS = [
    "ll_default_params", "ll_create", "ll_destroy", "ll_last_error", "ll_abi_version", "ll_stream", "ll_synchronize",
    "ll_upload_scan", "ll_extract_batch", "ll_get_scan_info", "ll_download_cloud", "ll_download_labels",
    "ll_download_features", "ll_set_target", "ll_upload_features", "ll_set_target_from_slot", "ll_associate_batch", "ll_get_pair_info",
    "ll_download_edge_corr", "ll_download_plane_corr", "ll_vote_batch", "ll_download_vote",
    "ll_normal_equations_batch", "ll_download_normal_equations", "ll_gn_step_batch", "ll_download_pose",
    "ll_residual_jacobian", "ll_hot_path_batch", "ll_algorithmic_bytes", "ll_profile_enable", "ll_profile_read", "ll_set_pose_guess", "ll_debug_counters", "ll_vote_host", "ll_debug_calibration_copy", "ll_debug_launch_stage", "ll_debug_exact_math", "ll_upload_scan_async", "ll_upload_scans_async", "ll_upload_scans_async_strided", "ll_stream_record", "ll_stream_wait", "ll_hot_path_chain", "ll_synchronize_copy", "ll_host_alloc", "ll_host_free",
    "ll_lm_default_options", "ll_lm_solve_batch", "ll_odometry_frames", "ll_set_two_stream",
    "ll_map_create", "ll_map_destroy", "ll_map_last_error", "ll_map_set_map", "ll_map_set_scan", "ll_map_associate",
    "ll_map_residual_jacobian", "ll_map_get_counts", "ll_map_get_map_sizes", "ll_map_download_edges", "ll_map_download_planes", "ll_map_normal_equations", "ll_map_optimize",
    "ll_cubemap_create", "ll_cubemap_destroy", "ll_cubemap_last_error", "ll_cubemap_prepare", "ll_cubemap_optimize", "ll_cubemap_update",
    "ll_cubemap_process", "ll_cubemap_process_slot", "ll_cubemap_info", "ll_cubemap_download_cloud", "ll_cubemap_download_cube",
    "ll_map_set_map_ids", "ll_map_knn_partial", "ll_map_associate_merged", "ll_map_solve", "ll_map_set_row_shard", "ll_cubemap_set_shard", "ll_cubemap_map",
    "ll_factor_blocks_set", "ll_factor_blocks_set_s", "ll_factor_blocks_evaluate",
    "ll_map_evaluate_dev", "ll_map_lm_begin_dev", "ll_map_lm_propose_dev", "ll_map_lm_accept_dev", "ll_map_knn_partial_dev",
    "ll_map_associate_merged_dev", "ll_map_solve_dev",
    "ll_voxel_grid", "ll_map_set_pose", "ll_map_get_pose", "ll_map_evaluate", "ll_map_lm_begin", "ll_map_lm_propose", "ll_map_lm_accept",
]


class Params(C.Structure):
    _fields_ = [("n_scans", C.c_int), ("ring_model", C.c_int), ("minimum_range", C.c_float),
                ("lower_bound", C.c_float), ("up_bound", C.c_float), ("max_points", C.c_int),
                ("max_ring_points", C.c_int), ("batch", C.c_int), ("curv_threshold", C.c_float),
                ("gap_sq_threshold", C.c_float), ("leaf_size", C.c_float), ("nn_dist_sq_max", C.c_float),
                ("nearby_scan", C.c_float), ("huber_delta", C.c_float), ("write_curvature", C.c_int),
                ("chunk", C.c_int), ("distortion", C.c_int), ("voxel_sort_ranks", C.c_int), ("input_stride_floats", C.c_int)]


class ScanInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("n_in", C.c_int), ("n", C.c_int), ("n_sharp", C.c_int),
                ("n_less_sharp", C.c_int), ("n_flat", C.c_int), ("n_less_flat", C.c_int), ("max_ring", C.c_int)]


class LmOptions(C.Structure):
    _fields_ = [("max_num_iterations", C.c_int), ("initial_radius", C.c_double), ("max_radius", C.c_double),
                ("min_radius", C.c_double), ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double), ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double), ("jacobi_scaling", C.c_int)]


class PairInfo(C.Structure):
    _fields_ = [("n_edge", C.c_int), ("n_plane", C.c_int), ("n_plane_selected", C.c_int)]


class LightLoamError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


_lib = None


def load_library():
    """dlopen the in-tree HIP library.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        _lib = C.CDLL(path)
        _lib.ll_last_error.restype = C.c_char_p
        _lib.ll_last_error.argtypes = [C.c_void_p]
        _lib.ll_stream.restype = C.c_void_p
        _lib.ll_stream.argtypes = [C.c_void_p]
        _lib.ll_create.argtypes = [C.c_int, C.POINTER(Params), C.POINTER(C.c_void_p)]
        _lib.ll_destroy.argtypes = [C.c_void_p]
        _lib.ll_host_alloc.restype = C.c_void_p
        _lib.ll_host_alloc.argtypes = [C.c_size_t]
        _lib.ll_host_free.argtypes = [C.c_void_p]
    return _lib


class PinnedStaging:
    """`count` scan slots of `stride_points` points each in ONE page-locked area: what an ingest thread fills and
    ll_upload_scans_async_strided copies with two enqueues"""

    def __init__(self, count, stride_points, floats_per_point=4):
        self.lib = load_library()
        self.fpp = int(floats_per_point)             # the context's resident layout (Params.input_stride_floats): 4, or 3 = x, y, z packed
        self.count, self.stride = int(count), int(stride_points) * 4 * self.fpp
        self.ptr = self.lib.ll_host_alloc(self.count * self.stride)
        if not self.ptr:
            raise MemoryError("ll_host_alloc failed")
        self.n = (C.c_int * self.count)()

    def put(self, i, xyz4):
        a = np.ascontiguousarray(np.asarray(xyz4, np.float32)[:, :self.fpp])   # an ingest thread's repack (KITTI .bin records carry a 4th float)
        assert a.ndim == 2 and a.shape[1] == self.fpp and a.nbytes <= self.stride
        C.memmove(self.ptr + i * self.stride, a.ctypes.data, a.nbytes)
        self.n[i] = len(a)

    def close(self):
        if getattr(self, "ptr", None):
            self.lib.ll_host_free(C.c_void_p(self.ptr)); self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedScan:
    """an (n, 4) float32 scan in page-locked host memory (ll_host_alloc): the source of ll_upload_scan_async"""

    def __init__(self, xyz4, floats_per_point=4):
        a = np.ascontiguousarray(np.asarray(xyz4, np.float32)[:, :int(floats_per_point)])
        assert a.ndim == 2 and a.shape[1] == int(floats_per_point)
        self.n = len(a)
        self.lib = load_library()
        self.ptr = self.lib.ll_host_alloc(max(a.nbytes, 16))
        if not self.ptr:
            raise MemoryError("ll_host_alloc failed")
        C.memmove(self.ptr, a.ctypes.data, a.nbytes)

    def close(self):
        if getattr(self, "ptr", None):
            self.lib.ll_host_free(C.c_void_p(self.ptr)); self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_params(n_scans=64, **kw):
    p = Params()
    load_library().ll_default_params(C.byref(p), n_scans)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One ll_ctx: `batch` scan slots resident in HBM on one GPU."""

    def __init__(self, params, device=0):
        self.lib = load_library()
        self.params = params
        h = C.c_void_p()
        rc = self.lib.ll_create(device, C.byref(params), C.byref(h))
        if rc != LL_OK:
            raise LightLoamError(rc, self.lib.ll_last_error(None).decode())
        self.h = h
        self.R = params.n_scans
        self._children = weakref.WeakSet()      # Map / CubeMap objects living on this context: destroyed before it

    def close(self):
        if getattr(self, "h", None):
            for child in list(getattr(self, "_children", ())):   # a map outliving its context would free through a dangling pointer
                try:
                    child.close()
                except Exception:
                    pass
            self.lib.ll_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != LL_OK:
            raise LightLoamError(rc, self.lib.ll_last_error(self.h).decode())

    @property
    def stream(self):
        return self.lib.ll_stream(self.h)

    def synchronize(self):
        self._ck(self.lib.ll_synchronize(self.h))

    # ---- input
    def upload_scan(self, slot, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        n, stride = (0, 4) if xyz.size == 0 else xyz.shape
        self._ck(self.lib.ll_upload_scan(self.h, slot, _ptr(xyz), stride, n))

    def upload_scan_async(self, slot, pinned):
        """enqueue the host -> device copy of a PinnedScan on the copy stream (ll_upload_scan_async)"""
        self._ck(self.lib.ll_upload_scan_async(self.h, slot, C.c_void_p(pinned.ptr), pinned.n))

    def upload_scans_async(self, first, pinned_list):
        """one call for a run of slots (the loop is in C: a Python call per scan would cost more than the copy's enqueue)"""
        k = len(pinned_list)
        ptrs = (C.c_void_p * k)(*[p.ptr for p in pinned_list]); ns = (C.c_int * k)(*[p.n for p in pinned_list])
        self._ck(self.lib.ll_upload_scans_async(self.h, first, k, ptrs, ns))

    def upload_staging_async(self, first, staging, i0=0, count=None):
        """slots first .. from entries i0 .. of a PinnedStaging (ll_upload_scans_async_strided)"""
        count = staging.count - i0 if count is None else count
        n = (C.c_int * count).from_buffer(staging.n, i0 * 4)
        self._ck(self.lib.ll_upload_scans_async_strided(self.h, first, count, C.c_void_p(staging.ptr + i0 * staging.stride), C.c_size_t(staging.stride), n))

    def stream_record(self, stream, ev):
        """mark "everything enqueued so far" on the compute (0) / copy (1) stream as event ev (0 .. 7)"""
        self._ck(self.lib.ll_stream_record(self.h, int(stream), int(ev)))

    def stream_wait(self, stream, ev):
        """what is enqueued on `stream` from now on waits for event ev"""
        self._ck(self.lib.ll_stream_wait(self.h, int(stream), int(ev)))

    def hot_path_chain(self, first, count, vote=True):
        """hot_path continuing a batch: slot `first` is matched against slot first - 1"""
        self._ck(self.lib.ll_hot_path_chain(self.h, first, count, int(bool(vote))))

    def synchronize_copy(self):
        self._ck(self.lib.ll_synchronize_copy(self.h))

    # ---- stages
    def extract(self, first=0, count=1):
        self._ck(self.lib.ll_extract_batch(self.h, first, count))

    def scan_info(self, slot=0):
        info = ScanInfo()
        self._ck(self.lib.ll_get_scan_info(self.h, slot, C.byref(info)))
        return info

    def cloud(self, slot=0):
        info = self.scan_info(slot)
        cloud = np.zeros((max(info.n, 1), 4), np.float32)
        ss = np.zeros(self.R, np.int32); se = np.zeros(self.R, np.int32)
        self._ck(self.lib.ll_download_cloud(self.h, slot, _ptr(cloud), len(cloud), _ptr(ss), _ptr(se)))
        return cloud[:info.n], ss, se

    def labels(self, slot=0, curvature=False):
        info = self.scan_info(slot)
        lab = np.zeros(max(info.n, 1), np.int8)
        cv = np.zeros(max(info.n, 1), np.float32) if curvature else None
        self._ck(self.lib.ll_download_labels(self.h, slot, _ptr(lab), _ptr(cv), len(lab)))
        return (lab[:info.n], cv[:info.n]) if curvature else lab[:info.n]

    def features(self, slot=0):
        info = self.scan_info(slot)
        arr = lambda n: np.zeros((max(n, 1), 4), np.float32)
        sh, ls, fl, lf = arr(info.n_sharp), arr(info.n_less_sharp), arr(info.n_flat), arr(info.n_less_flat)
        self._ck(self.lib.ll_download_features(self.h, slot, _ptr(sh), len(sh), _ptr(ls), len(ls), _ptr(fl), len(fl),
                                               _ptr(lf), len(lf)))
        return dict(sharp=sh[:info.n_sharp], less_sharp=ls[:info.n_less_sharp], flat=fl[:info.n_flat],
                    less_flat=lf[:info.n_less_flat])

    def set_target(self, corner_last, surf_last):
        c = np.ascontiguousarray(corner_last, np.float32).reshape(-1, 4)
        s = np.ascontiguousarray(surf_last, np.float32).reshape(-1, 4)
        self._ck(self.lib.ll_set_target(self.h, _ptr(c), len(c), _ptr(s), len(s)))

    def upload_features(self, slot, sharp, less_sharp, flat, less_flat):
        """the four feature clouds of a scan from host arrays into a slot (what a separate odometry process gets by topic)"""
        a = [np.ascontiguousarray(x, np.float32).reshape(-1, 4) for x in (sharp, less_sharp, flat, less_flat)]
        self._ck(self.lib.ll_upload_features(self.h, slot, _ptr(a[0]), len(a[0]), _ptr(a[1]), len(a[1]), _ptr(a[2]), len(a[2]),
                                             _ptr(a[3]), len(a[3])))

    def set_target_from_slot(self, slot):
        self._ck(self.lib.ll_set_target_from_slot(self.h, slot))

    @staticmethod
    def _poses(pose, count):
        if pose is None:
            return None
        p = np.ascontiguousarray(pose, np.float64).reshape(-1, 7)
        if len(p) == 1 and count > 1:
            p = np.repeat(p, count, axis=0)
        assert len(p) == count
        return np.ascontiguousarray(p)

    def associate(self, first=0, count=1, pose=None):
        p = self._poses(pose, count)
        self._ck(self.lib.ll_associate_batch(self.h, first, count, _ptr(p)))

    def set_pose_guess(self, first, count, pose):
        p = self._poses(pose, count)
        self._ck(self.lib.ll_set_pose_guess(self.h, first, count, _ptr(p)))

    def vote(self, first=0, count=1, enable=True):
        self._ck(self.lib.ll_vote_batch(self.h, first, count, int(bool(enable))))

    def pair_info(self, slot=0):
        info = PairInfo()
        self._ck(self.lib.ll_get_pair_info(self.h, slot, C.byref(info)))
        return info

    def edge_corr(self, slot=0):
        n = self.pair_info(slot).n_edge
        a = [np.zeros(max(n, 1), np.int32) for _ in range(3)]
        self._ck(self.lib.ll_download_edge_corr(self.h, slot, _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), len(a[0])))
        return tuple(x[:n] for x in a)

    def plane_corr(self, slot=0):
        n = self.pair_info(slot).n_plane
        a = [np.zeros(max(n, 1), np.int32) for _ in range(4)]
        self._ck(self.lib.ll_download_plane_corr(self.h, slot, _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), len(a[0])))
        return tuple(x[:n] for x in a)

    def vote_result(self, slot=0):
        n = self.pair_info(slot).n_plane
        cnt = np.zeros(max(n, 1), np.int32); sel = np.zeros(max(n, 1), np.uint8); w = np.zeros(max(n, 1), np.float32)
        self._ck(self.lib.ll_download_vote(self.h, slot, _ptr(cnt), _ptr(sel), _ptr(w), len(cnt)))
        return cnt[:n], sel[:n].astype(bool), w[:n]

    def normal_equations(self, first=0, count=1, pose=None):
        p = self._poses(pose, count)
        self._ck(self.lib.ll_normal_equations_batch(self.h, first, count, _ptr(p)))

    def normal_equations_result(self, slot=0):
        H = np.zeros((6, 6)); g = np.zeros(6); cost = C.c_double(0)
        self._ck(self.lib.ll_download_normal_equations(self.h, slot, _ptr(H), _ptr(g), C.byref(cost)))
        return H, g, cost.value

    def gn_step(self, first=0, count=1):
        self._ck(self.lib.ll_gn_step_batch(self.h, first, count))

    def pose(self, slot=0):
        p = np.zeros(7)
        self._ck(self.lib.ll_download_pose(self.h, slot, _ptr(p)))
        return p

    def residual_jacobian(self, slot=0, pose=None):
        pi = self.pair_info(slot)
        rows = 3 * pi.n_edge + pi.n_plane_selected
        r = np.zeros(max(rows, 1)); Jq = np.zeros((max(rows, 1), 4)); Jt = np.zeros((max(rows, 1), 3))
        p = None if pose is None else np.ascontiguousarray(pose, np.float64)
        self._ck(self.lib.ll_residual_jacobian(self.h, slot, _ptr(p), _ptr(r), _ptr(Jq), _ptr(Jt), len(r)))
        return r[:rows], Jq[:rows], Jt[:rows]

    def lm_solve(self, first=0, count=1, opt=None):
        """ceres::Solve (LM, max 4 iterations) on the slot's current residual blocks, from the slot's current pose."""
        self._ck(self.lib.ll_lm_solve_batch(self.h, first, count, None if opt is None else C.byref(opt)))

    def odometry_frames(self, first, count, pose0=None, n_outer=3, first_frame_index=1, opt=None):
        """laserOdometry's frame loop over consecutive slots; returns the (count, 7) relative poses (q_last_curr, t_last_curr)."""
        out = np.zeros((count, 7))
        p0 = None if pose0 is None else np.ascontiguousarray(pose0, np.float64)
        self._ck(self.lib.ll_odometry_frames(self.h, first, count, _ptr(p0), n_outer, first_frame_index,
                                             None if opt is None else C.byref(opt), _ptr(out)))
        return out

    def hot_path(self, first=0, count=1, pose=None, vote=True):
        p = self._poses(pose, count)
        self._ck(self.lib.ll_hot_path_batch(self.h, first, count, _ptr(p), int(bool(vote))))

    def set_two_stream(self, on=True):
        """association stage of hot_path on two streams (default) or kernel after kernel (ll_set_two_stream)"""
        self._ck(self.lib.ll_set_two_stream(self.h, int(bool(on))))

    def profile_enable(self, on=True):
        self._ck(self.lib.ll_profile_enable(self.h, int(bool(on))))

    def profile_read(self, reset=True):
        """{kernel name: (total_ms, launches)} measured with HIP events on the ctx stream."""
        n = C.c_int(16)
        names = (C.c_char_p * 16)(); ms = (C.c_double * 16)(); launches = (C.c_int * 16)()
        self._ck(self.lib.ll_profile_read(self.h, C.byref(n), names, ms, launches, int(bool(reset))))
        return {names[i].decode(): (ms[i], launches[i]) for i in range(n.value)}

    def voxel_grid(self, points, leaf):
        """pcl::VoxelGrid on a whole cloud (N x 4 float32) -> filtered cloud"""
        pts = np.ascontiguousarray(points, np.float32)
        out = np.zeros((max(len(pts), 1), 4), np.float32); n = C.c_int(0)
        self._ck(self.lib.ll_voxel_grid(self.h, _ptr(pts), len(pts), C.c_float(leaf), _ptr(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def factor_blocks_set(self, edge9=None, plane13=None, pnorm7=None):
        """lidarFactor.hpp blocks of one problem: edge [n, 9], plane [n, 13], plane-norm [n, 7] (f64)."""
        e = np.ascontiguousarray(edge9 if edge9 is not None else np.zeros((0, 9)), np.float64).reshape(-1, 9)
        p = np.ascontiguousarray(plane13 if plane13 is not None else np.zeros((0, 13)), np.float64).reshape(-1, 13)
        n = np.ascontiguousarray(pnorm7 if pnorm7 is not None else np.zeros((0, 7)), np.float64).reshape(-1, 7)
        self._fb_rows = 3 * len(e) + len(p) + len(n)
        self._ck(self.lib.ll_factor_blocks_set(self.h, len(e), _ptr(e), len(p), _ptr(p), len(n), _ptr(n)))

    def factor_blocks_set_s(self, edge_s=None, plane_s=None):
        """the functors' s_ of every edge / plane block (DISTORTION 1); None = all ones"""
        e = None if edge_s is None else np.ascontiguousarray(edge_s, np.float64)
        p = None if plane_s is None else np.ascontiguousarray(plane_s, np.float64)
        self._ck(self.lib.ll_factor_blocks_set_s(self.h, _ptr(e), _ptr(p)))

    def factor_blocks_evaluate(self, q, t):
        q = np.ascontiguousarray(q, np.float64); t = np.ascontiguousarray(t, np.float64)
        rows = self._fb_rows
        r = np.zeros(max(rows, 1)); Jq = np.zeros((max(rows, 1), 4)); Jt = np.zeros((max(rows, 1), 3))
        self._ck(self.lib.ll_factor_blocks_evaluate(self.h, _ptr(q), _ptr(t), _ptr(r), _ptr(Jq), _ptr(Jt), len(r)))
        return r[:rows], Jq[:rows], Jt[:rows]

    def exact_math(self, op, a, b=None, c=None):
        """the DEVICE's bit-exact libm restatements over float32 arrays (ll_debug_exact_math); ops 5 / 6 return int32 ring ids"""
        a = np.ascontiguousarray(a, np.float32)
        b = None if b is None else np.ascontiguousarray(b, np.float32)
        c = None if c is None else np.ascontiguousarray(c, np.float32)
        out = np.zeros(len(a), np.int32 if op >= 5 else np.float32)
        self._ck(self.lib.ll_debug_exact_math(self.h, int(op), _ptr(a), _ptr(b), _ptr(c), len(a), _ptr(out)))
        return out

    def algorithmic_bytes(self, first=0, count=1):
        b = [C.c_double(0) for _ in range(4)]
        self._ck(self.lib.ll_algorithmic_bytes(self.h, first, count, *[C.byref(x) for x in b]))
        return dict(ext=b[0].value, assoc=b[1].value, vote=b[2].value, rj=b[3].value)


class Map:
    """One ll_map: laserMapping's scan-to-submap optimisation (laserMapping.cpp:1822-2095) on the device of `ctx`."""

    def __init__(self, ctx, max_map_corner, max_map_surf, max_scan_corner, max_scan_surf, _borrowed=None):
        self.ctx = ctx; self.lib = ctx.lib
        self.lib.ll_map_last_error.restype = C.c_char_p
        self.lib.ll_map_last_error.argtypes = [C.c_void_p]
        self.lib.ll_map_destroy.argtypes = [C.c_void_p]
        self._owned = _borrowed is None
        if _borrowed is not None:                      # the inner map of a CubeMap: the cube map destroys it
            self.h = _borrowed
            return
        self.h = C.c_void_p()
        rc = self.lib.ll_map_create(ctx.h, int(max_map_corner), int(max_map_surf), int(max_scan_corner), int(max_scan_surf), C.byref(self.h))
        if rc != LL_OK:
            raise LightLoamError(rc, ctx.lib.ll_last_error(ctx.h).decode())
        ctx._children.add(self)

    def close(self):
        if getattr(self, "h", None):
            if self._owned:
                self.lib.ll_map_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != LL_OK:
            raise LightLoamError(rc, self.lib.ll_map_last_error(self.h).decode())

    @staticmethod
    def _pts(a):
        a = np.ascontiguousarray(a, np.float32)
        assert a.ndim == 2 and a.shape[1] == 4
        return a

    def set_map(self, corner_from_map, surf_from_map):
        c, s_ = self._pts(corner_from_map), self._pts(surf_from_map)
        self._ck(self.lib.ll_map_set_map(self.h, _ptr(c), len(c), _ptr(s_), len(s_)))

    def set_scan(self, corner_stack, surf_stack):
        c, s_ = self._pts(corner_stack), self._pts(surf_stack)
        self._n_stack = (len(c), len(s_))
        self._ck(self.lib.ll_map_set_scan(self.h, _ptr(c), len(c), _ptr(s_), len(s_)))

    def associate(self, pose_w=None):
        p = None if pose_w is None else np.ascontiguousarray(pose_w, np.float64)
        self._ck(self.lib.ll_map_associate(self.h, _ptr(p)))

    def counts(self):
        ne, npl = C.c_int(0), C.c_int(0)
        self._ck(self.lib.ll_map_get_counts(self.h, C.byref(ne), C.byref(npl)))
        return ne.value, npl.value

    def map_sizes(self):
        """(laserCloudCornerFromMap, laserCloudSurfFromMap) sizes: what laserMapping.cpp:1822 tests before it optimises."""
        nc, ns = C.c_int(0), C.c_int(0)
        self._ck(self.lib.ll_map_get_map_sizes(self.h, C.byref(nc), C.byref(ns)))
        return nc.value, ns.value

    def edges(self):
        ne, _ = self.counts()
        src = np.zeros(max(ne, 1), np.int32); a = np.zeros((max(ne, 1), 3)); b = np.zeros((max(ne, 1), 3))
        self._ck(self.lib.ll_map_download_edges(self.h, _ptr(src), _ptr(a), _ptr(b), len(src)))
        return src[:ne], a[:ne], b[:ne]

    def planes(self):
        _, npl = self.counts()
        src = np.zeros(max(npl, 1), np.int32); n = np.zeros((max(npl, 1), 3)); d = np.zeros(max(npl, 1))
        self._ck(self.lib.ll_map_download_planes(self.h, _ptr(src), _ptr(n), _ptr(d), len(src)))
        return src[:npl], n[:npl], d[:npl]

    def normal_equations(self, pose_w=None):
        p = None if pose_w is None else np.ascontiguousarray(pose_w, np.float64)
        H = np.zeros((6, 6)); g = np.zeros(6); cost = C.c_double(0)
        self._ck(self.lib.ll_map_normal_equations(self.h, _ptr(p), _ptr(H), _ptr(g), C.byref(cost)))
        return H, g, cost.value

    def residual_jacobian(self, pose_w=None):
        ne, npl = self.counts()
        rows = 3 * ne + npl
        p = None if pose_w is None else np.ascontiguousarray(pose_w, np.float64)
        r = np.zeros(max(rows, 1)); Jq = np.zeros((max(rows, 1), 4)); Jt = np.zeros((max(rows, 1), 3))
        self._ck(self.lib.ll_map_residual_jacobian(self.h, _ptr(p), _ptr(r), _ptr(Jq), _ptr(Jt), len(r)))
        return r[:rows], Jq[:rows], Jt[:rows]

    def optimize(self, pose_w, n_outer=2, opt=None):
        p = np.ascontiguousarray(pose_w, np.float64).copy()
        ran = C.c_int(0)
        self._ck(self.lib.ll_map_optimize(self.h, _ptr(p), n_outer, None if opt is None else C.byref(opt), C.byref(ran)))
        return p, bool(ran.value)

    # ---- tile-parallel search: this map holds a shard of the cubes, the candidates are all-gathered
    def set_map_ids(self, corner_gid, surf_gid):
        c = None if corner_gid is None else np.ascontiguousarray(corner_gid, np.int32)
        s_ = None if surf_gid is None else np.ascontiguousarray(surf_gid, np.int32)
        self._ck(self.lib.ll_map_set_map_ids(self.h, _ptr(c), _ptr(s_)))

    def knn_partial(self, pose_w=None, n_stack=None):
        """-> corner_nn [n, 5, 4] f32 (x, y, z, d2), corner_id [n, 5] i32, surf_nn, surf_id of this rank's points."""
        nc, ns = n_stack if n_stack is not None else self._n_stack
        p = None if pose_w is None else np.ascontiguousarray(pose_w, np.float64)
        cn = np.zeros((nc, 5, 4), np.float32); ci = np.zeros((nc, 5), np.int32)
        sn = np.zeros((ns, 5, 4), np.float32); si = np.zeros((ns, 5), np.int32)
        self._ck(self.lib.ll_map_knn_partial(self.h, _ptr(p), _ptr(cn), _ptr(ci), _ptr(sn), _ptr(si)))
        return cn, ci, sn, si

    def associate_merged(self, corner_nn, corner_id, surf_nn, surf_id, pose_w=None):
        """The candidates of all parts, arrays [parts, n, 5(, 4)] -> the residual blocks of the whole map."""
        cn = np.ascontiguousarray(corner_nn, np.float32); ci = np.ascontiguousarray(corner_id, np.int32)
        sn = np.ascontiguousarray(surf_nn, np.float32); si = np.ascontiguousarray(surf_id, np.int32)
        assert cn.ndim == 4 and sn.ndim == 4 and cn.shape[0] == sn.shape[0] == ci.shape[0] == si.shape[0]
        p = None if pose_w is None else np.ascontiguousarray(pose_w, np.float64)
        self._ck(self.lib.ll_map_associate_merged(self.h, _ptr(p), int(cn.shape[0]), _ptr(cn), _ptr(ci), _ptr(sn), _ptr(si)))

    def set_row_shard(self, rank, world):
        """evaluate() / normal_equations() sum the residual blocks i with i % world == rank ((0, 1): all of them)."""
        self._ck(self.lib.ll_map_set_row_shard(self.h, int(rank), int(world)))

    def solve(self, pose_w, opt=None):
        p = np.ascontiguousarray(pose_w, np.float64).copy()
        self._ck(self.lib.ll_map_solve(self.h, _ptr(p), None if opt is None else C.byref(opt)))
        return p

    # ---- row-parallel stepping: the caller sums `evaluate()` over ranks between the LM stages
    def set_pose(self, pose_w):
        p = np.ascontiguousarray(pose_w, np.float64)
        self._ck(self.lib.ll_map_set_pose(self.h, _ptr(p)))

    def pose(self):
        p = np.zeros(7)
        self._ck(self.lib.ll_map_get_pose(self.h, _ptr(p)))
        return p

    def evaluate(self):
        v = np.zeros(44)
        self._ck(self.lib.ll_map_evaluate(self.h, _ptr(v)))
        return v

    def lm_begin(self, neq44, opt=None):
        v = np.ascontiguousarray(neq44, np.float64)
        self._ck(self.lib.ll_map_lm_begin(self.h, _ptr(v), None if opt is None else C.byref(opt)))

    def lm_propose(self, opt=None):
        self._ck(self.lib.ll_map_lm_propose(self.h, None if opt is None else C.byref(opt)))

    def lm_accept(self, neq44, opt=None):
        v = np.ascontiguousarray(neq44, np.float64)
        self._ck(self.lib.ll_map_lm_accept(self.h, _ptr(v), None if opt is None else C.byref(opt)))

    # ---- device-resident variants: raw DEVICE pointers (ints), enqueue only -- for RCCL collectives on ll_stream(ctx)
    def evaluate_dev(self, neq44_ptr):
        self._ck(self.lib.ll_map_evaluate_dev(self.h, C.c_void_p(neq44_ptr)))

    def lm_begin_dev(self, neq44_ptr, opt=None):
        self._ck(self.lib.ll_map_lm_begin_dev(self.h, C.c_void_p(neq44_ptr), None if opt is None else C.byref(opt)))

    def lm_propose_dev(self, opt=None):
        self._ck(self.lib.ll_map_lm_propose_dev(self.h, None if opt is None else C.byref(opt)))

    def lm_accept_dev(self, neq44_ptr, opt=None):
        self._ck(self.lib.ll_map_lm_accept_dev(self.h, C.c_void_p(neq44_ptr), None if opt is None else C.byref(opt)))

    def knn_partial_dev(self, cn_ptr, ci_ptr, sn_ptr, si_ptr):
        self._ck(self.lib.ll_map_knn_partial_dev(self.h, C.c_void_p(cn_ptr), C.c_void_p(ci_ptr), C.c_void_p(sn_ptr), C.c_void_p(si_ptr)))

    def associate_merged_dev(self, n_parts, cn_ptr, ci_ptr, sn_ptr, si_ptr):
        self._ck(self.lib.ll_map_associate_merged_dev(self.h, int(n_parts), C.c_void_p(cn_ptr), C.c_void_p(ci_ptr), C.c_void_p(sn_ptr), C.c_void_p(si_ptr)))

    def solve_dev(self, opt=None):
        self._ck(self.lib.ll_map_solve_dev(self.h, None if opt is None else C.byref(opt)))


class CubeMap:
    """One ll_cubemap: laserMapping's cube map + per-frame body (laserMapping.cpp:1584-2165) on the device of `ctx`."""

    def __init__(self, ctx, max_scan_corner, max_scan_surf, pool_points=1 << 20, line_res=0.4, plane_res=0.8):
        self.ctx = ctx; self.lib = ctx.lib
        self.lib.ll_cubemap_last_error.restype = C.c_char_p
        self.lib.ll_cubemap_last_error.argtypes = [C.c_void_p]
        self.lib.ll_cubemap_destroy.argtypes = [C.c_void_p]
        self.h = C.c_void_p()
        rc = self.lib.ll_cubemap_create(ctx.h, C.c_float(line_res), C.c_float(plane_res), int(max_scan_corner), int(max_scan_surf),
                                        int(pool_points), C.byref(self.h))
        if rc != LL_OK:
            raise LightLoamError(rc, ctx.lib.ll_last_error(ctx.h).decode())
        ctx._children.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ll_cubemap_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != LL_OK:
            raise LightLoamError(rc, self.lib.ll_cubemap_last_error(self.h).decode())

    def set_shard(self, rank, world):
        """Keep only the cubes of `rank` out of `world` (tile-parallel mapping); before the first scan."""
        self._ck(self.lib.ll_cubemap_set_shard(self.h, int(rank), int(world)))

    def map(self):
        """The inner Map (borrowed): knn_partial / associate_merged / solve / edges / planes on the gathered clouds."""
        self.lib.ll_cubemap_map.restype = C.c_void_p
        self.lib.ll_cubemap_map.argtypes = [C.c_void_p]
        return Map(self.ctx, 0, 0, 0, 0, _borrowed=C.c_void_p(self.lib.ll_cubemap_map(self.h)))

    def prepare(self, t_w, corner_last, surf_last):
        t = np.ascontiguousarray(t_w, np.float64)
        c = np.ascontiguousarray(corner_last, np.float32); s_ = np.ascontiguousarray(surf_last, np.float32)
        self._ck(self.lib.ll_cubemap_prepare(self.h, _ptr(t), _ptr(c), len(c), _ptr(s_), len(s_)))

    def optimize(self, pose_w, n_outer=2, opt=None):
        p = np.ascontiguousarray(pose_w, np.float64).copy(); ran = C.c_int(0)
        self._ck(self.lib.ll_cubemap_optimize(self.h, _ptr(p), n_outer, None if opt is None else C.byref(opt), C.byref(ran)))
        return p, bool(ran.value)

    def update(self, pose_w):
        p = np.ascontiguousarray(pose_w, np.float64)
        self._ck(self.lib.ll_cubemap_update(self.h, _ptr(p)))

    def process(self, pose_w, corner_last, surf_last):
        p = np.ascontiguousarray(pose_w, np.float64).copy(); ran = C.c_int(0)
        c = np.ascontiguousarray(corner_last, np.float32); s_ = np.ascontiguousarray(surf_last, np.float32)
        self._ck(self.lib.ll_cubemap_process(self.h, _ptr(p), _ptr(c), len(c), _ptr(s_), len(s_), C.byref(ran)))
        return p, bool(ran.value)

    def process_slot(self, pose_w, slot):
        p = np.ascontiguousarray(pose_w, np.float64).copy(); ran = C.c_int(0)
        self._ck(self.lib.ll_cubemap_process_slot(self.h, _ptr(p), int(slot), C.byref(ran)))
        return p, bool(ran.value)

    def info(self):
        cen = (C.c_int * 3)(); cnt = (C.c_int * 4)()
        self._ck(self.lib.ll_cubemap_info(self.h, cen, cnt))
        return tuple(cen), tuple(cnt)

    def cloud(self, which):
        _, cnt = self.info()
        out = np.zeros((max(cnt[which], 1), 4), np.float32); n = C.c_int(0)
        self._ck(self.lib.ll_cubemap_download_cloud(self.h, which, _ptr(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def cube(self, surf, index, cap=1 << 18):
        out = np.zeros((cap, 4), np.float32); n = C.c_int(0)
        self._ck(self.lib.ll_cubemap_download_cube(self.h, int(surf), int(index), _ptr(out), len(out), C.byref(n)))
        return out[:n.value].copy()
