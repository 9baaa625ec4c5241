"""ctypes binding of the host-side synthetic scan generator (host/ll_synth.c)."""
import ctypes as C
import numpy as np
from .build import build_synth


class SynthCfg(C.Structure):
    _fields_ = [("rings", C.c_int), ("azimuths", C.c_int), ("elev_lo_deg", C.c_double), ("elev_hi_deg", C.c_double),
                ("range_sigma", C.c_double), ("max_range", C.c_double), ("az_jitter_deg", C.c_double),
                ("order", C.c_int), ("seed", C.c_uint32), ("speed", C.c_double), ("yaw_rate", C.c_double),
                ("period", C.c_double), ("drop_prob", C.c_double), ("emit_nan", C.c_int)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_synth())
        _lib.ll_synth_default.argtypes = [C.POINTER(SynthCfg), C.c_int]
        _lib.ll_synth_scan.argtypes = [C.POINTER(SynthCfg), C.c_int, C.c_void_p]
        _lib.ll_synth_scan.restype = C.c_int
        _lib.ll_synth_pose.argtypes = [C.POINTER(SynthCfg), C.c_int, C.POINTER(C.c_double)]
    return _lib


def default_cfg(rings=64, **kw):
    cfg = SynthCfg()
    _load().ll_synth_default(C.byref(cfg), rings)
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def scan(cfg, k):
    """Synthetic scan k as an (n, 4) float32 array (x, y, z, reflectance), sensor frame."""
    buf = np.empty((cfg.rings * cfg.azimuths, 4), dtype=np.float32)
    n = _load().ll_synth_scan(C.byref(cfg), k, buf.ctypes.data)
    return buf[:n].copy()


def pose(cfg, k):
    p = (C.c_double * 3)()
    _load().ll_synth_pose(C.byref(cfg), k, p)
    return np.array(p[:])
