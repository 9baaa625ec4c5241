"""Time of each stage's launch alone over B = 1024 .. 8192 resident S64 scans: the part that does not scale with B (ramp-up, the tail behind the last
workgroups) is what a longer batch amortises and what a second stream could fill.   gpurun -- 'python tools/experiments/launch_fixed_cost.py'"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from lightloam_amd import api, synth

BMAX = 8192
cfg = synth.default_cfg(64)
base = [synth.scan(cfg, k) for k in range(33)]
mp = max(map(len, base))
rng = np.random.default_rng(7)
ctx = api.Context(api.default_params(64, batch=BMAX + 1, max_points=mp))
ctx.lib.ll_debug_launch_stage.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
ctx.upload_scan(BMAX, base[32]); ctx.extract(BMAX, 1); ctx.set_target_from_slot(BMAX)
order = rng.integers(0, 32, BMAX)
for i in range(BMAX):
    ctx.upload_scan(i, base[order[i]])
pose = np.tile(np.array([0.0, 0.0, 0.0, 1.0, 0.05, 0.0, 0.0]), (BMAX, 1))
ctx.set_pose_guess(0, BMAX, pose)
ctx.hot_path(0, BMAX, None, vote=True); ctx.synchronize()
NAMES = ["organize", "pick", "voxel", "grid", "associate", "vote", "normal_eq"]
def timed(st, B, n=6):
    ctx._ck(ctx.lib.ll_debug_launch_stage(ctx.h, st, 0, B)); ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ctx._ck(ctx.lib.ll_debug_launch_stage(ctx.h, st, 0, B))
    ctx.synchronize(); return (time.perf_counter() - t0) / n * 1e3
Bs = [1024, 2048, 4096, 8192]
print("%-10s" % "stage" + "".join("%9d" % b for b in Bs) + "   fixed ms (fit)   ms per 1024 scans")
for st, nm in enumerate(NAMES):
    t = [timed(st, b) for b in Bs]
    A = np.vstack([np.ones(len(Bs)), np.array(Bs) / 1024.0]).T
    a, b = np.linalg.lstsq(A, np.array(t), rcond=None)[0]
    print("%-10s" % nm + "".join("%9.3f" % x for x in t) + "   %8.3f   %8.3f" % (a, b))
