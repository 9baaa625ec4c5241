"""Which kernels of the hot path run faster side by side -- two contexts, two streams, one device -- than one after the other?
  gpurun -- 'python tools/experiments/overlap_probe.py [batch]'
Every stage alone over `batch` resident scans, then every pair (stage X on context 1 beside stage Y on context 2): wall time of the pair against the sum
of the two alone.  (ll_debug_launch_stage: 0 organise, 1 pick, 2 voxel filter + lists, 3 grid tables, 4 association.)"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from lightloam_amd import api
from lightloam_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = synth.default_cfg(64)
base = [synth.scan(cfg, k) for k in range(17)]
mp = max(map(len, base))
pose = np.array([0.0, 0.0, 0.0, 1.0, 0.05, 0.0, 0.0])
ctxs = []
for c in range(2):
    ctx = api.Context(api.default_params(64, batch=B + 1, max_points=mp))
    ctx.lib.ll_debug_launch_stage.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    ctx.upload_scan(B, base[16]); ctx.extract(B, 1); ctx.set_target_from_slot(B)
    for i in range(B):
        ctx.upload_scan(i, base[i % 16])
    ctx.set_pose_guess(0, B, np.tile(pose, (B, 1)))
    ctx.hot_path(0, B, None, vote=True); ctx.synchronize()
    ctxs.append(ctx)
c1, c2 = ctxs
NAMES = ["organize", "pick", "voxel", "grid", "associate"]
def launch(c, st): c._ck(c.lib.ll_debug_launch_stage(c.h, st, 0, B))
def sync(): c1.synchronize(); c2.synchronize()
def timed(fn, n=4):
    fn(); sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    sync(); return (time.perf_counter() - t0) / n * 1e3
alone = [timed(lambda s=s: launch(c1, s)) for s in range(5)]
print("batch %d, alone (ms): " % B + "  ".join("%s %.2f" % (n, t) for n, t in zip(NAMES, alone)))
for x in range(5):
    for y in range(x, 5):
        t = timed(lambda: (launch(c1, x), launch(c2, y)))
        print("%-9s beside %-9s %6.2f ms = %5.1f %% of %6.2f" % (NAMES[x], NAMES[y], t, 100 * t / (alone[x] + alone[y]), alone[x] + alone[y]))
