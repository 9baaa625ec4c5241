import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lightloam_amd
from lightloam_amd import api, synth
rings=64; nframes=40
cfg = synth.default_cfg(rings)
scans = [synth.scan(cfg, k) for k in range(nframes)]
reg = api.Context(api.default_params(rings, batch=2, max_points=max(map(len, scans))))
odo = api.Context(api.default_params(rings, batch=2, max_points=max(map(len, scans))))
T = {k: [] for k in ("upload_scan","extract","dl_cloud","dl_features","upload_features","odometry_frames","set_target","sync")}
guess = np.array([0, 0, 0, 1.0, 0.9, 0, 0])
def tick(name, fn):
    t0=time.perf_counter(); r=fn(); T[name].append((time.perf_counter()-t0)*1e3); return r
for k, s in enumerate(scans):
    tick("upload_scan", lambda: reg.upload_scan(0, s))
    tick("extract", lambda: (reg.extract(0, 1), reg.synchronize()))
    tick("dl_cloud", lambda: reg.cloud(0))
    f = tick("dl_features", lambda: reg.features(0))
    tick("upload_features", lambda: odo.upload_features(0, f["sharp"], f["less_sharp"], f["flat"], f["less_flat"]))
    if k > 0:
        guess = tick("odometry_frames", lambda: odo.odometry_frames(0, 1, pose0=guess, n_outer=3, first_frame_index=k)[0])
    tick("set_target", lambda: odo.set_target_from_slot(0))
    tick("sync", lambda: odo.synchronize())
print({k: round(float(np.median(v[3:])),4) for k,v in T.items()})
