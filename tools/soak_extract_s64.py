"""ad-hoc soak: synthetic 64-ring scans of several generator settings (ring-major / azimuth-major, azimuth jitter, dropped returns, NaNs)
at many poses through the batch path (k_organize + ring kernel, > 64 slots) and the whole hot path against the oracle."""
import os, sys, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lightloam_amd  # noqa
from lightloam_amd import api, synth
from oracle import orc
from conftest import assert_bit_equal
api.load_library(); orc.build()
SORT_SEEN, SORT_BAD = [0], [0]


def sort_check(ctx):
    """-DLL_SORT_CHECK builds count, per context, the adjacent pairs of every ring's sorted records and those out of order"""
    if os.environ.get("LL_SORT_CHECK"):
        import ctypes as C
        buf = (C.c_ulonglong * 16)(); ctx._ck(ctx.lib.ll_debug_counters(ctx.h, buf, 0))
        SORT_SEEN[0] += buf[10]; SORT_BAD[0] += buf[9]

N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
total = 0
for name, kw in (("ringmajor", {}), ("azmajor_jitter_nan", dict(order=1, az_jitter_deg=0.4, drop_prob=0.03, emit_nan=1)), ("ringmajor_jitter_drop", dict(az_jitter_deg=0.7, drop_prob=0.1))):
    cfg = synth.default_cfg(64, **kw)
    scans = [synth.scan(cfg, 100 + 3 * k) for k in range(N)]
    P = orc.params(64)
    ctx = api.Context(api.default_params(64, batch=N, max_points=max(map(len, scans)) + 7))
    for k, s in enumerate(scans): ctx.upload_scan(k, s)
    ctx.extract(0, N)
    refs = [orc.extract(s, P) for s in scans]
    for k in range(N):
        assert ctx.scan_info(k).status == 0 and refs[k]["rc"] == 0
        cloud, ss, se = ctx.cloud(k)
        assert_bit_equal(cloud, refs[k]["cloud"], f"{name} {k} cloud")
        f = ctx.features(k)
        for nm in ("sharp", "less_sharp", "flat", "less_flat"):
            assert_bit_equal(f[nm], refs[k][nm], f"{name} {k} {nm}")
    # association of every scan against its predecessor at a fixed guess: index tuples exact
    q = np.array([0.001, -0.002, 0.004, 1.0]); q /= np.linalg.norm(q); t = np.array([0.8, 0.02, -0.01]); pose = np.concatenate([q, t])
    ctx.set_target_from_slot(0)
    ctx.associate(1, N - 1, pose); ctx.synchronize()
    orc.set_nn_mode(1)
    for k in range(1, N, 5):
        es, ea, eb = orc.associate_corner(q, t, refs[k]["sharp"], refs[k - 1]["less_sharp"])
        ps, pa, pb, pc = orc.associate_plane(q, t, refs[k]["flat"], refs[k - 1]["less_flat"])
        ges, gea, geb = ctx.edge_corr(k) if hasattr(ctx, "edge_corr") else (None, None, None)
        ctx.vote(k, 1, True)
        ges, gea, geb = ctx.edge_corr(k); gps, gpa, gpb, gpc = ctx.plane_corr(k)
        assert (ges == es).all() and (gea == ea).all() and (geb == eb).all(), (name, k, "edge")
        assert (gps == ps).all() and (gpa == pa).all() and (gpb == pb).all() and (gpc == pc).all(), (name, k, "plane")
    orc.set_nn_mode(0)
    sort_check(ctx); ctx.close(); total += N
    print(name, "ok", flush=True)
print("soak passed:", total, "scans")
if os.environ.get("LL_SORT_CHECK"):      # the library is a -DLL_SORT_CHECK build (LIGHTLOAM_HIP_LIB): the voxel sort's order asserted on every ring
    print("sort check: %d adjacent sorted pairs looked at, %d out of (voxel, input order) order" % (SORT_SEEN[0], SORT_BAD[0]))
    assert SORT_SEEN[0] > 0 and SORT_BAD[0] == 0
