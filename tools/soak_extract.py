"""ad-hoc soak: many random irregular scans (the generator of test_random_irregular_scans_stress, other seeds and sizes) through the batch path
(k_organize + ring kernel) against the oracle: laserCloud, labels, four feature clouds bit-exact."""
import os, sys, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lightloam_amd  # noqa
from lightloam_amd import api
from oracle import orc
import test_gpu_parity as T
from conftest import assert_bit_equal
api.load_library(); orc.build()
SORT_SEEN, SORT_BAD = [0], [0]


def sort_check(ctx):
    """-DLL_SORT_CHECK builds count, per context, the adjacent pairs of every ring's sorted records and those out of order"""
    if os.environ.get("LL_SORT_CHECK"):
        import ctypes as C
        buf = (C.c_ulonglong * 16)(); ctx._ck(ctx.lib.ll_debug_counters(ctx.h, buf, 0))
        SORT_SEEN[0] += buf[10]; SORT_BAD[0] += buf[9]

bad = 0; checked = 0
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0           # soak_extract.py <seeds> [first seed]
for seed in range(FIRST, FIRST + (int(sys.argv[1]) if len(sys.argv) > 1 else 6)):
    rng = np.random.default_rng(777 + seed)
    scans = []
    for s in range(400):
        rings = []
        for k in range(16):
            n = int(rng.choice([0, 3, 9, 11, 12, 17, 30, 47, 64, 65, 129, 250, 400, 700], p=[.04, .04, .06, .06, .06, .1, .12, .12, .1, .1, .07, .05, .05, .03]))
            if n == 0: continue
            base = rng.uniform(0.6, 30.0)
            r = base * (1.0 + rng.uniform(0.0005, 0.01) * np.cumsum(rng.standard_normal(n)))
            r = np.where(rng.random(n) < rng.uniform(0.0, 0.3), r * rng.uniform(1.2, 2.0), r)
            if rng.random() < 0.3: r = np.round(r * 8) / 8
            ring = T._vlp16_ring(-15 + 2 * k, n, np.abs(r) + 0.35, phase=rng.random())
            if rng.random() < 0.2 and n > 4: ring[rng.integers(0, n, 2)] = np.nan
            if rng.random() < 0.2 and n > 6:
                j = rng.integers(1, n - 1); ring[j] = ring[j - 1]
            rings.append(ring)
        if not rings: rings.append(T._vlp16_ring(1, 40, 5.0))
        scans.append(T._ring_scan(rings))
    P = orc.params(16, minimum_range=0.3)
    ctx = api.Context(api.default_params(16, batch=len(scans), max_points=max(map(len, scans)) + 8, minimum_range=0.3))
    for k, sc in enumerate(scans): ctx.upload_scan(k, sc)
    ctx.extract(0, len(scans))
    for k, sc in enumerate(scans):
        ref = orc.extract(sc, P); info = ctx.scan_info(k)
        if ref["rc"] != 0:
            assert info.status != 0; continue
        assert info.status == 0, (seed, k, info.status)
        cloud, ss, se = ctx.cloud(k)
        assert_bit_equal(cloud, ref["cloud"], f"seed {seed} scan {k} cloud")
        lab = ctx.labels(k); n = len(lab)
        assert n == len(ref["label"]) and (lab[5:n - 5].astype(np.int32) == ref["label"][5:n - 5]).all(), f"seed {seed} scan {k} labels"
        f = ctx.features(k)
        for name in ("sharp", "less_sharp", "flat", "less_flat"):
            assert_bit_equal(f[name], ref[name], f"seed {seed} scan {k} {name}")
        checked += 1
    sort_check(ctx); ctx.close()
    print("seed", seed, "ok, checked so far", checked, flush=True)
print("soak passed:", checked, "scans")
if os.environ.get("LL_SORT_CHECK"):      # the library is a -DLL_SORT_CHECK build (LIGHTLOAM_HIP_LIB): the voxel sort's order asserted on every ring
    print("sort check: %d adjacent sorted pairs looked at, %d out of (voxel, input order) order" % (SORT_SEEN[0], SORT_BAD[0]))
    assert SORT_SEEN[0] > 0 and SORT_BAD[0] == 0
