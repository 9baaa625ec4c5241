#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

WHAT THESE ARE: regression snapshots of inputs and expected outputs for the hot path, produced by OUR CPU oracle
(oracle/ll_oracle.c) on deterministic synthetic scans.  They are NOT outputs of the reference: BrenYi/Light-LOAM ships
no tests or fixtures and none of its translation units builds in this image (ROS1 / PCL / Eigen / Ceres are absent and
no stand-in headers are written for them), so no reference-generated vector can exist here ("parity unpinned").
They travel to the GPU box (which has no /root/reference and must not need the generator) and pin both the oracle
(tests/test_golden.py, CPU) and the HIP path (-m gpu) against silent drift.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lightloam_amd  # noqa: E402,F401
from lightloam_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (rings, azimuths, synth overrides, pose guess)
    "vlp16_450": (16, 450, dict(), [0, 0, 0, 1, 0.9, 0.0, 0.0]),
    "hdl64_256_azmajor": (64, 256, dict(order=1, az_jitter_deg=0.4, drop_prob=0.02, emit_nan=1), [0.001, -0.002, 0.004, 1, 0.8, 0.02, -0.01]),
}


def build(name):
    rings, az, kw, pose = CASES[name]
    cfg = synth.default_cfg(rings, azimuths=az, **kw)
    P = orc.params(rings)
    scans = [synth.scan(cfg, k) for k in range(2)]
    ex = [orc.extract(s, P) for s in scans]
    q = np.array(pose[:4], float); q /= np.linalg.norm(q); t = np.array(pose[4:], float)
    es, ea, eb = orc.associate_corner(q, t, ex[1]["sharp"], ex[0]["less_sharp"])
    ps, pa, pb, pc = orc.associate_plane(q, t, ex[1]["flat"], ex[0]["less_flat"])
    cnt, sidx, sw = orc.vote(ex[1]["flat"][ps], ex[0]["less_flat"][pa])
    order = np.sort(sidx); wmap = np.ones(len(ps), np.float32); wmap[sidx] = sw
    H, g, cost = orc.normal_equations(q, t, ex[1]["sharp"], es, ex[0]["less_sharp"], ea, eb, ex[1]["flat"], ps[order],
                                      ex[0]["less_flat"], pa[order], pb[order], pc[order], wmap[order], 0.1)
    rc, d = orc.gn_solve(H, g)
    q1, t1 = orc.pose_update(q, t, d)
    out = dict(rings=rings, pose=np.concatenate([q, t]), scan0=scans[0], scan1=scans[1])
    for k in (0, 1):
        for key in ("cloud", "label", "scan_start", "scan_end", "sharp", "less_sharp", "flat", "less_flat"):
            out[f"s{k}_{key}"] = ex[k][key]
        out[f"s{k}_curv"] = ex[k]["curv"]
    sel = np.zeros(len(ps), bool); sel[sidx] = True
    out.update(e_src=es, e_a=ea, e_b=eb, p_src=ps, p_a=pa, p_b=pb, p_c=pc, v_count=cnt, v_sel=sel, v_w=wmap,
               H=H, g=g, cost=cost, pose_after=np.concatenate([q1, t1]))
    return out


if __name__ == "__main__":
    for name in CASES:
        data = build(name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB, n_in {len(data['scan1'])}, edges {len(data['e_src'])}, planes {len(data['p_src'])}")
