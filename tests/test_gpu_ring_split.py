"""Round 4 split the ring stage into k_ring_pick + k_ring_features<split>; the single kernel of rounds 1-3 stays in the library behind
LIGHTLOAM_RING_SPLIT=0 as the A/B reference.  Both must write the same bytes: laserCloud labels, curvature, the four feature clouds and
their counts, on regular scans, on the HDL-64E table (ring capacity 4608: every tier of both pipelines) and through both organise paths."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _extract(api, scans, rings, split, **prm):
    os.environ["LIGHTLOAM_RING_SPLIT"] = "1" if split else "0"
    try:
        ctx = api.Context(api.default_params(rings, batch=len(scans), max_points=max(map(len, scans)) + 7, write_curvature=1, **prm))
    finally:
        os.environ.pop("LIGHTLOAM_RING_SPLIT", None)
    for k, s in enumerate(scans):
        ctx.upload_scan(k, s)
    ctx.extract(0, len(scans))
    out = []
    for k in range(len(scans)):
        info = ctx.scan_info(k)
        lab, curv = ctx.labels(k, curvature=True)
        f = ctx.features(k)
        out.append((info.status, info.n, lab.tobytes(), curv[5:len(curv) - 5].tobytes()) + tuple(f[n].tobytes() for n in ("sharp", "less_sharp", "flat", "less_flat")))
    ctx.close()
    return out


@pytest.mark.parametrize("org", ["tiles", "walk"])
@pytest.mark.parametrize("shape", ["S64", "S16_jitter", "hdl64"])
def test_fused_and_split_ring_pipelines_write_the_same_bytes(api, synth, shape, org):
    from conftest import set_org_path
    import scangen
    set_org_path(org)
    try:
        if shape == "hdl64":
            scans, rings, prm = [scangen.hdl64_scan(k, order="kitti") for k in range(3)], 64, dict(max_ring_points=4608)
        elif shape == "S64":
            cfg = synth.default_cfg(64); scans, rings, prm = [synth.scan(cfg, k) for k in range(3)], 64, {}
        else:
            cfg = synth.default_cfg(16, az_jitter_deg=0.4, drop_prob=0.05); scans, rings, prm = [synth.scan(cfg, k) for k in range(3)], 16, {}
        a = _extract(api, scans, rings, True, **prm)
        b = _extract(api, scans, rings, False, **prm)
    finally:
        set_org_path("tiles")
    for k, (x, y) in enumerate(zip(a, b)):
        assert x[0] == 0 and x[1] > 1000
        for i, name in enumerate(("status", "n", "labels", "curvature", "sharp", "less_sharp", "flat", "less_flat")):
            assert x[i] == y[i], f"{shape} scan {k}: {name} differs between the split and the fused ring pipeline"
