"""Committed fixtures (tests/golden/*.npz, generator tests/golden/make_golden.py): oracle on CPU, HIP path on GPU.
See the generator's header for what the fixtures are (oracle-produced regression snapshots, not reference output)."""
import glob
import os

import numpy as np
import pytest

from conftest import assert_bit_equal

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def test_fixtures_are_committed():
    assert len(FIXTURES) >= 2


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_golden(path, orc):
    G = np.load(path)
    P = orc.params(int(G["rings"]))
    ex = [orc.extract(G["scan0"], P), orc.extract(G["scan1"], P)]
    for k in (0, 1):
        for key in ("cloud", "sharp", "less_sharp", "flat", "less_flat"):
            assert_bit_equal(ex[k][key], G[f"s{k}_{key}"], f"{key}[{k}]")
        assert (ex[k]["label"] == G[f"s{k}_label"]).all()
        assert (ex[k]["scan_start"] == G[f"s{k}_scan_start"]).all() and (ex[k]["scan_end"] == G[f"s{k}_scan_end"]).all()
    q, t = G["pose"][:4], G["pose"][4:]
    es, ea, eb = orc.associate_corner(q, t, ex[1]["sharp"], ex[0]["less_sharp"])
    ps, pa, pb, pc = orc.associate_plane(q, t, ex[1]["flat"], ex[0]["less_flat"])
    for got, key in ((es, "e_src"), (ea, "e_a"), (eb, "e_b"), (ps, "p_src"), (pa, "p_a"), (pb, "p_b"), (pc, "p_c")):
        assert (got == G[key]).all(), key
    cnt, sidx, sw = orc.vote(ex[1]["flat"][ps], ex[0]["less_flat"][pa])
    assert (cnt == G["v_count"]).all() and set(sidx.tolist()) == set(np.flatnonzero(G["v_sel"]).tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_reproduces_golden(path, api):
    G = np.load(path)
    rings = int(G["rings"])
    ctx = api.Context(api.default_params(rings, batch=2, write_curvature=1, max_points=max(len(G["scan0"]), len(G["scan1"]))))
    ctx.upload_scan(0, G["scan0"]); ctx.upload_scan(1, G["scan1"])
    ctx.extract(0, 2)
    for k in (0, 1):
        cloud, ss, se = ctx.cloud(k)
        assert_bit_equal(cloud, G[f"s{k}_cloud"], f"cloud[{k}]")
        assert (ss == G[f"s{k}_scan_start"]).all() and (se == G[f"s{k}_scan_end"]).all()
        lab, cv = ctx.labels(k, curvature=True)
        n = len(lab)
        assert (lab[5:n - 5] == G[f"s{k}_label"][5:n - 5]).all()
        assert_bit_equal(cv[5:n - 5], G[f"s{k}_curv"][5:n - 5], f"curv[{k}]")
        f = ctx.features(k)
        for key in ("sharp", "less_sharp", "flat", "less_flat"):
            assert_bit_equal(f[key], G[f"s{k}_{key}"], f"{key}[{k}]")
    ctx.set_target_from_slot(0)
    ctx.hot_path(1, 1, G["pose"], vote=True)
    es, ea, eb = ctx.edge_corr(1); ps, pa, pb, pc = ctx.plane_corr(1)
    for got, key in ((es, "e_src"), (ea, "e_a"), (eb, "e_b"), (ps, "p_src"), (pa, "p_a"), (pb, "p_b"), (pc, "p_c")):
        assert len(got) == len(G[key]) and (got == G[key]).all(), key
    cnt, sel, w = ctx.vote_result(1)
    assert (cnt == G["v_count"]).all() and (sel == G["v_sel"]).all() and (w[sel] == G["v_w"][sel]).all()
    H, g, cost = ctx.normal_equations_result(1)
    scale = np.abs(G["H"]).max()
    assert np.abs(H - G["H"]).max() <= 1e-9 * scale and np.abs(g - G["g"]).max() <= 1e-9 * max(1, np.abs(G["g"]).max())
    assert abs(cost - float(G["cost"])) <= 1e-9 * max(1.0, float(G["cost"]))
    assert np.abs(ctx.pose(1) - G["pose_after"]).max() <= 1e-9
    ctx.close()
