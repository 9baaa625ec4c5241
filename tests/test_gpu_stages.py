"""ll_debug_launch_stage (timing probes: tools/experiments/overlap_probe.py, launch_fixed_cost.py) runs exactly the launches of ll_hot_path_batch,
one stage per call: the seven stages in order leave what the one call leaves -- clouds, correspondences, votes, normal equations, byte for byte."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu


def _state(ctx, n):
    out = []
    for k in range(n):
        f = ctx.features(k)
        H, g, cost = ctx.normal_equations_result(k)
        out.append((f, ctx.edge_corr(k), ctx.plane_corr(k), ctx.vote_result(k), H, g, cost))
    return out


def test_stage_launches_equal_the_hot_path(api, synth):
    cfg = synth.default_cfg(64)
    scans = [synth.scan(cfg, 20 + 2 * k) for k in range(6)]
    pose = np.array([0.0, 0.0, 0.001, 1.0, 0.3, 0.01, 0.0]); pose[:4] /= np.linalg.norm(pose[:4])
    res = []
    for staged in (False, True):
        ctx = api.Context(api.default_params(64, batch=6, max_points=max(map(len, scans))))
        ctx.lib.ll_debug_launch_stage.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        try:
            ctx.upload_scan(5, scans[5]); ctx.extract(5, 1); ctx.set_target_from_slot(5)
            for k in range(5):
                ctx.upload_scan(k, scans[k])
            ctx.set_pose_guess(0, 5, pose)
            if not staged:
                ctx.hot_path(0, 5, None, vote=True)
            else:
                ctx.hot_path(0, 5, None, vote=True)            # poses, carry binding
                for k in range(5):
                    ctx.upload_scan(k, scans[(k + 1) % 5])     # scramble every slot, then restore the inputs: the stages must rebuild it all
                ctx.extract(0, 5)
                for k in range(5):
                    ctx.upload_scan(k, scans[k])
                ctx.set_pose_guess(0, 5, pose)                  # the association starts from the guess, as ll_hot_path_batch(NULL) does
                for st in range(7):
                    ctx._ck(ctx.lib.ll_debug_launch_stage(ctx.h, st, 0, 5))
                assert ctx.lib.ll_debug_launch_stage(ctx.h, 7, 0, 5) != 0
            ctx.synchronize()
            res.append(_state(ctx, 5))
        finally:
            ctx.close()
    for k, (a, b) in enumerate(zip(*res)):
        for nm in ("sharp", "less_sharp", "flat", "less_flat"):
            assert_bit_equal(a[0][nm], b[0][nm], f"slot {k} {nm}")
        for i in (1, 2, 3):
            for x, y in zip(a[i], b[i]):
                assert np.array_equal(x, y), (k, i)
        assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5]) and a[6] == b[6], k
