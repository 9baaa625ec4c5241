"""GPU parity of what round 5 changed behind the C ABI (results must not show any of it):

* the less-flat cloud of an extracted slot lies in ring rows on the device (no hand-over between the rings of a scan); the ABI
  hands out the reference's contiguous cloud and contiguous indices -- through the slot itself, through the carry copy, through a
  mix of uploaded (contiguous) and extracted (ring-strided) slots;
* the association FIXES a slot's target: stage calls over sub-ranges afterwards refer to the same clouds.
(The capacity tiers of long rings, now run over work lists, keep their parity cases in tests/test_gpu_parity.py.)"""
import numpy as np
import pytest

from conftest import assert_bit_equal

pytestmark = pytest.mark.gpu

POSE = np.array([0.001, -0.002, 0.004, 1.0, 0.8, 0.02, -0.01])
POSE[:4] /= np.linalg.norm(POSE[:4])                  # the library takes the quaternion as it comes (Ceres keeps para_q normalised)


def _norm(pose):
    return pose[:4].copy(), pose[4:].copy()


@pytest.fixture(scope="module")
def drive(api, orc, synth):
    cfg = synth.default_cfg(64)
    scans = [synth.scan(cfg, 10 + 3 * k) for k in range(5)]
    P = orc.params(64)
    refs = [orc.extract(s, P) for s in scans]
    return dict(scans=scans, refs=refs, max_points=max(map(len, scans)))


def _check_corr(ctx, orc, k, cur, tgt, pose):
    q, t = _norm(pose)
    oes, oea, oeb = orc.associate_corner(q, t, cur["sharp"], tgt["less_sharp"])
    ops, opa, opb, opc = orc.associate_plane(q, t, cur["flat"], tgt["less_flat"])
    es, ea, eb = ctx.edge_corr(k)
    ps, pa, pb, pc = ctx.plane_corr(k)
    for got, want, nm in ((es, oes, "e_src"), (ea, oea, "e_a"), (eb, oeb, "e_b"), (ps, ops, "p_src"), (pa, opa, "p_a"), (pb, opb, "p_b"), (pc, opc, "p_c")):
        assert len(got) == len(want) and (got == want).all(), (k, nm)
    return ops, opa, opb, opc, oes, oea, oeb


def test_slot_target_and_carry_target_give_the_references_indices(api, orc, drive):
    """The same pair of scans associated twice: slot 2 against slot 1 inside a batch (the target's less-flat cloud is read in ring
    rows, its points are named by their place) and slot 2 against the carry copied from slot 1 (closed up on the device).  Both
    must come out as the oracle's contiguous indices, and the four clouds as the oracle's bytes."""
    scans, refs = drive["scans"], drive["refs"]
    ctx = api.Context(api.default_params(64, batch=4, max_points=drive["max_points"]))
    for k in range(4):
        ctx.upload_scan(k, scans[k])
    ctx.extract(0, 4)
    for k in range(4):
        f = ctx.features(k)
        for nm in ("sharp", "less_sharp", "flat", "less_flat"):
            assert_bit_equal(f[nm], refs[k][nm], f"slot {k} {nm}")
    orc.set_nn_mode(1)
    try:
        ctx.set_target_from_slot(0)
        ctx.associate(1, 3, POSE); ctx.vote(1, 3, True)
        for k in (1, 2, 3):
            _check_corr(ctx, orc, k, refs[k], refs[k - 1], POSE)
        assert ctx.pair_info(1).n_plane > 100
        ctx.set_target_from_slot(1)                       # slot 1's rows -> the contiguous carry
        ctx.associate(2, 1, POSE); ctx.vote(2, 1, True)
        _check_corr(ctx, orc, 2, refs[2], refs[1], POSE)
    finally:
        orc.set_nn_mode(0)
        ctx.close()


def test_stage_calls_over_sub_ranges_keep_the_associations_target(api, orc, drive):
    """ll_associate_batch(1, 4) then ll_vote_batch / ll_normal_equations_batch on single slots in the middle of the range: the slot's
    target is the one the association searched (slot k - 1), not "the carry, because the call starts here"."""
    scans, refs = drive["scans"], drive["refs"]
    ctx = api.Context(api.default_params(64, batch=5, max_points=drive["max_points"]))
    for k in range(5):
        ctx.upload_scan(k, scans[k])
    ctx.extract(0, 5)
    orc.set_nn_mode(1)
    try:
        ctx.set_target_from_slot(0)
        ctx.associate(1, 4, POSE)
        for k in (3, 2, 4):                               # any order, one slot per call
            ctx.vote(k, 1, True)
            ops, opa, opb, opc, oes, oea, oeb = _check_corr(ctx, orc, k, refs[k], refs[k - 1], POSE)
            q, t = _norm(POSE)
            cnt, sidx, sw = orc.vote(refs[k]["flat"][ops], refs[k - 1]["less_flat"][opa])
            order = np.sort(sidx); wmap = np.ones(len(ops), np.float32); wmap[sidx] = sw
            H, g, cost = orc.normal_equations(q, t, refs[k]["sharp"], oes, refs[k - 1]["less_sharp"], oea, oeb, refs[k]["flat"], ops[order],
                                              refs[k - 1]["less_flat"], opa[order], opb[order], opc[order], wmap[order], 0.1)
            ctx.normal_equations(k, 1, POSE)
            Hg, gg, cg = ctx.normal_equations_result(k)
            assert np.allclose(Hg, H, rtol=1e-9, atol=1e-9 * np.abs(H).max()) and np.allclose(gg, g, rtol=1e-9, atol=1e-9 * np.abs(g).max()), k
    finally:
        orc.set_nn_mode(0)
        ctx.close()


def test_uploaded_and_extracted_slots_mix(api, orc, drive):
    """A slot filled by ll_upload_features holds the caller's contiguous less-flat cloud, an extracted one ring rows: either can be
    the other's target inside one batch."""
    scans, refs = drive["scans"], drive["refs"]
    ctx = api.Context(api.default_params(64, batch=4, max_points=drive["max_points"]))
    r = refs
    ctx.upload_scan(0, scans[0]); ctx.extract(0, 1)                                                 # slot 0: extracted (rows)
    ctx.upload_features(1, r[1]["sharp"], r[1]["less_sharp"], r[1]["flat"], r[1]["less_flat"])      # slot 1: uploaded (contiguous), target = slot 0
    ctx.upload_scan(2, scans[2]); ctx.extract(2, 1)                                                 # slot 2: extracted, target = slot 1 (contiguous)
    ctx.upload_features(3, r[3]["sharp"], r[3]["less_sharp"], r[3]["flat"], r[3]["less_flat"])      # slot 3: uploaded, target = slot 2 (rows)
    orc.set_nn_mode(1)
    try:
        ctx.set_target(r[4]["less_sharp"], r[4]["less_flat"])                                       # slot 0's own target: a host cloud
        ctx.associate(0, 4, POSE); ctx.vote(0, 4, True)
        _check_corr(ctx, orc, 0, r[0], r[4], POSE)
        for k in (1, 2, 3):
            _check_corr(ctx, orc, k, r[k], r[k - 1], POSE)
        f = ctx.features(1)
        assert_bit_equal(f["less_flat"], r[1]["less_flat"], "uploaded slot's less_flat comes back as it went in")
    finally:
        orc.set_nn_mode(0)
        ctx.close()


def test_irregular_ring_rows_as_association_targets(api, orc):
    """Scans whose rings hold 0, 3, 9, ... 64, 65, 129, 400 points (empty rows, rows shorter than a 64-point chunk, rows ending on and
    just behind a chunk edge) as each other's targets: k_build_grid's walk over the ring rows -- prefix table, chunk cursor, the rows'
    ragged ends -- must hand k_associate exactly the reference's clouds, for refused scans too (no target: no correspondence)."""
    from test_gpu_parity import _vlp16_ring, _ring_scan
    rng = np.random.default_rng(5150)
    scans = []
    for s in range(24):
        rings = []
        for k in range(16):
            n = int(rng.choice([0, 3, 9, 12, 17, 30, 47, 63, 64, 65, 128, 129, 250, 400], p=[.06, .04, .05, .05, .08, .1, .1, .08, .08, .08, .08, .08, .06, .06]))
            if n == 0:
                continue
            base = rng.uniform(2.0, 12.0)
            r = base * (1.0 + rng.uniform(0.0005, 0.01) * np.cumsum(rng.standard_normal(n)))
            r = np.where(rng.random(n) < rng.uniform(0.0, 0.3), r * rng.uniform(1.2, 2.0), r)
            rings.append(_vlp16_ring(-15 + 2 * k, n, np.abs(r) + 0.35, phase=rng.random()))
        if not rings:
            rings.append(_vlp16_ring(1, 40, 5.0))
        scans.append(_ring_scan(rings))
    P = orc.params(16, minimum_range=0.3)
    refs = [orc.extract(sc, P) for sc in scans]
    ctx = api.Context(api.default_params(16, batch=len(scans), max_points=max(map(len, scans)) + 8, minimum_range=0.3))
    orc.set_nn_mode(1)
    try:
        for k, sc in enumerate(scans):
            ctx.upload_scan(k, sc)
        ctx.extract(0, len(scans))
        pose = np.array([0.0, 0.0, 0.002, 1.0, 0.05, -0.02, 0.0]); pose[:4] /= np.linalg.norm(pose[:4])
        ctx.set_target_from_slot(0)
        ctx.associate(1, len(scans) - 1, pose); ctx.vote(1, len(scans) - 1, False)
        checked = 0
        for k in range(1, len(scans)):
            if refs[k]["rc"] != 0:
                assert ctx.scan_info(k).status != 0
                continue
            tgt = refs[k - 1] if refs[k - 1]["rc"] == 0 else dict(less_sharp=np.zeros((0, 4), np.float32), less_flat=np.zeros((0, 4), np.float32))
            f = ctx.features(k)
            assert_bit_equal(f["less_flat"], refs[k]["less_flat"], f"scan {k} less_flat")
            _check_corr(ctx, orc, k, refs[k], tgt, pose)
            checked += 1
        assert checked >= 18
    finally:
        orc.set_nn_mode(0)
        ctx.close()
