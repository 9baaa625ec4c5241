"""SURVEY.md section 8e row 3, tile-parallel mapping: the MAP is split over ranks (cube by cube), every rank searches its
own points, the candidates are all-gathered and merged.  The bar: exactly what the unsplit map gives -- the same
residual blocks, bit-identical optimised poses, and cube maps whose union is the unsplit cube map bit for bit."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from test_gpu_mapping import scene  # noqa: F401  (module-scoped fixture: one map + one scan per ring count)
from test_gpu_cubemap import frames, _pose7, _nonempty  # noqa: F401

pytestmark = pytest.mark.gpu


def _full_and_parts(api, sc, n_parts, seed):
    """the unsplit Map and n_parts Maps that each hold a random subset of both clouds (ids = positions in the unsplit cloud)"""
    ctx = api.Context(api.default_params(sc["rings"], batch=1, max_points=4096))
    caps = (len(sc["corner_map"]) + 8, len(sc["surf_map"]) + 8, len(sc["corner_stack"]) + 8, len(sc["surf_stack"]) + 8)
    full = api.Map(ctx, *caps)
    full.set_map(sc["corner_map"], sc["surf_map"]); full.set_scan(sc["corner_stack"], sc["surf_stack"])
    rng = np.random.default_rng(seed)
    own_c = rng.integers(0, n_parts, len(sc["corner_map"])); own_s = rng.integers(0, n_parts, len(sc["surf_map"]))
    parts = []
    for r in range(n_parts):
        ic = np.flatnonzero(own_c == r).astype(np.int32); is_ = np.flatnonzero(own_s == r).astype(np.int32)
        m = api.Map(ctx, *caps)
        m.set_map(sc["corner_map"][ic], sc["surf_map"][is_]); m.set_map_ids(ic, is_)
        m.set_scan(sc["corner_stack"], sc["surf_stack"])
        parts.append(m)
    return ctx, full, parts


def _merged_associate(parts, pose):
    cand = [m.knn_partial(pose) for m in parts]
    stacked = [np.stack([c[k] for c in cand]) for k in range(4)]
    for m in parts:
        m.associate_merged(*stacked, pose_w=pose)


@pytest.mark.parametrize("n_parts", [1, 2, 5])
def test_split_map_gives_the_blocks_of_the_whole_map(api, scene, n_parts):  # noqa: F811
    ctx, full, parts = _full_and_parts(api, scene, n_parts, 7 + n_parts)
    full.associate(scene["guess"])
    _merged_associate(parts, scene["guess"])
    ne, npl = full.counts()
    assert ne > 20 and npl > 100
    for m in parts:
        assert m.counts() == (ne, npl)
        for a, b in zip(m.edges(), full.edges()):
            assert_bit_equal(a, b, "edges")
        for a, b in zip(m.planes(), full.planes()):
            assert_bit_equal(a, b, "planes")
    for m in parts + [full]:
        m.close()
    ctx.close()


def test_candidates_are_sorted_and_padded(api, scene):  # noqa: F811
    ctx, full, parts = _full_and_parts(api, scene, 3, 3)
    cn, ci, sn, si = parts[0].knn_partial(scene["guess"])
    for nn, ids in ((cn, ci), (sn, si)):
        d = nn[:, :, 3]
        assert (d[:, 1:] >= d[:, :-1]).all()                                     # ascending distance
        pad = ids == np.iinfo(np.int32).max
        assert (np.isinf(d) == pad).all()                                        # unused slots: INFINITY + INT_MAX, at the end
        assert (np.diff(pad.astype(np.int8), axis=1) >= 0).all()
        assert (ids[~pad] >= 0).all()
    for m in parts + [full]:
        m.close()
    ctx.close()


def test_split_map_optimises_to_the_same_pose(api, scene):  # noqa: F811
    from lightloam_amd import parallel
    ctx, full, parts = _full_and_parts(api, scene, 3, 11)
    ref, ran = full.optimize(scene["guess"])
    assert ran
    # the ranks in lock step inside one process: the "all-gather" hands every rank everybody's candidates
    pose = [scene["guess"].copy() for _ in parts]
    for _ in range(2):
        cand = [m.knn_partial(p) for m, p in zip(parts, pose)]
        stacked = [np.stack([c[k] for c in cand]) for k in range(4)]
        for r, m in enumerate(parts):
            m.associate_merged(*stacked, pose_w=pose[r])
            pose[r] = m.solve(pose[r])
    for p in pose:
        assert (p == ref).all(), (p, ref)
    # a shard refuses the single-rank entry point instead of quietly searching a part of the map
    with pytest.raises(api.LightLoamError):
        parts[0].optimize(scene["guess"])
    # too small a map (:1822): nothing runs
    out, ran = parallel.map_optimize_tile_parallel(parts[0], scene["guess"], (1, 1), (5, 500), gather=lambda a: a[None])
    assert not ran and (out == scene["guess"]).all()
    for m in parts + [full]:
        m.close()
    ctx.close()


@pytest.mark.parametrize("world,offset", [(2, (0.0, 0.0, 0.0)), (3, (-431.0, 512.5, 30.0))])
def test_sharded_cube_maps_are_the_unsplit_cube_map(api, frames, world, offset):  # noqa: F811
    ctx = api.Context(api.default_params(16, batch=1, max_points=4096))
    whole = api.CubeMap(ctx, 4096, 32768, pool_points=1 << 18)
    shards = [api.CubeMap(ctx, 4096, 32768, pool_points=1 << 18) for _ in range(world)]
    for r, cm in enumerate(shards):
        cm.set_shard(r, world)
    owners_seen = set()
    for k, f in enumerate(frames):
        guess = _pose7(f["pose3"], offset); guess[4:] += [0.08, -0.05, 0.02]
        ref, ran_ref = whole.process(guess, f["corner"], f["surf"])
        for cm in shards:
            cm.prepare(guess[4:], f["corner"], f["surf"])
        infos = [cm.info() for cm in shards]
        assert all(i[0] == whole.info()[0] for i in infos)                        # same window position everywhere
        tot = np.sum([i[1][:2] for i in infos], axis=0)
        assert tuple(int(v) for v in tot) == tuple(whole.info()[1][:2])          # together: the clouds gathered from the whole map
        n_stack = infos[0][1][2:]
        pose = [guess.copy() for _ in shards]
        ran = tot[0] > 10 and tot[1] > 50
        assert ran == ran_ref
        if ran:
            maps = [cm.map() for cm in shards]
            for _ in range(2):
                cand = [m.knn_partial(p, n_stack) for m, p in zip(maps, pose)]
                stacked = [np.stack([c[j] for c in cand]) for j in range(4)]
                for r, m in enumerate(maps):
                    m.associate_merged(*stacked, pose_w=pose[r])
                    pose[r] = m.solve(pose[r])
        for r, cm in enumerate(shards):
            assert (pose[r] == ref).all(), (k, r, pose[r], ref)
            cm.update(pose[r])
        # cube by cube: one owner holds exactly the unsplit cube, the others nothing
        for s in (0, 1):
            for i in range(4851):
                w = whole.cube(s, i, cap=1 << 15)
                if not len(w):
                    continue
                held = [cm.cube(s, i, cap=1 << 15) for cm in shards]
                own = [r for r, h in enumerate(held) if len(h)]
                assert len(own) == 1, (k, s, i, own)
                owners_seen.add(own[0])
                assert_bit_equal(held[own[0]], w, f"frame {k} cube {s} {i}")
    assert len(owners_seen) >= 2                                                  # the map really was split
    with pytest.raises(api.LightLoamError):
        shards[0].process(guess, frames[0]["corner"], frames[0]["surf"])          # shards refuse the single-rank path
    with pytest.raises(api.LightLoamError):
        shards[0].set_shard(0, 2)                                                 # and re-sharding a filled map
    for cm in shards + [whole]:
        cm.close()
    ctx.close()


# ---- two processes, gloo all-gather (RCCL needs one device per rank; the same code path runs over it on a multi-GPU node)
def _tile_parallel_worker(rank, world, port, out):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        import numpy as np
        import torch
        import torch.distributed as dist
        import lightloam_amd  # noqa: F401
        from lightloam_amd import api, parallel, synth
        from oracle import orc
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        cfg = synth.default_cfg(16)
        P = orc.params(16)
        ctx = api.Context(api.default_params(16, batch=1, max_points=4096))
        cm = api.CubeMap(ctx, 4096, 32768, pool_points=1 << 18)
        cm.set_shard(rank, world)
        cm_rows = api.CubeMap(ctx, 4096, 32768, pool_points=1 << 18)              # tiles for the search AND rows for the solve
        cm_rows.set_shard(rank, world)
        whole = api.CubeMap(ctx, 4096, 32768, pool_points=1 << 18) if rank == 0 else None
        for k in range(5):
            f = orc.extract(synth.scan(cfg, k), P)
            guess = _pose7(synth.pose(cfg, k)); guess[4:] += [0.08, -0.05, 0.02]
            pose, ran = parallel.cubemap_process_tile_parallel(cm, guess, f["less_sharp"], f["less_flat"])
            assert ran == (k > 0)
            if whole is not None:
                ref, ran_ref = whole.process(guess, f["less_sharp"], f["less_flat"])
                assert ran_ref == ran and (pose == ref).all(), (k, pose, ref)
            got = [torch.zeros(7, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(got, torch.from_numpy(pose))
            assert all((got[0] == g).all() for g in got)
            # BASELINE config 4 to the letter: all-reduce of JtJ / Jtr between the ranks' row shards
            pose_r, ran_r = parallel.cubemap_process_tile_parallel(cm_rows, guess, f["less_sharp"], f["less_flat"], row_parallel=True)
            assert ran_r == ran and np.abs(pose_r - pose).max() < 1e-7, (k, pose_r, pose)
            dist.all_gather(got, torch.from_numpy(pose_r))
            assert all((got[0] == g).all() for g in got)
            if ran:
                m = cm_rows.map()
                full_rows = m.evaluate()[43]
                m.set_row_shard(rank, world); mine = m.evaluate()[43]; m.set_row_shard(0, 1)
                t = torch.tensor([mine]); dist.all_reduce(t)
                assert t.item() == full_rows and 0 < mine < full_rows                # the shards partition the rows
                m.set_row_shard(rank, world)
                try:
                    m.solve(pose_r)
                    raise AssertionError("a row shard must refuse the single-rank solve")
                except api.LightLoamError as e:
                    assert e.code == -7
                m.set_row_shard(0, 1)
        cm_rows.close()
        cm.close()
        if whole is not None:
            whole.close()
        ctx.close()
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        out.put((rank, repr(e) + traceback.format_exc()))
    finally:
        try:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass


def test_tile_parallel_two_ranks_match_one():
    import multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    out = mpc.Queue()
    procs = [mpc.Process(target=_tile_parallel_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
