"""The N > 1 path on CPU: world_size-2 gloo process group (127.0.0.1), the same host logic bench.py runs over RCCL.

 * scan-parallel: block partition of scans over ranks covers every scan exactly once; the job time is the max over ranks.
 * row-parallel GN: each rank accumulates the normal equations of ITS share of the residual blocks (here with the CPU
   oracle -- this is a test of the sharding + all-reduce logic, no GPU involved), the 28-double all-reduce must
   reproduce the single-process H, g, cost, and the GN step computed from it must be identical on both ranks.
 * tile-parallel mapping: each rank finds the five nearest map points among ITS share of the map (numpy here, the
   device's ll_map_knn_partial on the GPU box -- tests/test_gpu_tile_parallel.py), parallel._all_gather_np stacks the
   candidates [rank][query][5]; the five smallest (distance, id) of the stack must be the whole map's five nearest.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import lightloam_amd  # noqa: F401
    from lightloam_amd import parallel, synth
    from oracle import orc
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        # ---- scan-parallel bookkeeping
        first, count = parallel.shard_range(37, rank, world)
        mine = np.zeros(37); mine[first:first + count] = 1
        import torch
        t = torch.from_numpy(mine); dist.all_reduce(t)
        assert (t.numpy() == 1).all()
        assert parallel.max_over_ranks(1.0 + rank) == float(world)

        # ---- row-parallel Gauss-Newton on one VLP-16 scan pair
        cfg = synth.default_cfg(16)
        P = orc.params(16)
        e0 = orc.extract(synth.scan(cfg, 0), P); e1 = orc.extract(synth.scan(cfg, 1), P)
        q = np.array([0, 0, 0, 1.0]); tt = np.array([0.9, 0.0, 0.0])
        es, ea, eb = orc.associate_corner(q, tt, e1["sharp"], e0["less_sharp"])
        ps, pa, pb, pc = orc.associate_plane(q, tt, e1["flat"], e0["less_flat"])
        w = np.ones(len(ps), np.float32)
        full = orc.normal_equations(q, tt, e1["sharp"], es, e0["less_sharp"], ea, eb, e1["flat"], ps, e0["less_flat"], pa, pb, pc, w)
        fe, ce = parallel.shard_range(len(es), rank, world); fp, cp = parallel.shard_range(len(ps), rank, world)
        se, sp = slice(fe, fe + ce), slice(fp, fp + cp)
        part = orc.normal_equations(q, tt, e1["sharp"], es[se], e0["less_sharp"], ea[se], eb[se], e1["flat"], ps[sp],
                                    e0["less_flat"], pa[sp], pb[sp], pc[sp], w[sp])
        H, g, cost = parallel.allreduce_normal_equations(*part)
        assert np.allclose(H, full[0], rtol=1e-12, atol=1e-9) and np.allclose(g, full[1], rtol=1e-12, atol=1e-9)
        assert abs(cost - full[2]) < 1e-9 * max(1.0, abs(full[2]))
        rc, d = orc.gn_solve(H, g)
        assert rc == 0
        gathered = [torch.zeros(6, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(d))
        assert all((gathered[0] == x).all() for x in gathered)      # every rank takes the identical step

        # ---- tile-parallel K = 5 search: shard the map, all-gather the candidates, merge
        rng = np.random.default_rng(5)                                  # same stream on both ranks
        cloud = rng.uniform(-3, 3, (400, 3)).astype(np.float32)
        cloud[100:110] = cloud[90:100]                                  # exact duplicates: distance ties across ranks
        qs = rng.uniform(-3, 3, (64, 3)).astype(np.float32)
        owner = rng.integers(0, world, len(cloud))
        ids = np.flatnonzero(owner == rank).astype(np.int32)

        def five(points, gid):
            d = ((qs[:, None, :] - points[None, :, :]) ** 2).astype(np.float32)
            d = (d[:, :, 0] + d[:, :, 1]) + d[:, :, 2]
            order = np.lexsort((np.broadcast_to(gid, d.shape), d), axis=1)[:, :5]
            return np.take_along_axis(d, order, 1), gid[order]
        d_loc, id_loc = five(cloud[ids], ids)
        d_all = parallel._all_gather_np(d_loc); id_all = parallel._all_gather_np(id_loc)
        assert d_all.shape == (world, 64, 5) and id_all.dtype == np.int32
        assert (d_all[rank] == d_loc).all() and (id_all[rank] == id_loc).all()          # rank-major, own part in place
        dm = d_all.transpose(1, 0, 2).reshape(64, -1); im = id_all.transpose(1, 0, 2).reshape(64, -1)
        order = np.lexsort((im, dm), axis=1)[:, :5]
        d_ref, id_ref = five(cloud, np.arange(len(cloud), dtype=np.int32))
        assert (np.take_along_axis(im, order, 1) == id_ref).all() and (np.take_along_axis(dm, order, 1) == d_ref).all()
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        out.put((rank, repr(e) + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_world_size_two_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = [out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_shard_range_partitions():
    from lightloam_amd import parallel
    for n in (0, 1, 7, 64, 257):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                f, c = parallel.shard_range(n, r, world)
                seen += list(range(f, f + c))
            assert seen == list(range(n))


def test_pack_unpack_roundtrip():
    from lightloam_amd import parallel
    rng = np.random.default_rng(0)
    A = rng.normal(size=(9, 6)); H = A.T @ A; g = rng.normal(size=6)
    H2, g2, c2 = parallel.unpack_normal_equations(parallel.pack_normal_equations(H, g, 3.5))
    assert np.allclose(H, H2) and np.allclose(g, g2) and c2 == 3.5
