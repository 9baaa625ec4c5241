"""The two mapping modes of SURVEY.md section 8e with their collectives on DEVICE buffers over RCCL (torch.distributed backend
"nccl" is RCCL on ROCm): the 44-double all-reduce of the normal equations and the all-gather of the K-NN candidates run
stream-ordered with the library's HIP stream, without a host hop (lightloam_amd.parallel.DeviceCollectives on the
ll_map_*_dev entry points).  World size 1 runs on every GPU box -- RCCL is initialised, the collectives are real RCCL calls on
the library's buffers; world size 2 needs two visible GPUs (the driver's multi-GPU node) and is skipped otherwise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    try:
        import torch
        import torch.distributed as dist
        import lightloam_amd  # noqa: F401
        from lightloam_amd import api, parallel, synth
        from oracle import orc
        from test_gpu_mapping import scene
        from test_gpu_cubemap import _pose7

        torch.cuda.set_device(rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))

        class Req:
            param = 16
        sc = scene.__wrapped__(Req, orc, synth)
        ctx = api.Context(api.default_params(16, batch=1, max_points=4096), device=rank)
        coll = parallel.DeviceCollectives(ctx, rank)
        caps = (len(sc["corner_map"]) + 8, len(sc["surf_map"]) + 8, len(sc["corner_stack"]) + 8, len(sc["surf_stack"]) + 8)
        full = api.Map(ctx, *caps)
        full.set_map(sc["corner_map"], sc["surf_map"]); full.set_scan(sc["corner_stack"], sc["surf_stack"])
        ref, ran = full.optimize(sc["guess"])
        assert ran
        # ---- row-parallel: every rank holds the map and a slice of the scan; all-reduce of the normal equations on the device
        m = api.Map(ctx, *caps)
        m.set_map(sc["corner_map"], sc["surf_map"])
        m.set_scan(sc["corner_stack"][rank::world], sc["surf_stack"][rank::world])
        pose = parallel.map_optimize_row_parallel_dev(m, coll, sc["guess"])
        host = parallel.map_optimize_row_parallel(m, sc["guess"], device=coll.dev)      # the host-hopped transport, same kernels
        assert np.abs(pose - ref).max() < 1e-7, (pose, ref)
        assert (pose == host).all(), (pose, host)
        if world == 1:
            assert (pose == ref).all()
        assert coll.n_allreduce == 2 * 5                         # 2 outer iterations x (1 + 4) evaluations
        m.close()
        # ---- tile-parallel: the MAP is split; all-gather of the candidates on the device
        rng = np.random.default_rng(5)
        own_c = rng.integers(0, world, len(sc["corner_map"])); own_s = rng.integers(0, world, len(sc["surf_map"]))
        ic = np.flatnonzero(own_c == rank).astype(np.int32); is_ = np.flatnonzero(own_s == rank).astype(np.int32)
        part = api.Map(ctx, *caps)
        part.set_map(sc["corner_map"][ic], sc["surf_map"][is_]); part.set_map_ids(ic, is_)
        part.set_scan(sc["corner_stack"], sc["surf_stack"])
        n_stack = (len(sc["corner_stack"]), len(sc["surf_stack"])); n_tot = (len(sc["corner_map"]), len(sc["surf_map"]))
        p_tile, ran = parallel.map_optimize_tile_parallel_dev(part, coll, sc["guess"], n_stack, n_tot)
        assert ran and (p_tile == ref).all(), (p_tile, ref)     # bit-identical to the unsplit map on one GPU
        p_rows, ran = parallel.map_optimize_tile_parallel_dev(part, coll, sc["guess"], n_stack, n_tot, row_parallel=True)
        assert ran and np.abs(p_rows - ref).max() < 1e-7
        part.close(); full.close()
        # ---- the whole frame loop on a sharded cube map against the unsplit one (rank 0 keeps the reference)
        cfg = synth.default_cfg(16)
        P = orc.params(16)
        cm = api.CubeMap(ctx, 4096, 32768, pool_points=1 << 18)
        cm.set_shard(rank, world)
        whole = api.CubeMap(ctx, 4096, 32768, pool_points=1 << 18) if rank == 0 else None
        for k in range(4):
            f = orc.extract(synth.scan(cfg, k), P)
            guess = _pose7(synth.pose(cfg, k)); guess[4:] += [0.08, -0.05, 0.02]
            pose, ran = parallel.cubemap_process_tile_parallel_dev(cm, coll, guess, f["less_sharp"], f["less_flat"])
            assert ran == (k > 0)
            if whole is not None:
                refp, ran_ref = whole.process(guess, f["less_sharp"], f["less_flat"])
                assert ran_ref == ran and (pose == refp).all(), (k, pose, refp)
        t = torch.from_numpy(pose).to(coll.dev); got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        assert all((got[0] == g).all() for g in got)
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


def _run(world):
    import multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    out = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_rccl_device_resident_collectives_world_1():
    _run(1)


def test_rccl_device_resident_collectives_world_2():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device: RCCL needs one GPU per rank (the driver's multi-GPU node runs this)")
    _run(2)
