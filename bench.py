#!/usr/bin/env python3
"""bench.py -- scans/s of the Light-LOAM per-scan hot path on MI355X.

One STEP = one pass of the hot path (extract + associate + vote + normal equations + one GN step,
BASELINE.json's metric unit) over a batch of synthetic KITTI-shape 64-ring scans that is already resident in
HBM when the timed region starts.  `value` = scans processed by all ranks per second.

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Started under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) this
process IS a rank; started as a plain command line it only LAUNCHES the N ranks as fresh child processes -- before it imports
torch or makes any HIP call -- relays rank 0's JSON line as its own last line and returns the children's exit code.

Scans shard across GPUs with no data-path collective (scan-parallel): every rank owns
`--batch` scans, so scaling is weak.  torch is used for the process group (RCCL), the barrier and the device
synchronisation only; the work is the HIP library behind include/lightloam_hip.h.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md chip table: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0        # ... and the float4 copy rate the guide quotes for the part (same table); stream_rates() has what THIS tree measured
N_SIMD = 1024                # 256 CUs x 4 SIMD-32 (same table)
CLOCK_GHZ = 2.4              # peak engine clock; the board sits at 2.34-2.35 GHz during the bench (tools/sample_clocks.sh)
VALU_CYCLES = 2.2            # cycles a SIMD needs per wave64 vector instruction with >= 4 waves feeding it (tools/ubench/valu_rate.hip)


def emit(out):
    """The ONE JSON line, as the last line of stdout: whatever native libraries (RCCL's version banner) still hold in
    their C stdio buffers is flushed first."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


def workload_source(args, seed):
    """(scan(k) -> (n, 4) float32, pose(k) -> (x, y, yaw)) of the named workload.
    synthetic: host/ll_synth.c, every ring on the bin centre of scanRegistration.cpp:162.
    hdl64:     lightloam_amd/hdl64.py, the HDL-64E true laser table in a KITTI .bin's order (BASELINE config 3 stand-in:
               elevations anywhere inside the bins, some bins hold two lasers -> ring capacity 4608)."""
    if args.workload == "hdl64":
        from lightloam_amd import hdl64
        return (lambda k: hdl64.hdl64_scan(k, order="kitti", seed=64 + (seed & 0xffff))), (lambda k: hdl64.pose(k))
    from lightloam_amd import synth
    cfg = synth.default_cfg(args.rings, seed=seed)
    return (lambda k: synth.scan(cfg, k)), (lambda k: synth.pose(cfg, k))


def build_workload(args, batch, seed):
    """batch+1 scans: slot i (and the carry target, index -1) follow a ping-pong walk over `distinct`+1 consecutive
    poses, so every (slot k-1, slot k) pair is a pair of ADJACENT poses (one step forward or backward)."""
    distinct = args.distinct
    scan, pose = workload_source(args, seed)
    base = [scan(k) for k in range(distinct + 1)]
    poses = [pose(k) for k in range(distinct + 1)]

    def tri(i):
        period = 2 * distinct
        j = i % period
        return j if j <= distinct else period - j

    order = [tri(i) for i in range(batch + 1)]          # order[0] is the carry, order[i+1] is slot i
    guesses = np.zeros((batch, 7))
    for i in range(batch):
        a, b = poses[order[i]], poses[order[i + 1]]      # previous, current
        dyaw = b[2] - a[2]
        c, s = np.cos(a[2]), np.sin(a[2])
        dx, dy = b[0] - a[0], b[1] - a[1]
        # current -> previous frame transform, perturbed: the warm start the reference carries between frames
        t = np.array([c * dx + s * dy, -s * dx + c * dy, 0.0]) * 0.9
        yaw = dyaw * 0.9
        guesses[i] = [0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2), t[0], t[1], t[2]]
    return base, order, guesses


def ring_model_params(args):
    """context / oracle parameters beyond the ring count: the linear ring model for ring counts the reference's switch does not
    know (BASELINE config 5's 128 rings), the ring capacity"""
    extra = dict(ring_model=1, lower_bound=-25.0, up_bound=15.0, minimum_range=0.3) if args.rings not in (16, 32, 64) else {}
    return extra


def metric_name(args, suffix=""):
    return "scans/sec (feature-extract+match+one GN iter), %d-ring cloud%s" % (args.rings, suffix)


def workload_name(args):
    if args.workload == "hdl64":
        return ("HDL-64E true laser table, KITTI .bin order (64 lasers in two blocks, ~1900 azimuths, per-laser mounting offsets; "
                "elevations off the bin centres of scanRegistration.cpp:162, ring capacity %d)" % args.max_ring_points)
    return ("HDL-64E / KITTI-shape scan (64 rings x 2048 azimuths, min_range 5 m)" if args.rings == 64 else
            "dense 128-ring scan (128 rings x 2048 azimuths over [-25, +15] deg, linear ring model)" if args.rings == 128 else
            "%d-ring synthetic scan" % args.rings)


def source_digest():
    """sha256 (16 hex digits) over the library's sources: what ties profiles/pmc_traffic.json to the code it was measured on"""
    import hashlib
    h = hashlib.sha256()
    for d, exts in ((os.path.join(ROOT, "light-loam_amd", "csrc"), (".hip", ".h")), (os.path.join(ROOT, "include"), (".h",))):
        for f in sorted(os.listdir(d)):
            if f.endswith(exts):
                h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(orc, rings, base, order, guesses, budget_s=15.0):
    """The oracle (kind 'port': our CPU restatement, 1 thread, grid-accelerated exact NN) over a bounded sample
    of the same units of work."""
    P = orc.params(rings) if rings in (16, 32, 64) else orc.params(rings, minimum_range=0.3, lower_bound=-25.0, up_bound=15.0, ring_model=1)
    orc.set_nn_mode(1)
    units = 0
    t0 = time.perf_counter()
    prev = orc.extract(base[order[0]], P)                 # the carry is not counted
    t0 = time.perf_counter()
    while True:
        i = units % (len(order) - 1)
        cur = orc.extract(base[order[i + 1]], P)
        if i == 0:
            prev_use = orc.extract(base[order[0]], P) if units else prev
        q, t = guesses[i][:4], guesses[i][4:]
        es, ea, eb = orc.associate_corner(q, t, cur["sharp"], prev_use["less_sharp"])
        ps, pa, pb, pc = orc.associate_plane(q, t, cur["flat"], prev_use["less_flat"])
        cnt, sidx, sw = orc.vote(cur["flat"][ps], prev_use["less_flat"][pa])
        H, g, cost = orc.normal_equations(q, t, cur["sharp"], es, prev_use["less_sharp"], ea, eb, cur["flat"], ps[sidx],
                                          prev_use["less_flat"], pa[sidx], pb[sidx], pc[sidx], sw, 0.1)
        rc, d = orc.gn_solve(H, g)
        prev_use = cur
        units += 1
        el = time.perf_counter() - t0
        if el >= budget_s or units >= 4096:
            break
    orc.set_nn_mode(0)
    return units / el, units, el


def _host_topology():
    """(logical cpus this process may use, ordered so that the first P entries are one hardware thread of each of the P
    physical cores; P)"""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    first, rest, seen = [], [], set()
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib in seen:
            rest.append(c)
        else:
            seen.add(sib); first.append(c)
    return first + rest, len(first)


def cpu_baseline_host(args, rank_seed):
    """The CPU baseline on the box's host cores, run BEFORE this process touches the GPU, in child processes
    (`bench.py --cpu-worker`, each rebuilding the same synthetic stream from its seed and PINNED to its own logical cpu, one
    hardware thread per physical core first): a sweep over 1 / 8 / 64 / all cpus, one scan stream per process.  The one-core
    figure is what a reference node gets (the ROS nodes are single-threaded, scanRegistration.cpp:475); `value` is the
    all-cpu figure, the most the host can give."""
    import subprocess
    cpus, physical = _host_topology()

    def run(n_proc, budget):
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", NUMEXPR_NUM_THREADS="1")   # one thread per process
        procs = []
        for i in range(n_proc):
            cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--rings", str(args.rings), "--batch", str(args.batch),
                   "--distinct", str(args.distinct), "--cpu-budget", str(budget), "--seed", str(rank_seed), "--pin", str(cpus[i % len(cpus)]),
                   "--workload", args.workload]
            procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
        res = []
        for p in procs:
            try:
                so, se = p.communicate(timeout=budget + 240.0)
            except subprocess.TimeoutExpired:
                p.kill(); so, se = p.communicate()
            if p.returncode != 0:
                raise RuntimeError(f"CPU baseline worker exited with {p.returncode}: " + (se or so)[-2000:])
            res.append(json.loads(so.strip().splitlines()[-1]))
        return res

    points = sorted({n for n in (1, 8, 64, len(cpus)) if n <= len(cpus)})
    share = args.cpu_budget / (len(points) + 1.0)                  # the one-core point gets a double share
    sweep, detail = {}, {}
    one = None
    for n in points:
        try:
            r = run(n, share * (2.0 if n == 1 else 1.0))
        except Exception as e:                                     # a reported baseline, not the product: keep what was measured, say why
            detail[str(n)] = f"failed: {e}"
            continue
        sweep[str(n)] = sum(x["value"] for x in r)
        detail[str(n)] = f"{sum(x['units'] for x in r)} scan pairs in {max(x['elapsed'] for x in r):.1f} s"
        if n == 1:
            one = r[0]
    if one is None:
        raise RuntimeError("CPU baseline: the one-core run failed: " + str(detail))
    top = max((int(k) for k in sweep), default=1)
    return {"value": sweep[str(top)], "unit": "scans/s", "cores": top, "kind": "port", "single_thread": one["value"],
            "physical_cores": physical, "logical_cpus": len(cpus), "pinned": True, "sweep_scans_per_s": sweep,
            "sample": f"one scan stream per process, every process pinned to its own logical cpu (one hardware thread per physical "
                      f"core first; {physical} physical cores, {len(cpus)} logical cpus): " + "; ".join(f"{k} cpu(s): {v}" for k, v in detail.items()) +
                      f"; one core = {one['value']:.1f} scans/s is what a single-threaded reference node gets; "
                      "oracle/ll_oracle.c (extract + grid-NN associate + vote + autodiff normal equations + solve)"}


def rccl_check(torch, dist, world, rank, local_rank, iters=200):
    """{"rccl_world": N, "rccl_allreduce_28xf64_us": t} -- one RCCL all-reduce of 28 doubles per iteration on a device buffer;
    a failure to bring RCCL up is reported, not fatal (the headline number does not depend on it)."""
    try:
        import torch.distributed as tdist
        if not tdist.is_initialized():
            import socket
            if "MASTER_PORT" not in os.environ:
                s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            tdist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dev = torch.device("cuda", local_rank)
        v = torch.arange(28, dtype=torch.float64, device=dev) * (rank + 1)
        tdist.all_reduce(v)
        want = torch.arange(28, dtype=torch.float64, device=dev) * (world * (world + 1) / 2)
        ok = bool((v == want).all().item())
        buf = torch.zeros(28, dtype=torch.float64, device=dev)
        for _ in range(10):
            tdist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            tdist.all_reduce(buf)
        e1.record(); torch.cuda.synchronize()
        return {"rccl_world": world if ok else None, "rccl_allreduce_28xf64_us": 1e3 * e0.elapsed_time(e1) / iters, "rccl_sum_correct": ok}
    except Exception as e:                                   # pragma: no cover
        return {"rccl_world": None, "rccl_error": repr(e)[:300]}


def bench_stream(args, rank, local_rank, world):
    """BASELINE config 5 (a stream of scans, batched multi-scan pipeline): the hot path with EVERY scan uploaded from page-locked
    host memory inside the timed region.  The slots form two halves; while one half is processed (compute stream) the other
    half's scans are copied in (copy stream), ordered by events only.  Reports the sustained scans/s including H2D, the
    copy-only and compute-only times of the same work, and checks the poses against the all-resident run bit for bit."""
    import torch
    from lightloam_amd import api
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    B = args.batch if not args.batch_defaulted else (2048 if args.rings <= 64 else 1024)      # default: 2048 slots (two halves of 1024)
    B -= B % 2
    H = B // 2
    base, order, guesses = build_workload(args, B, 0x5EED0000 + rank)
    extra = ring_model_params(args)
    if args.max_ring_points > 0:
        extra["max_ring_points"] = args.max_ring_points
    extra["input_stride_floats"] = args.input_stride
    ctx = api.Context(api.default_params(args.rings, batch=B + 1, max_points=max(len(s) for s in base), **extra), device=local_rank)
    # the ingest buffer: one page-locked area with a slot per scan of the step (what a driver thread would fill from the sensor)
    NPs = (max(len(s) for s in base) + 63) // 64 * 64
    staging = api.PinnedStaging(B, NPs, args.input_stride)
    for i in range(B):
        staging.put(i, base[order[i + 1]])
    ctx.upload_scan(B, base[order[0]]); ctx.extract(B, 1); ctx.set_target_from_slot(B)
    ctx.set_pose_guess(0, B, guesses)
    COMPUTE, COPY = 0, 1

    def resident_step():
        ctx.hot_path(0, B, None, vote=True)

    def upload_all():
        ctx.upload_staging_async(0, staging, 0, H); ctx.upload_staging_async(H, staging, H, H)

    def stream_step():
        for h in (0, 1):
            f = h * H
            ctx.stream_wait(COPY, 2 + h)                       # the slots of this half were last read by the compute marked 2 + h
            ctx.upload_staging_async(f, staging, f, H)
            ctx.stream_record(COPY, h)
            ctx.stream_wait(COMPUTE, h)
            if h == 0:
                ctx.hot_path(0, H, None, vote=True)
            else:
                ctx.hot_path_chain(H, H, vote=True)
            ctx.stream_record(COMPUTE, 2 + h)

    def timed(fn, n, sync):
        sync(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync(); return (time.perf_counter() - t0) / n

    both = lambda: (ctx.synchronize_copy(), ctx.synchronize())
    upload_all(); both()
    resident_step(); ctx.synchronize()
    ref = np.stack([ctx.pose(i) for i in range(B)])
    t_compute = timed(resident_step, max(2, args.steps // 4), ctx.synchronize)
    t_copy = timed(upload_all, max(2, args.steps // 4), both)
    for _ in range(args.warmup):
        stream_step()
    t_stream = timed(stream_step, args.steps, both)
    got = np.stack([ctx.pose(i) for i in range(B)])
    assert got.tobytes() == ref.tobytes(), "streamed run differs from the resident run"
    bad = [i for i in range(B) if ctx.scan_info(i).status != 0 or ctx.pair_info(i).n_plane_selected <= 0]
    assert not bad, bad[:5]
    nbytes = sum(int(staging.n[i]) * 4 * args.input_stride for i in range(B))
    out = {"metric": metric_name(args, ", input streamed over PCIe"),
           "value": B / t_stream, "unit": "scans/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_stream,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
           "dtype": "f32 (features, association, vote) + f64 (residuals, Jacobians, normal equations)",
           "config": {"workload": workload_name(args) + ": the hot path with every scan copied from page-locked host memory inside the timed region, "
                                  "double-buffered slot halves (the chained half votes through the dynamic-LDS k_vote path at 128 rings)",
                      "points_per_scan_in": int(staging.n[0]), "input_bytes_per_point": 4 * args.input_stride,
                      "scans_per_step": B, "half": H, "h2d_bytes_per_step": nbytes, "distinct_scans": args.distinct + 1},
           "stream": {"ms_copy_only": 1e3 * t_copy, "ms_compute_only": 1e3 * t_compute, "ms_overlapped": 1e3 * t_stream,
                      "h2d_GBps_alone": nbytes / t_copy / 1e9, "h2d_GBps_sustained": nbytes / t_stream / 1e9,
                      "hidden_fraction_of_the_shorter_leg": (t_copy + t_compute - t_stream) / min(t_copy, t_compute),
                      "efficiency_vs_the_longer_leg": max(t_copy, t_compute) / t_stream,
                      "bit_identical_to_resident_run": True}}
    emit(out)
    staging.close()
    ctx.close()


def bench_map(args, rank, local_rank, world):
    """BASELINE config 4: laserMapping scan-to-submap (laserMapping.cpp:1584-2165) with the 21 x 21 x 11 cube map sharded over
    the ranks (tile-parallel K-NN + all-gather of the candidates) and, with --row-parallel, the Levenberg-Marquardt evaluations
    split as well (RCCL all-reduce of the 44-double normal-equation record per evaluation).  Every rank gets the whole scan of
    every frame; a STEP is one frame.  All collectives run on device buffers on the library's stream (parallel.DeviceCollectives)."""
    import torch
    import torch.distributed as dist
    from lightloam_amd import api, parallel
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
    dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    scan_of, pose_of = workload_source(args, 0x5EED0000)
    n_frames = args.warmup + args.steps
    distinct = min(n_frames, 40)
    scans = [scan_of(k) for k in range(distinct)]
    poses = [pose_of(k) for k in range(distinct)]
    extra = ring_model_params(args)
    if args.max_ring_points > 0:
        extra["max_ring_points"] = args.max_ring_points
    ctx = api.Context(api.default_params(args.rings, batch=distinct, max_points=max(map(len, scans)), **extra), device=local_rank)
    for k, sc in enumerate(scans):
        ctx.upload_scan(k, sc)
    ctx.extract(0, distinct)
    feats = [ctx.features(k) for k in range(distinct)]            # what laserOdometry publishes to laserMapping (host side of the seam)
    coll = parallel.DeviceCollectives(ctx, local_rank)
    cm = api.CubeMap(ctx, 16384, 65536, pool_points=1 << 22)
    cm.set_shard(rank, world)

    def frame(i):
        k = i % distinct if (i // distinct) % 2 == 0 else distinct - 1 - (i % distinct)     # drive to and fro over the same road
        x, y, yaw = poses[k]
        guess = np.array([0.0, 0.0, np.sin(yaw / 2), np.cos(yaw / 2), x + 0.05, y - 0.04, 0.02])
        return parallel.cubemap_process_tile_parallel_dev(cm, coll, guess, feats[k]["less_sharp"], feats[k]["less_flat"],
                                                          row_parallel=args.row_parallel)

    for i in range(args.warmup):
        frame(i)
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ran = 0
    for i in range(args.warmup, n_frames):
        pose, r = frame(i); ran += int(r)
    dist.barrier(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda"); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    assert np.isfinite(pose).all() and ran == args.steps
    # every rank must hold the same pose: the LM state is replicated (tile-parallel) or all-reduced (row-parallel)
    pt = torch.from_numpy(pose).to("cuda"); got = [torch.zeros_like(pt) for _ in range(world)]
    dist.all_gather(got, pt)
    assert all(bool((got[0] == g).all()) for g in got), "ranks disagree on the pose"
    _, cnt = cm.info()
    rccl = rccl_check(torch, dist, world, rank, local_rank)
    # the collectives by themselves, on the buffers of the last frame
    own, al = coll.candidates((cnt[2], cnt[3]))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        for o, a in zip(own, al):
            dist.all_gather_into_tensor(a, o)
    e1.record(); torch.cuda.synchronize()
    gather_us = 1e3 * e0.elapsed_time(e1) / 50
    if rank == 0:
        out = {"metric": "laserMapping frames/sec (scan-to-submap, voxel-tiled map sharded over the GPUs)", "value": args.steps / elapsed, "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32 (K-NN, voxel filter) + f64 (line / plane fits, normal equations, LM)", "data": "synthetic",
               "config": {"workload": f"{args.rings}-ring synthetic drive, one laserMapping frame per step (prepare + 2 x (K=5 search, line/plane fit, LM <= 4 "
                                      "iterations) + map update), map = 21 x 21 x 11 cubes of 50 m sharded by cube over the ranks",
                          "parallelism": f"tile-parallel x{world}" + (" + row-parallel LM" if args.row_parallel else " (LM replicated)"),
                          "stack_points": [int(cnt[2]), int(cnt[3])], "map_points_this_rank": [int(cnt[0]), int(cnt[1])],
                          "collectives_per_frame": {"all_gather": coll.n_allgather / n_frames, "all_reduce_44xf64": coll.n_allreduce / n_frames},
                          "candidate_all_gather_us": gather_us, "candidate_bytes_per_rank": int(sum(o.numel() * o.element_size() for o in own))}}
        out.update(rccl)
        emit(out)
    cm.close(); ctx.close()
    dist.barrier(); dist.destroy_process_group()


def profile_suffix(args):
    """which counter files describe this workload: pmc_traffic.json / sq_issue.json for the 64-ring headline, _hdl64 for BASELINE config 3's
    stand-in, _s128 / _s16 for the other synthetic ring counts (tools/profile_round.sh writes them all)"""
    if args.workload != "synthetic":
        return "_" + args.workload
    return "" if args.rings == 64 else "_s%d" % args.rings


def traffic_from_profile(args, kernel, launches_per_step, path=None):
    """(HBM bytes per launch of `kernel` or None, why) from profiles/pmc_traffic.json -- the separate rocprofv3 FETCH_SIZE / WRITE_SIZE
    passes of this same command (tools/pmc_traffic.py, stamped by tools/profile_round.sh) -- used ONLY when the file was measured on
    THIS code (source digest), ring count, workload and batch."""
    tpath = path or os.path.join(ROOT, "profiles", "pmc_traffic%s.json" % profile_suffix(args))
    if not os.path.exists(tpath):
        return None, "profiles/pmc_traffic.json absent"
    try:
        T = json.load(open(tpath))
        want = {"source_digest": source_digest(), "rings": args.rings, "batch": args.batch, "workload": args.workload}
        diff = sorted(k for k, v in want.items() if T.get(k) != v)
        if diff:
            return None, "profiles/pmc_traffic.json was measured on another " + ", ".join(diff) + ": not used"
        if kernel not in T.get("kernels", {}):
            return None, "profiles/pmc_traffic.json has no entry for " + kernel
        return (T["kernels"][kernel]["hbm_bytes_per_scan"] * args.batch / launches_per_step,
                "rocprofv3 FETCH_SIZE + WRITE_SIZE passes of this command at this source digest (tools/profile_round.sh)")
    except Exception as e:
        return None, "profiles/pmc_traffic.json unreadable: " + repr(e)[:80]


def stream_rates(path=None):
    """What tools/ubench/stream_rate.hip measured on an MI355X of this pool on the library's own strides (rows of ring_cap float4,
    ~1650 used): the best read-only, write-only and copy rate over 1..8 loads in flight per lane and 2..8 workgroups per CU.
    These are the ceilings the streaming kernels are read against -- not the 4.6 TB/s of the naive
    calibration copy.  {} when the file is absent."""
    spath = path or os.path.join(ROOT, "profiles", "r05_stream_rate.json")
    try:
        R = json.load(open(spath))["stream_rate"]["results"]
    except Exception:
        return {}
    best = {}
    for r in R:
        if r["pattern"] != "rows":
            continue
        k = "read" if r["mode"].startswith("read") else r["mode"]
        best[k] = max(best.get(k, 0.0), float(r["GBps"]))
    return {"read_GBps": best.get("read"), "write_GBps": best.get("write"), "copy_GBps": best.get("copy"),
            "source": "profiles/r05_stream_rate.json (tools/stream_rate.sh: laserCloud's row stride, best of 1-8 loads in flight x 2-8 workgroups per CU)"}


def all_traffic_from_profile(args, kernels, launches_per_step):
    """Counter bytes per step of every kernel in `kernels` (same file and stamp rule as traffic_from_profile); None unless all are there."""
    tot = 0.0
    for k in kernels:
        t, _ = traffic_from_profile(args, k, launches_per_step)
        if t is None:
            return None
        tot += t * launches_per_step
    return tot


def issue_from_profile(args, kernels, path=None):
    """Instruction-issue figures of `kernels` from profiles/sq_issue.json -- SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS per full-batch
    launch, collected by tools/profile_round.sh (rocprofv3 --pmc, its own pass) and stamped like pmc_traffic.json; used only when
    the stamp matches this code, ring count, workload and batch.  (dict or None, why)"""
    ipath = path or os.path.join(ROOT, "profiles", "sq_issue%s.json" % profile_suffix(args))
    if not os.path.exists(ipath):
        return None, "profiles/sq_issue.json absent"
    try:
        T = json.load(open(ipath))
        want = {"source_digest": source_digest(), "rings": args.rings, "batch": args.batch, "workload": args.workload}
        diff = sorted(k for k, v in want.items() if T.get(k) != v)
        if diff:
            return None, "profiles/sq_issue.json was measured on another " + ", ".join(diff) + ": not used"
        got = {k: T["kernels"][k] for k in kernels if k in T.get("kernels", {})}
        if len(got) != len(kernels):
            return None, "profiles/sq_issue.json lacks " + ", ".join(k for k in kernels if k not in got)
        return got, "rocprofv3 SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS pass of this command at this source digest (tools/profile_round.sh)"
    except Exception as e:
        return None, "profiles/sq_issue.json unreadable: " + repr(e)[:80]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes of this same command line, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what torch.distributed.run would set), relay rank 0's stdout --
    its last line is the JSON line -- and return the worst exit code.  The parent never imports torch and never touches HIP:
    replacing or re-launching a process that has initialised the GPU is what this avoids."""
    import socket
    import subprocess
    n = args.gpus
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--dry-launch"]
    envs = []
    for r in range(n):
        e = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
        if "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ:           # the ranks see what a torchrun-launched rank would see: the parent's setting, if any
            e["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ["HSA_ENABLE_IPC_MODE_LEGACY"]
        envs.append(e)
    if args.dry_launch:
        emit({"dry_launch": True, "n_ranks": n, "command": cmd, "rank_env": envs, "parent_imported_torch": "torch" in sys.modules})
        return 0
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")                   # rank 0's stdout (a file, not a pipe: nothing to drain while polling)
    for r in range(n):
        env = dict(os.environ); env.update(envs[r])
        procs.append(subprocess.Popen(cmd, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=None, text=True))
    # a rank that dies (no such device, out of memory ...) would leave the others waiting in the rendezvous or a collective for
    # minutes: the first non-zero exit ends the run -- the survivors get a moment to finish, then they are terminated (they are
    # this process's own children, addressed by their exact pids)
    rcs = [None] * n
    failed_at = None
    started = time.monotonic()
    timed_out = False
    while any(c is None for c in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if failed_at is None and any(c not in (None, 0) for c in rcs):
            failed_at = time.monotonic()
        if not timed_out and args.launch_timeout > 0 and time.monotonic() - started > args.launch_timeout:
            # every rank alive but nothing moves (a rendezvous or a collective that never completes): end the run, never re-launch
            timed_out = True
            failed_at = time.monotonic() - 16.0
        if failed_at is not None and time.monotonic() - failed_at > 15.0:
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        p.kill(); rcs[r] = p.wait()
            break
        time.sleep(0.1)
    out0.seek(0)
    lines = [ln for ln in out0.read().splitlines() if ln.strip()]
    out0.close()
    for ln in lines[:-1]:
        print(ln)
    rc = 0 if all(c == 0 for c in rcs) and not timed_out else 1   # a signal-killed child reports a negative code: any failure is exit code 1
    if lines:
        sys.stdout.flush()
        print(lines[-1], flush=True)                           # rank 0's JSON line is this process's last line
    if rc != 0:
        print(f"bench.py: rank exit codes {rcs}" + (f" (no progress within --launch-timeout {args.launch_timeout:.0f} s: ranks terminated)" if timed_out else ""),
              file=sys.stderr)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="scans resident per GPU and processed per step; 0 = by ring count: 16384 for 64 rings "
                    "(189 GB of the 288; 8192: -2 %%), 4096 for more rings, 32768 for fewer; --stream-input: 2048")
    ap.add_argument("--chunk", type=int, default=0, help="scans per launch sequence inside a step (0 = whole batch)")
    ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic poses the batch cycles through (65 scans, 128 scan pairs: "
                    "ring lengths, pick counts and search depths differ from slot to slot)")
    ap.add_argument("--mode", choices=["hot", "map"], default="hot",
                    help="hot: the headline metric (scans/s of the per-scan hot path, scan-parallel).  map: BASELINE config 4, laserMapping "
                         "frames/s with the voxel-tiled map sharded over the GPUs and the RCCL all-reduce of JtJ / Jtr (strong scaling)")
    ap.add_argument("--row-parallel", action="store_true", help="--mode map: split the LM evaluations over the ranks too (all-reduce per evaluation)")
    ap.add_argument("--stream-input", action="store_true",
                    help="BASELINE config 5: every scan crosses PCIe in the timed region -- double-buffered slot halves, the upload of one half "
                         "overlapping the processing of the other; reports scans/s including H2D and how much of the work the copies hide")
    ap.add_argument("--input-stride", type=int, choices=[3, 4], default=4,
                    help="floats per resident raw point (ll_params.input_stride_floats): 4 = KITTI .bin / PointXYZ (the contract, default), "
                         "3 = x, y, z packed: the 4th float the reference never reads (scanRegistration.cpp:105-106) neither crosses PCIe nor is read from HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=24.0, help="seconds of CPU baseline, split over the 1 / 8 / 64 / all-cpu points of the sweep")
    ap.add_argument("--calibrate", action="store_true",
                    help="also launch k_calib_copy (1 GiB in + 1 GiB out) once: the known-byte launch tools/pmc_traffic.py "
                         "uses to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)       # child of cpu_baseline_host: no GPU, no torch
    ap.add_argument("--seed", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--pin", type=int, default=-1, help=argparse.SUPPRESS)                # cpu-worker: the logical cpu to run on
    ap.add_argument("--max-ring-points", type=int, default=0,
                    help="ring capacity of the context (0: 2304, enough for the synthetic 2048-column scans; --workload hdl64: 4608)")
    ap.add_argument("--workload", choices=["synthetic", "hdl64"], default="synthetic",
                    help="synthetic: SURVEY.md section 8d's scans (every ring on its bin centre; the headline workload).  hdl64: BASELINE config 3's "
                         "stand-in, the HDL-64E true laser table in KITTI .bin order (off-centre elevations, ring capacity 4608)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend of the N > 1 run (nccl = RCCL; gloo only to exercise the multi-rank path where RCCL cannot run)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="N > 1 ranks all on device 0 (a one-GPU box exercising the multi-rank path; needs --backend gloo and a small --batch)")
    ap.add_argument("--launch-timeout", type=float, default=1800.0,
                    help="--gpus N > 1 from a plain command line: seconds after which ranks that are all alive but not finished are terminated "
                         "(exit code 1; 0 = wait for ever)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="--gpus N > 1 from a plain command line: print the N child command lines + rank environments as JSON and exit "
                         "(nothing is started, torch is never imported)")
    args = ap.parse_args()
    args.batch_defaulted = args.batch <= 0
    if args.share_gpu and args.backend != "gloo":
        raise SystemExit("--share-gpu puts every rank on device 0: RCCL cannot run there, use --backend gloo")
    if args.mode == "map" and (args.share_gpu or args.backend != "nccl"):
        raise SystemExit("--mode map runs its collectives on device buffers over RCCL: --backend nccl, one GPU per rank (no --share-gpu)")
    if args.workload == "hdl64":
        if args.rings != 64:
            raise SystemExit("--workload hdl64 is a 64-ring sensor")
        if args.max_ring_points <= 0:
            args.max_ring_points = 4608
    if args.batch_defaulted:
        args.batch = 16384 if args.rings == 64 else 4096 if args.rings > 64 else 32768
        if args.max_ring_points > 2304:
            args.batch = 8192                                  # laserCloud's ring stride doubles: 8192 slots of capacity 4608 are ~150 GB
    if args.cpu_worker:
        if args.pin >= 0 and hasattr(os, "sched_setaffinity"):
            try:
                os.sched_setaffinity(0, {args.pin})
            except OSError:
                pass
        import lightloam_amd  # noqa: F401
        from oracle import orc
        base, order, guesses = build_workload(args, args.batch, args.seed)
        v, units, el = cpu_baseline(orc, args.rings, base, order, guesses, args.cpu_budget)
        print(json.dumps({"value": v, "units": units, "elapsed": el}), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: this process only launches the ranks (it has made no HIP call and never will)
        sys.exit(launch_ranks(args, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world                                      # the launcher's world size is the truth
    if args.share_gpu:
        local_rank = 0

    import lightloam_amd  # noqa: F401
    if args.mode == "map":
        return bench_map(args, rank, local_rank, world)
    if args.stream_input:
        return bench_stream(args, rank, local_rank, world)
    # every rank owns its own scans (different seed => different noise), same shape
    base, order, guesses = build_workload(args, args.batch, 0x5EED0000 + rank)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_host(args, 0x5EED0000 + rank)                              # before any HIP call in this process

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")
    red_dev = "cuda" if args.backend == "nccl" else "cpu"      # where the timing reduction lives (gloo reduces host tensors)

    from lightloam_amd import api
    max_pts = max(len(s) for s in base)
    extra = ring_model_params(args)
    if args.max_ring_points > 0:
        extra["max_ring_points"] = args.max_ring_points
    extra["input_stride_floats"] = args.input_stride
    prm = api.default_params(args.rings, batch=args.batch + 1, max_points=max_pts, chunk=args.chunk, **extra)
    ctx = api.Context(prm, device=local_rank)
    # slot B holds the carry scan: extract it once, make it the carry target, then load the batch
    ctx.upload_scan(args.batch, base[order[0]])
    ctx.extract(args.batch, 1)
    ctx.set_target_from_slot(args.batch)
    for i in range(args.batch):
        ctx.upload_scan(i, base[order[i + 1]])
    ctx.set_pose_guess(0, args.batch, guesses)
    ctx.synchronize()
    if args.calibrate:
        ctx._ck(ctx.lib.ll_debug_calibration_copy(ctx.h, C.c_ulonglong(1 << 30)))

    def step():
        ctx.hot_path(0, args.batch, None, vote=True)     # pose restarts from the stored guess, device-to-device

    # box calibration: a FIXED micro-run (the extract stage of the first 512 slots, three times, untimed warm-up first), HIP-event time
    # of the ring kernels -- the boxes of the pool differ by up to 30 % in exactly these kernels (docs/rounds/r03.md 12.3a), so a slow
    # draw shows here instead of looking like a regression of the headline
    calib_n = min(512, args.batch)
    ctx.extract(0, calib_n); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_read(reset=True)
    for _ in range(3):
        ctx.extract(0, calib_n)
    cprof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    box_calibration_ms = sum(cprof[k][0] / max(1, cprof[k][1]) for k in ("k_ring_pick", "k_ring_features") if k in cprof and cprof[k][1])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_read(reset=True)
    # With LIGHTLOAM_TWO_STREAM=1 the timed region runs k_build_grid and k_associate side by side on two streams (ll_set_two_stream; off by
    # default: measured, no gain), which the event profiler records as ONE interval.  Their own durations -- each kernel alone on the chip,
    # what a roofline figure is defined on -- then come from a short pass on one stream behind the timed region (not part of `value`).
    STAGE = "k_build_grid||k_associate"
    stage_two_stream_ms = None
    if STAGE in prof and prof[STAGE][1]:
        stage_two_stream_ms = prof[STAGE][0] / prof[STAGE][1]
        ctx.set_two_stream(False)
        step(); ctx.synchronize(); ctx.profile_read(reset=True)
        for _ in range(3):
            step()
        sprof = ctx.profile_read(reset=True)
        ctx.set_two_stream(True)
        for k in ("k_build_grid", "k_associate"):
            prof[k] = sprof[k]
        prof = {k: v for k, v in prof.items() if k != STAGE}
    ctx.profile_enable(False)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    # sanity on the result of the timed work (not timed): EVERY slot extracted (status 0),
    # found correspondences and solved to a finite pose; slots that hold the same (previous, current) pair of scans and the
    # same guess must give bit-identical poses (the batch cycles through `distinct`+1 scans)
    info = ctx.scan_info(0)
    ab = ctx.algorithmic_bytes(0, args.batch)
    # per-kernel share of SURVEY.md section 8d's algorithmic bytes (every array counted once, at the kernel that must touch
    # it), summed over the batch from the run's actual counts
    tot = dict(n_in=0, n=0, feat=0, lsharp=0, lflat=0, q=0, ne=0, np_=0, nsel=0)
    pose_of_pair = {}
    bad = []
    for i in range(args.batch):
        si = ctx.scan_info(i); pi = ctx.pair_info(i); pose = ctx.pose(i)
        if si.status != 0 or pi.n_edge <= 0 or pi.n_plane_selected <= 0 or not np.isfinite(pose).all():
            bad.append((i, si.status, pi.n_edge, pi.n_plane_selected))
        key = (order[i], order[i + 1])
        if pose_of_pair.setdefault(key, pose).tobytes() != pose.tobytes():
            bad.append((i, "pose differs from the slot with the same scan pair", key))
        tot["n_in"] += si.n_in; tot["n"] += si.n
        tot["feat"] += si.n_sharp + si.n_less_sharp + si.n_flat + si.n_less_flat
        tot["lsharp"] += si.n_less_sharp; tot["lflat"] += si.n_less_flat; tot["q"] += si.n_sharp + si.n_flat
        tot["ne"] += pi.n_edge; tot["np_"] += pi.n_plane; tot["nsel"] += pi.n_plane_selected
    if bad and not os.environ.get("LL_BENCH_TIMING_BUILD"):     # set only by tools/ that time deliberately incomplete builds
        raise SystemExit(f"bench self-check failed on rank {rank}: {len(bad)} slot(s), first {bad[:5]}")
    kernel_bytes = {
        "k_first_kept": 0.0, "k_offsets": 0.0, "k_gn_step": 0.0,
        "k_organize": 4.0 * args.input_stride * tot["n_in"] + 16.0 * tot["n"],   # read the raw scan, write laserCloud (one pass)
        "k_classify": 4.0 * args.input_stride * tot["n_in"],                            # (tile-parallel path of small calls) read the raw scan
        "k_scatter": 16.0 * tot["n"],                                # (tile-parallel path) write laserCloud
        # the ring stage is two launches since round 4 (ll_pick.hip, ll_features.hip)
        "k_ring_pick": 17.0 * tot["n"],                               # read laserCloud, write labels (+ 352 B of lists per ring)
        "k_ring_features": 16.0 * tot["n"] + 16.0 * tot["feat"],   # read laserCloud (again), write the four feature clouds
        "k_build_grid": 16.0 * (tot["lsharp"] + tot["lflat"]),       # read the target clouds once
        "k_associate": 16.0 * tot["q"] + 8.0 * tot["ne"] + 12.0 * tot["np_"],
        "k_vote": 32.0 * tot["np_"] + 8.0 * tot["nsel"],
        "k_normal_equations": 48.0 * tot["ne"] + 64.0 * tot["nsel"] + 216.0 * args.batch,
    }

    if rank == 0:
        total_scans = args.batch * world * args.steps
        value = total_scans / elapsed
        # dominant kernel by summed HIP-event time (events on the library's own stream)
        dom = max(prof, key=lambda k: prof[k][0])
        dom_ms, dom_launches = prof[dom]
        # launches of every kernel per step: one per chunk (the library's event pool holds ~800 steps; beyond that the
        # recorded launches are a prefix of the run, so per-step figures come from the per-launch average, not the sum)
        launches_per_step = 1 if args.chunk <= 0 else -(-args.batch // args.chunk)
        bytes_per_launch = kernel_bytes[dom] / launches_per_step
        avg_ms = dom_ms / max(1, dom_launches)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_note = traffic_from_profile(args, dom, launches_per_step)
        ms_of = {k: v[0] / v[1] * launches_per_step for k, v in prof.items() if v[1]}
        # the ring stage = what rounds 1-3 ran as ONE kernel: laserCloud counted once (17 n + 16 feat)
        stage_kernels = [k for k in ("k_ring_pick", "k_ring_features") if k in ms_of]
        stage_ms = sum(ms_of[k] for k in stage_kernels)
        stage_bytes = 17.0 * tot["n"] + 16.0 * tot["feat"]
        stage_gbps = stage_bytes / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else 0.0
        # which wall: vector / scalar instruction issue against bytes, from the stamped counter pass -- for EVERY kernel of the step, so
        # that whichever kernel dominates (the ring kernels on 64 rings, k_associate on 128) is read against its own counters
        all_kernels = [k for k in ms_of if kernel_bytes.get(k)]
        issue, issue_note = issue_from_profile(args, all_kernels)
        issue_out = {"source": issue_note}
        bound = "unknown (no counter pass of this code, workload and batch: %s)" % issue_note
        sr = stream_rates()
        if issue is not None:
            simd_slots = N_SIMD * CLOCK_GHZ * 1e9 / VALU_CYCLES                   # wave64 vector instructions per second, whole chip
            scalar_slots = (N_SIMD / 4) * CLOCK_GHZ * 1e9                          # one scalar unit per CU, one instruction per cycle
            for k in all_kernels:
                t = ms_of[k] / launches_per_step * 1e-3
                issue_out[k] = {"valu_wave_insts_per_launch": issue[k]["valu"], "salu_wave_insts_per_launch": issue[k]["salu"],
                                "lds_wave_insts_per_launch": issue[k].get("lds"),
                                "valu_busy": issue[k]["valu"] / (simd_slots * t), "salu_busy": issue[k]["salu"] / (scalar_slots * t),
                                "hbm_frac": kernel_bytes[k] / launches_per_step / t / 1e9 / HBM_PEAK_GBS}
            d = issue_out.get(dom)
            # the kernel answers to the LARGEST of: vector issue, scalar issue, and the bytes it actually moves across the L2 <-> fabric
            # boundary (counter traffic, not the algorithmic bytes: re-reads count) against what the chip streams (stream_rates(), else the peak)
            t_dom = avg_ms * 1e-3
            stream_ceiling = (sr.get("copy_GBps") or HBM_PEAK_GBS)
            traffic_frac = (traffic / t_dom / 1e9 / HBM_PEAK_GBS) if (traffic and t_dom > 0) else None
            traffic_of_stream = (traffic / t_dom / 1e9 / stream_ceiling) if (traffic and t_dom > 0) else None
            if d:
                byte_frac = max(d["hbm_frac"] * HBM_PEAK_GBS / stream_ceiling, traffic_of_stream or 0.0)
                if max(d["valu_busy"], d["salu_busy"]) > byte_frac:
                    bound = "issue (vector %.0f %%, scalar %.0f %% busy; bytes %.0f %% of the HBM peak)" % (100 * d["valu_busy"], 100 * d["salu_busy"], 100 * d["hbm_frac"])
                else:
                    bound = "hbm"
            issue_out["decision"] = {"valu_busy": d["valu_busy"] if d else None, "salu_busy": d["salu_busy"] if d else None,
                                     "algorithmic_bytes_frac_of_hbm_peak": d["hbm_frac"] if d else None,
                                     "counter_traffic_frac_of_hbm_peak": traffic_frac,
                                     "counter_traffic_frac_of_measured_stream_rate": traffic_of_stream,
                                     "rule": "bound = hbm when the counter traffic's (or the algorithmic bytes') fraction of the measured streaming rate "
                                             "(profiles/r05_stream_rate.json; the 8 TB/s peak without it) is the largest of the four, else issue; "
                                             "unknown without a counter pass of this code",
                                     "caveat": "the label names the LARGEST of the four fractions, not a saturated unit: below ~0.9 nothing is.  docs/rounds/r05.md 14.7: "
                                               "k_ring_features got ~1 % faster for 15 % fewer bytes and again for 31 % fewer vector instructions; fewer "
                                               "barriers bought 1-3 %"}
            issue_out["model"] = ("valu_busy = SQ_INSTS_VALU / (%d SIMDs x %.1f GHz / %.1f cycles x launch time); salu_busy = SQ_INSTS_SALU / "
                                  "(%d scalar units x %.1f GHz x launch time)" % (N_SIMD, CLOCK_GHZ, VALU_CYCLES, N_SIMD // 4, CLOCK_GHZ))
        whole_alg_bytes_per_step = ab["ext"] + ab["assoc"] + ab["vote"] + ab["rj"]
        whole_alg_gbps = whole_alg_bytes_per_step * args.steps / elapsed / 1e9
        counter_bytes_per_step = all_traffic_from_profile(args, all_kernels, launches_per_step)
        out = {
            "metric": metric_name(args),
            "value": value, "unit": "scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (features, association, vote) + f64 (residuals, Jacobians, normal equations)",
            "data": "synthetic",
            "config": {"workload": workload_name(args) + ", feature extraction + graph-match + one GN iteration per scan pair, inputs resident in HBM",
                       "workload_id": args.workload, "max_ring_points": args.max_ring_points or 2304,
                       "scans_per_gpu_per_step": args.batch, "chunk": args.chunk or args.batch,
                       "distinct_scans": args.distinct + 1, "distinct_scan_pairs": len(pose_of_pair),
                       "box_calibration_ms": box_calibration_ms,
                       "box_calibration": "HIP-event time of the ring kernels over the first %d slots (fixed micro-run before the timed region)" % calib_n,
                       "ring_pipeline": "k_ring_pick + k_ring_features; ring-strided less-flat cloud, no hand-over between the rings of a scan",
                       "self_check": "every slot: status 0, correspondences > 0, finite pose; equal scan pairs -> bit-identical poses",
                       "points_per_scan_in": int(info.n_in), "points_per_scan_kept": int(info.n),
                       "input_bytes_per_point": 4 * args.input_stride,
                       "association": "k_associate in one traversal (per-ring minima collected while the nearest neighbour is searched)",
                       "parallelism": f"scan-parallel x{world}, no data-path collective"},
            "roofline": {"bound": bound, "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "traffic_GBps": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms > 0) else None,
                         "frac_of_measured_copy": achieved / (sr.get("copy_GBps") or HBM_COPY_GBS), "measured_copy_GBps": sr.get("copy_GBps") or HBM_COPY_GBS,
                         "measured_stream": sr,
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "whole_path_algorithmic_GBps": whole_alg_gbps,
                         # comparable across rounds (BASELINE.md's own definition: B x scans/s / peak, the whole path, every array counted once)
                         "frac_whole_path": whole_alg_gbps / HBM_PEAK_GBS,
                         # sum of the kernels' counter bytes / sum of the algorithmic bytes (1.0 = nothing re-read); null without a counter pass of this code
                         "wasted_traffic": (counter_bytes_per_step / whole_alg_bytes_per_step) if counter_bytes_per_step else None,
                         "issue": issue_out,
                         "ring_stage": {"kernels": stage_kernels, "ms_per_step": stage_ms, "algorithmic_bytes_per_step": stage_bytes,
                                        "achieved": stage_gbps, "frac": stage_gbps / HBM_PEAK_GBS,
                                        "note": "the stage rounds 1-3 ran as one kernel: laserCloud counted once (17 n + 16 features); each kernel's own "
                                                "figure above counts what that launch must read and write"},
                         "kernel_ms_per_step": ms_of,
                         "association_stage": None if stage_two_stream_ms is None else {
                             "schedule": "timed region: k_build_grid of one quarter of the batch beside k_associate of the quarter before it, two HIP "
                                         "streams ordered by events (LIGHTLOAM_TWO_STREAM=1 / ll_set_two_stream); kernel_ms_per_step's k_build_grid / "
                                         "k_associate: each kernel alone, from 3 one-stream steps behind the timed region",
                             "two_stream_ms_per_step": stage_two_stream_ms * launches_per_step,
                             "one_stream_ms_per_step": ms_of.get("k_build_grid", 0.0) + ms_of.get("k_associate", 0.0),
                             "sum_of_kernel_ms_equals_ms_per_step_with": "the two-stream figure in place of the two kernels' own"},
                         "kernel_algorithmic_GBps": {k: kernel_bytes[k] / (v[0] / v[1] * launches_per_step * 1e-3) / 1e9
                                                     for k, v in prof.items() if v[1] and kernel_bytes.get(k)}},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
    # RCCL sanity on every run (also N = 1): the 28-double all-reduce of the row-parallel mode (21 + 6 + 1 unique values of
    # JtJ, Jtr, cost) on a device buffer, checked against the closed form and timed.  Not part of `value`.
    rccl = rccl_check(torch, dist, world, rank, local_rank) if args.backend == "nccl" else {"rccl_world": None, "backend": "gloo (RCCL not exercised)"}
    ctx.close()
    import torch.distributed as tdist
    if tdist.is_initialized():
        tdist.barrier()
        tdist.destroy_process_group()
    if rank == 0:                                    # last, so that nothing (RCCL's banner) lands behind the JSON line
        out.update(rccl)
        emit(out)


if __name__ == "__main__":
    main()
