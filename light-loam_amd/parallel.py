"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl") on the GPU box, gloo in tests.

Two ways the hot path spreads over the GPUs of a node:
  * scan-parallel  -- scans (or scan pairs) are independent work items: block-partition them over ranks, no
                      data-path collective at all; only the timing barrier / max-over-ranks in bench.py.
  * tile-parallel  -- the MAP of laserMapping is split: every rank keeps the cubes it owns, searches its own points for
                      the five nearest of every stack point, and the candidates (100 B per stack point and rank) are
                      all-gathered; every rank then merges them and solves the same small problem.
  * row-parallel   -- the residual rows of ONE scan pair are split over ranks (laserMapping-size problems); every
                      Gauss-Newton iteration ends with an all-reduce of the 28 unique doubles of the normal equations
                      (21 upper-triangular entries of J^T J, 6 of J^T r, the cost) over xGMI: 224 bytes, latency-bound.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous block partition of n_items over `world` ranks: (first, count) of `rank`; sizes differ by <= 1."""
    base, rem = divmod(int(n_items), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


_TRIU = np.triu_indices(6)


def pack_normal_equations(H, g, cost):
    """6x6 symmetric H, g[6], cost -> 28 doubles (the payload of the all-reduce)."""
    H = np.asarray(H, np.float64); g = np.asarray(g, np.float64)
    return np.concatenate([H[_TRIU], g, [float(cost)]])


def unpack_normal_equations(v):
    v = np.asarray(v, np.float64)
    H = np.zeros((6, 6)); H[_TRIU] = v[:21]
    H = H + np.triu(H, 1).T
    return H, v[21:27].copy(), float(v[27])


def allreduce_normal_equations(H, g, cost, group=None, device=None):
    """Sum the partial normal equations of all ranks (RCCL all-reduce of 28 f64 on GPU tensors, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    buf = torch.from_numpy(pack_normal_equations(H, g, cost))
    if device is not None:
        buf = buf.to(device)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return unpack_normal_equations(buf.cpu().numpy())


def max_over_ranks(value, group=None, device=None):
    """Elapsed time of the slowest rank (bench.py's timing contract)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def map_optimize_row_parallel(m, pose_w, n_outer=2, max_num_iterations=4, opt=None, group=None, device=None):
    """laserMapping's optimisation (api.Map.optimize) with the scan's stack points split over the ranks of `group`:
    every rank holds the same map and ITS slice of the stack clouds in `m`; per evaluation one all-reduce of the
    normal equations (44 doubles; RCCL when `device` is a GPU device, gloo on CPU tensors), identical LM state on all
    ranks.  Returns the optimised pose (the same on every rank); the guess itself when the map is below laserMapping.cpp:1822's
    sizes (every rank holds the same map, so every rank takes that exit before the first collective)."""
    import torch
    import torch.distributed as dist

    nc, ns = m.map_sizes()
    if not (nc > 10 and ns > 50):
        return np.ascontiguousarray(pose_w, np.float64).copy()

    def reduced():
        buf = torch.from_numpy(m.evaluate())
        if device is not None:
            buf = buf.to(device)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        return buf.cpu().numpy()

    if opt is not None:
        max_num_iterations = opt.max_num_iterations
    m.set_pose(pose_w)
    for _ in range(n_outer):
        m.associate()
        m.lm_begin(reduced(), opt)
        for _ in range(max_num_iterations):
            m.lm_propose(opt)
            m.lm_accept(reduced(), opt)
    return m.pose()


def _all_gather_np(a, group=None, device=None):
    """numpy array -> [world, ...] stacked over the ranks of `group` (RCCL when `device` is a GPU device, else gloo)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(a))
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return np.stack([o.cpu().numpy() for o in out])


def _lm_all_reduced(m, max_num_iterations, opt, group, device):
    """One ceres::Solve restated with the normal equations summed over the ranks (every rank evaluates its row shard)."""
    import torch
    import torch.distributed as dist

    def reduced():
        buf = torch.from_numpy(m.evaluate())
        if device is not None:
            buf = buf.to(device)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        return buf.cpu().numpy()

    m.lm_begin(reduced(), opt)
    for _ in range(max_num_iterations):
        m.lm_propose(opt)
        m.lm_accept(reduced(), opt)


def map_optimize_tile_parallel(m, pose_w, n_stack, n_map_total, n_outer=2, opt=None, group=None, device=None, gather=None,
                               row_parallel=False):
    """laserMapping's optimisation (laserMapping.cpp:1822-2095) with the MAP split over the ranks of `group`: `m` holds this
    rank's shard of the search clouds (api.Map.set_map + set_map_ids, or the inner map of a sharded api.CubeMap) and the
    whole scan.  Per outer iteration: K = 5 search on the local points, all-gather of the candidates, merge + line / plane
    fit, Levenberg-Marquardt on the (replicated) residual blocks.  n_map_total = (corner, surf) points of the whole map
    (the reference's > 10 / > 50 test, :1822).  `gather` replaces the torch.distributed all-gather (tests).
    Returns (pose, ran) -- bit-identical on every rank and to api.Map.optimize on the unsplit map.
    row_parallel=True (BASELINE config 4 to the letter): the Levenberg-Marquardt evaluations are split too -- every rank sums
    the residual blocks i with i % world == rank and the normal equations are all-reduced (RCCL over xGMI); the pose then
    agrees with one GPU to f64 summation-order rounding and is still identical on every rank."""
    pose = np.ascontiguousarray(pose_w, np.float64).copy()
    if not (n_map_total[0] > 10 and n_map_total[1] > 50):
        return pose, False
    if gather is None:
        gather = lambda a: _all_gather_np(a, group, device)
    if row_parallel:
        import torch.distributed as dist
        m.set_row_shard(dist.get_rank(group), dist.get_world_size(group))
    try:
        for _ in range(n_outer):
            cn, ci, sn, si = m.knn_partial(pose, n_stack)
            m.associate_merged(gather(cn), gather(ci), gather(sn), gather(si), pose)
            if row_parallel:
                _lm_all_reduced(m, 4 if opt is None else opt.max_num_iterations, opt, group, device)
                pose = m.pose()
            else:
                pose = m.solve(pose, opt)
    finally:
        if row_parallel:
            m.set_row_shard(0, 1)
    return pose, True


def cubemap_process_tile_parallel(cm, pose_w, corner_last, surf_last, opt=None, group=None, device=None, gather=None, row_parallel=False):
    """One laserMapping frame (api.CubeMap.process) on a cube map sharded with CubeMap.set_shard(rank, world): every rank
    gets the whole scan, keeps and searches only its cubes.  Collectives per frame: one all-gather of the two gathered
    cloud sizes and, per outer iteration, the candidate all-gather."""
    cm.prepare(np.asarray(pose_w, np.float64)[4:7], corner_last, surf_last)
    _, cnt = cm.info()
    if gather is None:
        gather = lambda a: _all_gather_np(a, group, device)
    tot = gather(np.array(cnt[:2], np.int64)).sum(axis=0)
    pose, ran = map_optimize_tile_parallel(cm.map(), pose_w, (cnt[2], cnt[3]), (int(tot[0]), int(tot[1])), 2, opt, group, device, gather, row_parallel)
    cm.update(pose)
    return pose, ran


# ------------------------------------------------------------------ device-resident collectives (RCCL on the GPU box)
class DeviceCollectives:
    """The collectives of the two mapping modes on DEVICE buffers, stream-ordered with the library's own HIP stream: the
    normal equations and the K-NN candidates never visit the host, and nothing synchronises until the caller reads the pose.
    torch owns the buffers; the library copies into / reads from them through raw pointers on ll_stream(ctx); the collective
    is issued with that stream current (ProcessGroupNCCL orders its internal stream against the current stream with events).
    Backend "nccl" is RCCL on ROCm; world size 1 is legal and still goes through RCCL."""

    def __init__(self, ctx, device_index, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.dev = torch.device("cuda", device_index)
        self.stream = torch.cuda.ExternalStream(ctx.stream, device=self.dev)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.neq = torch.zeros(44, dtype=torch.float64, device=self.dev)
        self._cand = {}
        self.n_allreduce = 0
        self.n_allgather = 0

    def reduced_neq(self, m):
        """this rank's normal equations (at the map's current device pose) summed over the ranks -> device pointer"""
        m.evaluate_dev(self.neq.data_ptr())
        with self.torch.cuda.stream(self.stream):
            self.dist.all_reduce(self.neq, op=self.dist.ReduceOp.SUM, group=self.group)
        self.n_allreduce += 1
        return self.neq.data_ptr()

    def candidates(self, n_stack):
        """(own, all) buffers of the four candidate arrays for (corner, surf) stack sizes n_stack; cached by size"""
        key = tuple(int(x) for x in n_stack)
        if key not in self._cand:
            t = self.torch
            nc, ns = key
            own = [t.empty(max(nc, 1) * 20, dtype=t.float32, device=self.dev), t.empty(max(nc, 1) * 5, dtype=t.int32, device=self.dev),
                   t.empty(max(ns, 1) * 20, dtype=t.float32, device=self.dev), t.empty(max(ns, 1) * 5, dtype=t.int32, device=self.dev)]
            al = [t.empty(self.world * o.numel(), dtype=o.dtype, device=self.dev) for o in own]
            self._cand = {key: (own, al)}                      # one size class alive at a time
        return self._cand[key]

    def gathered_candidates(self, m, n_stack):
        own, al = self.candidates(n_stack)
        m.knn_partial_dev(*[o.data_ptr() for o in own])
        with self.torch.cuda.stream(self.stream):
            for o, a in zip(own, al):
                self.dist.all_gather_into_tensor(a, o, group=self.group)
        self.n_allgather += 4
        return [a.data_ptr() for a in al]


def map_optimize_row_parallel_dev(m, coll, pose_w, n_outer=2, max_num_iterations=4, opt=None):
    """map_optimize_row_parallel with the 44-double all-reduce on device buffers: one host upload (the guess), one read-back
    (the result); everything between is enqueued."""
    if opt is not None:
        max_num_iterations = opt.max_num_iterations
    nc, ns = m.map_sizes()
    if not (nc > 10 and ns > 50):                               # laserMapping.cpp:1822, the same on every rank: no collective yet
        return np.ascontiguousarray(pose_w, np.float64).copy()
    m.set_pose(pose_w)
    for _ in range(n_outer):
        m.associate()                                           # at the device pose, asynchronous
        m.lm_begin_dev(coll.reduced_neq(m), opt)
        for _ in range(max_num_iterations):
            m.lm_propose_dev(opt)
            m.lm_accept_dev(coll.reduced_neq(m), opt)
    return m.pose()


def map_optimize_tile_parallel_dev(m, coll, pose_w, n_stack, n_map_total, n_outer=2, opt=None, row_parallel=False):
    """map_optimize_tile_parallel with the candidate all-gather (and, with row_parallel, the all-reduce of the normal
    equations) on device buffers.  Same results as the host-hopped version: the kernels are the same, only the transport differs."""
    pose = np.ascontiguousarray(pose_w, np.float64).copy()
    if not (n_map_total[0] > 10 and n_map_total[1] > 50):
        return pose, False
    max_it = 4 if opt is None else opt.max_num_iterations
    if row_parallel:
        m.set_row_shard(coll.rank, coll.world)
    try:
        m.set_pose(pose)
        nc, ns = int(n_stack[0]), int(n_stack[1])
        _, al = coll.candidates((nc, ns))
        for _ in range(n_outer):
            ptrs = coll.gathered_candidates(m, (nc, ns))
            # the all-gathered arrays are [world][max(n, 1)][5][..]: world consecutive parts of n rows when n > 0
            m.associate_merged_dev(coll.world, *ptrs)
            if row_parallel:
                m.lm_begin_dev(coll.reduced_neq(m), opt)
                for _ in range(max_it):
                    m.lm_propose_dev(opt)
                    m.lm_accept_dev(coll.reduced_neq(m), opt)
            else:
                m.solve_dev(opt)
        pose = m.pose()                                         # the frame's one synchronising read-back
    finally:
        if row_parallel:
            m.set_row_shard(0, 1)
    return pose, True


def cubemap_process_tile_parallel_dev(cm, coll, pose_w, corner_last, surf_last, opt=None, row_parallel=False, from_slot=None):
    """cubemap_process_tile_parallel with device-resident collectives (one small all-reduce of the two cloud sizes per frame
    still goes through the host: it decides control flow)."""
    torch, dist = coll.torch, coll.dist
    cm.prepare(np.asarray(pose_w, np.float64)[4:7], corner_last, surf_last)
    _, cnt = cm.info()
    tot = torch.tensor([cnt[0], cnt[1]], dtype=torch.int64, device=coll.dev)
    dist.all_reduce(tot, group=coll.group)
    tot = tot.cpu().numpy()
    pose, ran = map_optimize_tile_parallel_dev(cm.map(), coll, pose_w, (cnt[2], cnt[3]), (int(tot[0]), int(tot[1])), 2, opt, row_parallel)
    cm.update(pose)
    return pose, ran
