"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl") on the GPU box, gloo in tests.

Two ways the hot path spreads over the GPUs of a node:
  * scan-parallel  -- scans (or scan pairs) are independent work items: block-partition them over ranks, no
                      data-path collective at all; only the timing barrier / max-over-ranks in bench.py.
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
    ranks.  Returns the optimised pose (the same on every rank)."""
    import torch
    import torch.distributed as dist

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
