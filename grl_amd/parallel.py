"""Multi-GPU layer: replicas are independent clones (the reference's experiment/multi,
base/src/experiments/multi.cpp:44-75), so the path shards with NO data-path collective:
rank g of G owns the contiguous replica ids [g*n, (g+1)*n) (weak scaling, n per GPU).
The ONLY collective is one all-reduce(SUM) of the learning-curve statistics
double[rows][3] = {sum r, sum r^2, count} at the end of a run (RCCL over xGMI on GPUs,
gloo in the CPU tests); ~4.3 KB for 181 rows, i.e. latency-bound, never bandwidth-bound.

Every workload of bench.py (= every configuration of BASELINE.json) shards through `partition`:
  pendulum_sarsa, cart_pole_ac, acrobot_q, compass_walker_q, pendulum_fqi_ann -- one experiment graph, rank g runs
      the replica ids [g*n, (g+1)*n) of it;
  acrobot_walker (BASELINE configs[3]: 65536 rollouts = 32768 acrobot + 32768 compass walker over 8 GPUs) -- BOTH
      halves on every rank: rank g runs acrobot ids [g*n/2, (g+1)*n/2) and walker ids [g*n/2, (g+1)*n/2) as two
      contexts on two HIP streams (equal work per rank, whatever the two environments cost relative to each other;
      the alternative -- acrobot on the first N/2 ranks, walkers on the rest -- leaves half the GPUs waiting for the
      walkers, which cost about twice as much per env-step).
"""
import math
import os

import numpy as np

# workload -> [(experiment graph, share of the rank's replicas)]
WORKLOAD_PARTS = {
    "pendulum_sarsa": [("pendulum_sarsa", 1.0)],
    "cart_pole_ac": [("cart_pole_ac", 1.0)],
    "acrobot_q": [("acrobot_q", 1.0)],
    "compass_walker_q": [("compass_walker_q", 1.0)],
    "acrobot_walker": [("acrobot_q", 0.5), ("compass_walker_q", 0.5)],
    "acrobot_walker_x2": [("acrobot_q", 0.5), ("compass_walker_q", 0.5)],
    "pendulum_fqi_ann": [("pendulum_fqi_ann", 1.0)],
}


def replica_seeds(rank: int, world: int, n_per_rank: int, seed0: int = 1) -> np.ndarray:
    """Seeds of the replicas rank `rank` owns: replica id r (global) is seeded srand48(seed0 + r)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    first = rank * n_per_rank
    return seed0 + first + np.arange(n_per_rank, dtype=np.int64)


def partition(workload: str, rank: int, world: int, n_per_rank: int, seed0: int = 1):
    """What rank `rank` of `world` runs of `workload` with `n_per_rank` replicas per GPU: a list of
    (experiment graph, seeds).  Over all ranks every graph's replica ids 0 .. world*share*n_per_rank - 1 are
    covered exactly once, contiguously per rank; replica id r of a graph is seeded seed0 + r."""
    if workload not in WORKLOAD_PARTS:
        raise KeyError("unknown workload " + workload)
    out = []
    for graph, share in WORKLOAD_PARTS[workload]:
        n = int(round(n_per_rank * share))
        if n < 1 or abs(n - n_per_rank * share) > 1e-9:
            raise ValueError(f"{workload}: {n_per_rank} replicas per rank do not split into shares of {share}")
        out.append((graph, replica_seeds(rank, world, n, seed0)))
    return out


def init_distributed(backend: str = "nccl"):
    """Join the job launched by torch.distributed.run (one process per GPU).  Returns
    (rank, local_rank, world).  Single-process runs need no initialisation."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def reduce_curve(stats, world: int):
    """All-reduce(SUM) of this rank's [rows][3] statistics tensor in place (no-op for world 1)."""
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return stats


def curve_mean_stderr(stats):
    """mean and standard error of the test return per row from the reduced statistics."""
    s = np.asarray(stats, dtype=np.float64)
    n = s[:, 2]
    mean = s[:, 0] / n
    var = np.maximum(s[:, 1] / n - mean * mean, 0.0) * n / np.maximum(n - 1, 1)
    return mean, np.sqrt(var / n)


def max_over_ranks(value: float, world: int, device=None) -> float:
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, world: int, device=None):
    """Element-wise sum of a short list of numbers over the ranks (bookkeeping outside the timed region:
    env-steps counted by each device)."""
    if world <= 1:
        return [float(v) for v in values]
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]
