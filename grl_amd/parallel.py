"""Multi-GPU layer: replicas are independent clones (the reference's experiment/multi,
base/src/experiments/multi.cpp:44-75), so the path shards with NO data-path collective:
rank g of G owns the contiguous replica ids [g*n, (g+1)*n) (weak scaling, n per GPU).
The ONLY collective is one all-reduce(SUM) of the learning-curve statistics
double[rows][3] = {sum r, sum r^2, count} at the end of a run (RCCL over xGMI on GPUs,
gloo in the CPU tests); ~4.3 KB for 181 rows, i.e. latency-bound, never bandwidth-bound.
"""
import math
import os

import numpy as np


def replica_seeds(rank: int, world: int, n_per_rank: int, seed0: int = 1) -> np.ndarray:
    """Seeds of the replicas rank `rank` owns: replica id r (global) is seeded srand48(seed0 + r)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    first = rank * n_per_rank
    return seed0 + first + np.arange(n_per_rank, dtype=np.int64)


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
