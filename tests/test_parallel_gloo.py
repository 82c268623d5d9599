"""The N>1 path on CPU: two processes over gloo shard the replica ids and all-reduce the
learning-curve statistics exactly as bench.py does over RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_replica_partition_is_contiguous_and_disjoint():
    from grl_amd import parallel
    world, n = 4, 6
    allseeds = np.concatenate([parallel.replica_seeds(r, world, n) for r in range(world)])
    assert list(allseeds) == list(range(1, 1 + world * n))
    assert list(parallel.replica_seeds(0, 1, 3, seed0=7)) == [7, 8, 9]


def test_every_workload_partitions_into_disjoint_contiguous_shares():
    """bench.py's workloads (= BASELINE.json's configurations) on N ranks: per experiment graph the ranks' replica ids are
    contiguous, disjoint and cover 0 .. N*share*n - 1; the composite configs[3] puts BOTH halves on every rank."""
    from grl_amd import parallel
    for workload, parts in parallel.WORKLOAD_PARTS.items():
        for world, n in ((1, 8), (2, 8), (8, 16)):
            got = {}
            for rank in range(world):
                shares = parallel.partition(workload, rank, world, n)
                assert [g for g, _ in shares] == [g for g, _ in parts]
                assert sum(len(s) for _, s in shares) == n                  # every rank runs n replicas (weak scaling)
                for graph, seeds in shares:
                    assert list(np.diff(seeds)) == [1] * (len(seeds) - 1)
                    got.setdefault(graph, []).append(seeds)
            for graph, share in parts:
                allseeds = np.concatenate(got[graph])
                assert list(allseeds) == list(range(1, 1 + int(world * n * share))), (workload, graph, world)
    halves = parallel.partition("acrobot_walker", 3, 8, 8192)
    assert [(g, len(s), int(s[0])) for g, s in halves] == [("acrobot_q", 4096, 1 + 3 * 4096), ("compass_walker_q", 4096, 1 + 3 * 4096)]
    import pytest
    with pytest.raises(ValueError):
        parallel.partition("acrobot_walker", 0, 2, 7)                       # 7 replicas do not split into two halves
    with pytest.raises(KeyError):
        parallel.partition("no_such_workload", 0, 1, 4)


def test_curve_mean_stderr():
    from grl_amd import parallel
    r = np.array([[1.0, 2.0, 3.0, 6.0], [10.0, 10.0, 10.0, 10.0]])          # [rows][replicas]
    stats = np.stack([r.sum(1), (r ** 2).sum(1), np.full(2, 4.0)], axis=1)
    mean, se = parallel.curve_mean_stderr(stats)
    np.testing.assert_allclose(mean, r.mean(1))
    np.testing.assert_allclose(se, r.std(1, ddof=1) / 2.0, atol=1e-12)


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch
    from grl_amd import parallel
    rank, local_rank, world = parallel.init_distributed("gloo")
    assert world == 2
    n, rows = 5, 7
    seeds = parallel.replica_seeds(rank, world, n)
    # stand-in for grlx_curve_stats on this rank's replicas: returns are a function of the seed
    ret = np.stack([-(seeds.astype(np.float64) * (k + 1)) for k in range(rows)])       # [rows][n]
    stats = torch.tensor(np.stack([ret.sum(1), (ret ** 2).sum(1), np.full(rows, float(n))], axis=1))
    parallel.reduce_curve(stats, world)
    t = parallel.max_over_ranks(1.0 + rank, world)
    # the composite workload (BASELINE configs[3]): two curves per rank, one all-reduce of [2][rows][3]; env-steps summed over ranks
    halves = parallel.partition("acrobot_walker", rank, world, 2 * n)
    two = torch.tensor(np.stack([np.stack([-(s.astype(np.float64) * (k + 1)).sum() * np.array([1.0, 0.0, 0.0]) + np.array([0.0, 0.0, float(len(s))])
                                           for k in range(rows)]) for _, s in halves]))
    parallel.reduce_curve(two, world)
    steps = parallel.sum_over_ranks([100.0 * (rank + 1), 7.0], world)
    if rank == 0:
        allseeds = np.arange(1, 1 + world * n, dtype=np.float64)
        full = np.stack([-(allseeds * (k + 1)) for k in range(rows)])
        want = np.stack([full.sum(1), (full ** 2).sum(1), np.full(rows, float(world * n))], axis=1)
        np.testing.assert_allclose(stats.numpy(), want, rtol=1e-14)
        assert t == 2.0
        assert two.shape == (2, rows, 3) and (two[:, :, 2] == world * n).all()
        np.testing.assert_allclose(two[0, :, 0].numpy(), full.sum(1), rtol=1e-14)     # each half covers ids 0 .. world*n - 1
        np.testing.assert_allclose(two[1, :, 0].numpy(), full.sum(1), rtol=1e-14)
        assert steps == [300.0, 14.0]
        print("GLOO-OK")
    torch.distributed.destroy_process_group()
""")


def test_two_rank_gloo_reduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "GLOO-OK" in res.stdout
