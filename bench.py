#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused HIP rollout path (BASELINE.json metric).

Workload (BASELINE.json configs[1]): pendulum swing-up SARSA(lambda) tile coding,
4096 independent-seed replicas per GPU (replica r seeded srand48(1+r), weights
U(0,1) from the replica's own LCG stream -- synthetic inputs only).  One "step" is
one launch of the hot path: every replica advances by 11 trials (10 learning
episodes + 1 greedy test episode = 1100 env-steps).  Replica state, weight tables
and RNG streams are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Replicas shard across ranks with no data-path
collective (weak scaling: 4096 replicas per GPU); the only collective is the final
RCCL all-reduce of the learning-curve statistics [rows][3], inside the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REPLICAS_PER_GPU = 4096
TRIALS_PER_STEP = 11                      # test_interval 10 => 10 learning + 1 test episode
STEPS_PER_EPISODE = 100                   # control step 0.03 s, timeout 2.99 s
LEARN_STEPS_PER_STEP = 10 * STEPS_PER_EPISODE
TEST_STEPS_PER_STEP = 1 * STEPS_PER_EPISODE
# Algorithmic bytes (SURVEY.md section 8d / BASELINE.md section 5, restated in DESIGN.md):
BYTES_PER_LEARN_STEP = 2228               # 768 B of weight reads + 16 B x 91.2 read-modify-writes
BYTES_PER_TEST_STEP = 384                 # A*T 8-byte reads
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E peak 8 TB/s


def measured_traffic(n_replicas: int):
    """HBM-side bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run
    inside this process); only valid for the configuration it was measured on, else None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        if t.get("replicas") == n_replicas and t.get("trials_per_launch") == TRIALS_PER_STEP:
            return float(t["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_baseline(budget_s: float = 12.0):
    """The oracle (validated against the reference's golden curve) on ONE host core:
    replica seed 1, the same trial mix, for about `budget_s` seconds of CPU work."""
    from tests import oracle_binding as ob
    e = ob.Experiment(ob.pendulum_sarsa_spec(math=ob.MATH_LIBM), seed=1)      # libm = the reference's own arithmetic
    chunk = 110 * 5
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        e.run(chunk)
        steps += chunk * STEPS_PER_EPISODE
    dt = time.perf_counter() - t0
    out = {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"oracle (C restatement of grl's scalar path, libm arithmetic), 1 replica seed 1, "
                     f"{steps // STEPS_PER_EPISODE} trials = {steps} env-steps in {dt:.1f} s, weight init excluded"}
    # the same code on all host cores this job may use, one replica per core (the reference's
    # experiment/multi: one thread per clone); spawned processes, they never touch the GPU
    cores = min(len(os.sched_getaffinity(0)), 16)
    if cores > 1:
        import subprocess
        trials = max(110, int(steps / dt * 6.0 / STEPS_PER_EPISODE))          # about 6 s per core
        code = "import sys; from tests import oracle_binding as ob; s, t = ob.timed_run((int(sys.argv[1]), int(sys.argv[2]))); print(s, t)"
        root = os.path.dirname(os.path.abspath(__file__))
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-c", code, str(seed), str(trials)], cwd=root, stdout=subprocess.PIPE, text=True)
                 for seed in range(1, cores + 1)]
        res = []
        for pr in procs:
            line = pr.communicate(timeout=300)[0].split()
            if pr.returncode == 0 and len(line) == 2:
                res.append((int(line[0]), float(line[1])))
        wall = time.perf_counter() - t0
        if len(res) == cores:
            busy = max(r[1] for r in res)
            out["all_cores"] = {"value": sum(r[0] for r in res) / busy, "unit": "env-steps/s", "cores": cores,
                                "sample": f"{cores} processes x {trials} trials, slowest {busy:.1f} s (wall {wall:.1f} s incl. start-up and weight init)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--replicas", type=int, default=REPLICAS_PER_GPU, help="replicas per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--table-log2", type=int, default=17)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import numpy as np
    import torch
    import torch.distributed as dist
    import grl_amd
    from grl_amd import parallel

    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    parallel.init_distributed(args.backend)

    n = args.replicas
    total_steps = args.steps + args.warmup
    rows_total = total_steps                                   # one test row per step
    cfg = grl_amd.pendulum_sarsa_config(n, table_log2_capacity=args.table_log2, max_rows=rows_total + 1)
    seeds = parallel.replica_seeds(rank, world, n)             # contiguous partition of replica ids
    runner = grl_amd.Runner(cfg, seeds)
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream
    curve = torch.zeros((rows_total, 3), dtype=torch.float64, device="cuda")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner.run(TRIALS_PER_STEP, sptr)
    runner.sync(sptr)
    # warm the collective too (same shape and dtype as the timed one): communicator set-up and the first
    # launch of the RCCL kernel belong to start-up, not to the job
    parallel.reduce_curve(torch.zeros_like(curve), world)

    # per-launch kernel time: HIP events on the stream the kernel is launched on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        runner.run(TRIALS_PER_STEP, sptr)
        ev[k][1].record(stream)
    # the job's only collective: learning-curve statistics over all replicas of all GPUs
    runner.curve_stats(curve.data_ptr(), 0, rows_total, sptr)
    parallel.reduce_curve(curve, world)
    barrier()
    elapsed = time.perf_counter() - t0
    # the measured launches must have been the production instantiation of this configuration
    if runner.last_kernel() != 2:
        raise SystemExit("bench.py: the headline configuration did not run its specialised kernel (grlx_last_kernel = %d)" % runner.last_kernel())
    runner.sync(sptr)                                          # raises on table overflow etc.

    elapsed = parallel.max_over_ranks(elapsed, world, device="cuda")

    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    env_steps_per_step = n * (LEARN_STEPS_PER_STEP + TEST_STEPS_PER_STEP)
    total_env_steps = env_steps_per_step * args.steps * world
    learn, test = runner.step_counts()
    assert learn == n * LEARN_STEPS_PER_STEP * total_steps and test == n * TEST_STEPS_PER_STEP * total_steps, (learn, test)

    if rank == 0:
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        alg_bytes = n * (LEARN_STEPS_PER_STEP * BYTES_PER_LEARN_STEP + TEST_STEPS_PER_STEP * BYTES_PER_TEST_STEP)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        mean_curve = (curve[:, 0] / curve[:, 2]).cpu().numpy()
        out = {
            "metric": "env-steps/sec (batched rollouts), pendulum SARSA-tc",
            "value": total_env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "pendulum swing-up SARSA(lambda) hashed tile coding (cfg/pendulum/sarsa_tc.yaml semantics), "
                                   f"{n} independent-seed replicas per GPU, 11 trials (1100 env-steps) per replica per step",
                       "replicas_per_gpu": n, "trials_per_step": TRIALS_PER_STEP, "env_steps_per_step": env_steps_per_step * world,
                       "tilings": 16, "memory": 8388608, "parallelism": f"replicas x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(n),
                         "kernel": "rollout_kernel<pendulum, 3 actions, SpecPendulumTc(SARSA), deferred update>", "kernel_ms_avg": avg_ms,
                         "algorithmic_bytes_per_launch": alg_bytes},
            "mean_test_return_first_last": [float(mean_curve[0]), float(mean_curve[-1])],
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    runner.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
