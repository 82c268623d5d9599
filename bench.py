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


def measured_pmc(key: str, n_replicas: int, trials_per_launch: int):
    """What the PMC passes committed under profiles/ measured for this workload (rocprofv3 cannot run inside
    this process): HBM-side bytes per launch and the issue share of the wave cycles.  Only valid for the
    configuration it was measured on, else (None, None).  Written by tools/pmc_to_json.py."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)["workloads"][key]
        if t.get("replicas") == n_replicas and t.get("trials_per_launch") == trials_per_launch:
            return float(t["hbm_bytes_per_launch"]), t.get("issue")
    except (OSError, ValueError, KeyError, TypeError):
        pass
    return None, None


# ---- the other single-GPU configurations of BASELINE.json (configs[2], per-GPU share of configs[3]) ------------
# Reported under "secondary" of the same JSON line, N = 1 only.  Algorithmic bytes per step: the figure of
# SURVEY.md 8(d) for the Q agents (3 actions, 16 tilings: 768 B of reads + 16 B x 91.2 read-modify-writes = 2228 B per
# learning step, 384 B per test step); actor-critic: 512 B of reads + 16 B x (16 + 16 + 75) = 2224 B
# per learning step, 128 B per test step.
SECONDARY = [
    dict(key="cart_pole_ac", replicas=16384, trials=11, steps=5, warmup=1, bytes_learn=2224, bytes_test=128,
         workload="cart-pole swing-up actor-critic, two tile-coded tables (cfg/cart_pole/ac_tc.yaml), 16384 replicas, 11 trials (2200 env-steps) per replica per step",
         kernel="rollout_ac_wide_kernel<cart_pole, 8 replicas per wave, SpecCartPoleAc, deferred update>", want_kernel=2),
    dict(key="acrobot_q", replicas=8192, trials=32, steps=5, warmup=1, bytes_learn=2228, bytes_test=384,
         workload="acrobot balancing Q-learning tile coding (agent block of cfg/pendulum/q_tc.yaml), 8192 replicas (per-GPU share of BASELINE configs[3]), 32 trials per replica per step",
         kernel="rollout_wide_kernel<acrobot, 3 actions, 8 replicas per wave, SpecAcrobotQ, deferred update>", want_kernel=2),
    dict(key="compass_walker_q", replicas=8192, trials=32, steps=5, warmup=1, bytes_learn=2228, bytes_test=384,
         workload="compass walker Q-learning tile coding (cfg/compass_walker/qlearning_walk.yaml), 8192 replicas (per-GPU share of BASELINE configs[3]), 32 trials per replica per step",
         kernel="rollout_wide_kernel<compass_walker, 3 actions, 8 replicas per wave, SpecWalkerQ, deferred update>", want_kernel=2),
]


# The batch path (BASELINE.json configs[4], per-GPU share): tests/pendulum-fqi-ann.yaml scaled to 100,000 transitions per
# batch, 16 independent-seed replicas per GPU.  A step = one batch (100,000 new transitions, FQIPredictor::rebuild over the
# whole store: 10 iterations x 500 epochs, one greedy test trial); the first batch is the warm-up, the second (200,000
# stored transitions) is timed.  Unit of work: a sample-epoch (forward + backward pass of one stored transition);
# algorithmic FLOPs per sample-epoch for the 3-20-1 network, exp counted as one: 2*(3+1)*20 + 2*(20+1) (forward MACs)
# + 4*20 (logistic) + 3*20 (hidden deltas) + (3+1)*20 + 20 (gradient products) + 2 = 444.
FQI = dict(key="pendulum_fqi_ann", replicas=16, batch_size=100000, iterations=10, epochs=500, flops_per_sample_epoch=444,
           workload="pendulum fitted Q-iteration, 3-20-1 logistic network trained by RPROP (tests/pendulum-fqi-ann.yaml scaled: 100000 transitions per batch, "
                    "10 iterations x 500 epochs), 16 independent-seed replicas; step = the second batch (rebuild over 200000 stored transitions)",
           kernel="fqi_grad_kernel<20> (+ fqi_step_kernel, fqi_targets_kernel<20>)")
F64_PEAK_TFLOPS = 78.6                    # MI355X f64 vector = matrix rate: half the 157.3 TF f32 vector peak of MI355X_MICROARCH.md


def run_fqi(torch, no_cpu_baseline, replicas=None, batch_size=None, epochs=None):
    import numpy as np
    import grl_amd
    w = FQI
    R = replicas or w["replicas"]
    n = batch_size or w["batch_size"]
    ep = epochs or w["epochs"]
    cfg = grl_amd.pendulum_fqi_config(R, batch_size=n, iterations=w["iterations"], epochs=ep, max_batches=2)
    r = grl_amd.FqiRunner(cfg, np.arange(1, R + 1))
    stream = torch.cuda.current_stream()
    r.run_batch(stream.cuda_stream)
    r.sync(stream.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record(stream)
    r.run_batch(stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    r.sync(stream.cuda_stream)
    its = [r.info(k)["iterations"] for k in range(R)]
    returns = [float(r.rows(k, 2)[2][1]) for k in range(R)]
    r.close()
    sample_epochs = sum(its) * ep * 2 * n
    flops = sample_epochs * w["flops_per_sample_epoch"]
    ms = e0.elapsed_time(e1)
    out = {"workload": w["workload"].replace("100000 transitions", f"{n} transitions").replace("200000 stored", f"{2 * n} stored").replace("16 independent", f"{R} independent").replace("500 epochs", f"{ep} epochs"),
           "value": sample_epochs / elapsed, "unit": "sample-epochs/s", "steps": 1, "warmup": 1, "ms_per_step": 1e3 * elapsed,
           "replicas": R, "transitions_stored": 2 * n, "iterations_run": its, "epochs": ep, "dtype": "f64", "data": "synthetic",
           "mean_test_return": sum(returns) / len(returns), "parity": "unpinned by the reference (oracle/fqi.c D1-D4); HIP == oracle bit for bit",
           "roofline": {"bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / (ms * 1e-3) / 1e12 / F64_PEAK_TFLOPS, "traffic": None, "kernel": w["kernel"], "kernel_ms_total": ms,
                        "algorithmic_flops_per_step": flops,
                        "note": "f64 VALU (no MFMA: K = 3 and K = 20 contractions, DESIGN.md 4.3); the chain of 10000 tiny launches per rebuild is latency-bound"}}
    if not no_cpu_baseline:
        from tests import oracle_binding as ob
        e = ob.FqiExperiment(ob.pendulum_fqi_spec(math=ob.MATH_LIBM, sum_order=ob.SUM_SEQUENTIAL, batch_size=4000, iterations=4, epochs=100), seed=1)
        t0 = time.perf_counter()
        e.run_batch()
        dt = time.perf_counter() - t0
        se = e.info()["iterations"] * 100 * 4000
        e.close()
        out["cpu_baseline"] = {"value": se / dt, "unit": "sample-epochs/s", "cores": 1, "kind": "port",
                               "sample": f"oracle/fqi.c (libm, the reference's sample-order gradient sum), 1 replica, 4000 transitions x {se // 4000} epochs in {dt:.1f} s"}
    return out


def secondary_config(key, n):
    import grl_amd
    if key == "cart_pole_ac":
        return grl_amd.cart_pole_ac_config(n)
    if key == "acrobot_q":
        return grl_amd.acrobot_q_config(n)
    if key == "compass_walker_q":
        return grl_amd.compass_walker_q_config(n)
    raise KeyError(key)


def secondary_cpu_baseline(key, budget_s=4.0):
    """The oracle on ONE host core on the same experiment graph (libm arithmetic = the reference's own)."""
    from tests import configs, oracle_binding as ob
    make = {"cart_pole_ac": configs.cart_pole_ac, "acrobot_q": configs.acrobot, "compass_walker_q": configs.compass_walker}[key]
    _, spec = make(None, 1)
    spec.math = ob.MATH_LIBM
    e = ob.Experiment(spec, seed=1)
    trials = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        e.run(22)
        trials += 22
    dt = time.perf_counter() - t0
    st = e.stats()
    steps = int(st.learn_steps + st.test_steps)
    e.close()
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"oracle (C restatement of grl's scalar path, libm arithmetic), 1 replica seed 1, {trials} trials = {steps} env-steps in {dt:.1f} s, weight init excluded"}


def run_secondary(w, torch, no_cpu_baseline, replicas=None):
    """One secondary workload on the current GPU: `warmup` untimed launches, then `steps` timed launches of
    `trials` trials of every replica (HIP events per launch on the launch stream, wall clock around the lot).
    Episode lengths vary (absorbing states), so env-steps are counted by the device, not assumed."""
    import numpy as np
    import grl_amd
    n = replicas or w["replicas"]
    cfg = secondary_config(w["key"], n)
    cfg.max_rows = (w["steps"] + w["warmup"]) * w["trials"] + 1
    runner = grl_amd.Runner(cfg, np.arange(1, n + 1))
    stream = torch.cuda.current_stream()
    sptr = stream.cuda_stream
    for _ in range(w["warmup"]):
        runner.run(w["trials"], sptr)
    runner.sync(sptr)
    l0, t0s = runner.step_counts()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(w["steps"])]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(w["steps"]):
        ev[k][0].record(stream)
        runner.run(w["trials"], sptr)
        ev[k][1].record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    runner.sync(sptr)                                          # raises on table overflow etc.
    l1, t1s = runner.step_counts()
    variant = runner.last_kernel()
    rpw = runner.replicas_per_wave()
    runner.close()
    learn, test = l1 - l0, t1s - t0s
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    avg_ms = sum(kernel_ms) / len(kernel_ms)
    alg_bytes = (learn * w["bytes_learn"] + test * w["bytes_test"]) / w["steps"]
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    traffic, issue = measured_pmc(w["key"], n, w["trials"])
    out = {"workload": w["workload"], "value": (learn + test) / elapsed, "unit": "env-steps/s", "steps": w["steps"], "warmup": w["warmup"],
           "ms_per_step": 1e3 * elapsed / w["steps"], "replicas": n, "trials_per_step": w["trials"],
           "env_steps_per_step": (learn + test) / w["steps"], "learn_steps": learn, "test_steps": test, "dtype": "f64", "data": "synthetic",
           "last_kernel": variant, "kernel_is_expected_instantiation": variant == w["want_kernel"], "replicas_per_wave": rpw,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "issue": issue, "kernel": w["kernel"], "kernel_ms_avg": avg_ms,
                        "algorithmic_bytes_per_launch": alg_bytes}}
    if not no_cpu_baseline:
        out["cpu_baseline"] = secondary_cpu_baseline(w["key"])
    return out


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
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE.json configurations (cart-pole AC, acrobot, walker)")
    ap.add_argument("--no-fqi", action="store_true", help="skip the batch-path entry (its rebuild is ~20000 launches: keeps profiler output small)")
    ap.add_argument("--only", default="", help="profiling: run only this secondary workload (cart_pole_ac | acrobot_q | compass_walker_q) and print its entry")
    ap.add_argument("--secondary-replicas", type=int, default=0, help="tests: override the replica count of the secondary workloads")
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
    if args.only == FQI["key"]:
        print(json.dumps(run_fqi(torch, args.no_cpu_baseline, args.secondary_replicas or None)))
        return
    if args.only:
        w = [x for x in SECONDARY if x["key"] == args.only]
        if not w:
            raise SystemExit("--only: unknown workload " + args.only)
        print(json.dumps(run_secondary(w[0], torch, args.no_cpu_baseline, args.secondary_replicas or None)))
        return
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
        traffic, issue = measured_pmc("pendulum_sarsa", n, TRIALS_PER_STEP)
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
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "issue": issue,
                         "kernel": "rollout_kernel<pendulum, 3 actions, SpecPendulumTc(SARSA), deferred update>", "kernel_ms_avg": avg_ms,
                         "algorithmic_bytes_per_launch": alg_bytes},
            "mean_test_return_first_last": [float(mean_curve[0]), float(mean_curve[-1])],
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    runner.close()                                             # frees this workload's tables before the next one allocates
    if rank == 0:
        if world == 1 and not args.no_secondary:
            # the other single-GPU configurations BASELINE.json names, each timed the same way on this GPU
            out["secondary"] = [run_secondary(w, torch, args.no_cpu_baseline, args.secondary_replicas or None) for w in SECONDARY]
            if not args.no_fqi:
                out["secondary"].append(run_fqi(torch, args.no_cpu_baseline))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
