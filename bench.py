#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the fused HIP rollout path (BASELINE.json metric).

Headline workload (BASELINE.json configs[1]): pendulum swing-up SARSA(lambda) tile coding,
4096 independent-seed replicas per GPU (replica r seeded srand48(1+r), weights
U(0,1) from the replica's own LCG stream -- synthetic inputs only).  One "step" is
one launch of the hot path: every replica advances by 11 trials (10 learning
episodes + 1 greedy test episode = 1100 env-steps).  Replica state, weight tables
and RNG streams are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  EVERY workload (= every configuration of BASELINE.json) runs on any number of
ranks: replicas shard across ranks with no data-path collective (weak scaling, a fixed number of replicas per GPU,
grl_amd.parallel.partition); the only collective of a workload is the final RCCL all-reduce of its learning-curve
statistics [rows][3], inside the timed region.

  --workload pendulum_sarsa (default)   the headline line; the other workloads follow under "secondary", each sharded
                                        and timed the same way on the same ranks (--no-secondary / --no-fqi skip them)
  --workload cart_pole_ac               BASELINE configs[2]: 16384 replicas per GPU
  --workload acrobot_walker             BASELINE configs[3]: 8192 rollouts per GPU = 4096 acrobot + 4096 compass walker,
                                        both halves on every rank, two contexts on two HIP streams
  --workload acrobot_walker_x2          the same composite with twice the rollouts per GPU (8192 + 8192, 16 replicas per wave)
  --workload acrobot_q | compass_walker_q   one half of configs[3] alone, 8192 / 32768 replicas per GPU
  --workload pendulum_fqi_ann           BASELINE configs[4]: replicas only (16 independent-seed batch experiments per GPU)
"""
import argparse
import hashlib
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
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E peak 8 TB/s
F64_PEAK_TFLOPS = 78.6                    # MI355X f64 vector = matrix rate: half the 157.3 TF f32 vector peak of MI355X_MICROARCH.md

# Algorithmic bytes per env-step: 8 B per weight the algorithm reads + 16 B per read-modify-write.
#   headline: the declared figure of SURVEY.md 8(d) / BASELINE.md 5 (2228 B per learning step = 768 B of reads + 16 B x 91.2
#   read-modify-writes, the 20000-trial mean; 384 B per test step);
#   every other graph: counted by the oracle on the trials this file times (tools/algorithmic_bytes.py, mean of seeds 1-4,
#   libm arithmetic; the headline's own count in that regime is listed for comparison -- early learning updates longer traces).
#   The default run recounts them live in its cpu_baseline leg (same oracle run) and reports which figure it used.
ALGORITHMIC_BYTES = {
    "pendulum_sarsa": dict(learn=2228.0, test=384.0, reads_per_test=48, oracle_bench_regime=2629.8),
    "cart_pole_ac": dict(learn=3226.3, test=128.0, reads_per_test=16),
    "acrobot_q": dict(learn=2154.2, test=384.0, reads_per_test=48),
    "compass_walker_q": dict(learn=3446.6, test=384.0, reads_per_test=48),
}

# experiment graph -> how one GPU runs it
GRAPHS = {
    "pendulum_sarsa": dict(trials=TRIALS_PER_STEP, want_kernel=2, pmc_key="pendulum_sarsa",
                           kernel="rollout_served_kernel<3 actions, SpecPendulumTc(SARSA)> + env_server_kernel (co-resident pair, one timed launch; "
                                  "rollout_kernel<pendulum, 3 actions, SpecPendulumTc(SARSA), deferred update> with GRLX_ENV_SERVER=0)",
                           text="pendulum swing-up SARSA(lambda) hashed tile coding (cfg/pendulum/sarsa_tc.yaml semantics)"),
    "cart_pole_ac": dict(trials=11, want_kernel=2, pmc_key="cart_pole_ac",
                         kernel="rollout_ac_wide_kernel<cart_pole, 16 replicas per wave (four sub-batches, two parked in device memory), SpecCartPoleAc, deferred update>",
                         text="cart-pole swing-up actor-critic, two tile-coded tables (cfg/cart_pole/ac_tc.yaml)"),
    # absorbing environments: episodes of very different lengths, so a launch is bounded by a STEPS budget per replica -- the second bound
    # of the reference's own trial loop (experiment/online_learning:steps, online_learning.cpp:154; grlx_run_steps) -- instead of by a
    # trial count, which makes every launch wait for its slowest replica (A/B on one box, tools/steps_budget_ab.py: acrobot 246 -> 359 M,
    # walker 181 -> 217 M env-steps/s over the same total)
    "acrobot_q": dict(trials=0, budget=1100, want_kernel=2, pmc_key="acrobot_q",
                      kernel="rollout_wide_served_kernel<acrobot, SpecAcrobotQ> (8 replicas per wave, deferred update) + env_server_acrobot_pinned_kernel "
                             "(co-resident pair, one timed launch; rollout_wide_kernel<acrobot, 3 actions, 8 replicas per wave, SpecAcrobotQ> with GRLX_ENV_SERVER=0)",
                      text="acrobot balancing Q-learning tile coding (agent block of cfg/pendulum/q_tc.yaml)"),
    "compass_walker_q": dict(trials=0, budget=12200, want_kernel=2, pmc_key="compass_walker_q",
                             kernel="rollout_wide_kernel<compass_walker, 3 actions, 32 replicas per wave (eight sub-batches: one parked in LDS, two in registers, five in device memory), SpecWalkerQ, deferred update>",
                             text="compass walker Q-learning tile coding (cfg/compass_walker/qlearning_walk.yaml)"),
}

# workload -> replicas per GPU, default launches, which graph's kernel the roofline object describes
WORKLOADS = {
    "pendulum_sarsa": dict(replicas=REPLICAS_PER_GPU, steps=20, warmup=3, dominant="pendulum_sarsa", baseline_config=1),
    "cart_pole_ac": dict(replicas=16384, steps=5, warmup=1, dominant="cart_pole_ac", baseline_config=2),
    "acrobot_q": dict(replicas=8192, steps=5, warmup=1, dominant="acrobot_q", baseline_config=3),
    # 32 replicas per SIMD: EIGHT sub-batches per wave share one environment phase (A/B on one box, tools/wide16_ab.sh / wide32_ab.sh: 233 M at 8192
    # replicas with 8 per wave; 239 / 309 M at 16384 with 8 / 16 per wave; 311 / 364 M at 32768 with 16 / 32 per wave)
    "compass_walker_q": dict(replicas=32768, steps=4, warmup=1, dominant="compass_walker_q", baseline_config=3),
    # both halves on every rank, each as 512 waves of 8 replicas: together one wave per SIMD, both kernels resident for the whole launch
    # (the acrobot's waves beside their environment server's).
    # The acrobot's budget per launch is set so that its kernel lasts about as long as the walkers' (an acrobot step costs a twentieth of a
    # launch's walker work): otherwise its half of the chip idles for 95 % of every launch.
    "acrobot_walker": dict(replicas=8192, steps=5, warmup=1, dominant="compass_walker_q", baseline_config=3, replicas_per_wave=8,
                           budget={"acrobot_q": 36000}),
    # the same composite with TWICE BASELINE's rollouts per GPU (131072 over 8 GPUs): 8192 + 8192, 512 waves of 16 replicas each -- four
    # sub-batches per wave share one environment phase (round 4); sized for the GPU, not for the reference's rollout count
    "acrobot_walker_x2": dict(replicas=16384, steps=5, warmup=1, dominant="compass_walker_q", baseline_config=3, replicas_per_wave=16,
                              budget={"acrobot_q": 24000}),
}
SECONDARY_ORDER = ["cart_pole_ac", "acrobot_q", "compass_walker_q", "acrobot_walker", "acrobot_walker_x2"]

# The batch path (BASELINE.json configs[4], per-GPU share): tests/pendulum-fqi-ann.yaml scaled to 100,000 transitions per
# batch, 16 independent-seed replicas per GPU.  A step = one batch (100,000 new transitions, FQIPredictor::rebuild over the
# whole store: 10 iterations x 500 epochs, one greedy test trial); the first batch is the warm-up, the second (200,000
# stored transitions) is timed.  Unit of work: a sample-epoch (forward + backward pass of one stored transition);
# algorithmic FLOPs per sample-epoch for the 3-20-1 network, exp counted as one: 2*(3+1)*20 + 2*(20+1) (forward MACs)
# + 4*20 (logistic) + 3*20 (hidden deltas) + (3+1)*20 + 20 (gradient products) + 2 = 444.
FQI = dict(key="pendulum_fqi_ann", replicas=16, batch_size=100000, iterations=10, epochs=500, flops_per_sample_epoch=444,
           workload="pendulum fitted Q-iteration, 3-20-1 logistic network trained by RPROP (tests/pendulum-fqi-ann.yaml scaled: 100000 transitions per batch, "
                    "10 iterations x 500 epochs), 16 independent-seed replicas per GPU; step = the second batch (rebuild over 200000 stored transitions)",
           kernel="fqi_epoch_kernel<20> (+ fqi_targets_kernel<20>)")


def csrc_hash() -> str:
    """Identifies the device code a profile was taken on: sha256 over the sources of libgrlx.so."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "grl_amd", "csrc")
    for name in sorted(os.listdir(d)):
        p = os.path.join(d, name)
        if os.path.isfile(p) and name.endswith((".h", ".hip", ".cpp")):
            h.update(name.encode())
            with open(p, "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


# Bytes the environment server's mailboxes move per SERVED env-step, by the protocol (grl_amd/csrc/grlx_env_server.h, grlx_env_server_wide.h;
# 16-byte units {value, sequence}, device scope).  Stored: the candidates of all three actions + the command.  Loaded at least once: the
# candidate the sampler chose + the command (the server).  Repeated polls of a unit that has not arrived yet are NOT in these figures
# (a GRLX_ENV_SERVER_STATS build counts them: DESIGN.md 4.1g), nor are the S units a trial start writes and the server reads once per
# EPISODE (reported beside them).
MAILBOX = {
    "pendulum_sarsa": dict(units_per_candidate=5, state_units=3),          # {x0, x1, x2, obs0, reward}
    "acrobot_q": dict(units_per_candidate=5 + 1, state_units=5),           # {x[0..S), reward}, S = 5
    "compass_walker_q": dict(units_per_candidate=11 + 1, state_units=11),  # S = 11
}


def mailbox_bytes_per_step(graph: str):
    m = MAILBOX.get(graph)
    if not m:
        return None
    stored = 3 * m["units_per_candidate"] * 16 + 8
    loaded = m["units_per_candidate"] * 16 + 8
    return {"stored": stored, "loaded": loaded, "total": stored + loaded, "per_episode_not_included": 2 * m["state_units"] * 16}


def measured_pmc(key: str, n_replicas: int, trials_per_launch: int):
    """What the PMC passes committed under profiles/ measured for this workload (rocprofv3 cannot run inside this process):
    HBM-side bytes per launch and the issue share of the wave cycles.  Valid only for the configuration AND the device code
    it was measured on: a profile taken on other kernel sources is refused.  Returns (traffic, issue, traffic_source)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            doc = json.load(f)
        t = doc["workloads"][key]
    except (OSError, ValueError, KeyError, TypeError):
        return None, None, "none: profiles/pmc_traffic.json has no entry for " + key
    tag = f"profiles/{t.get('source', '?')}_pmc_raw.txt via profiles/pmc_traffic.json, csrc {doc.get('csrc_sha256', 'unrecorded')}"
    if doc.get("csrc_sha256") != csrc_hash():
        return None, None, f"stale, refused ({tag}; this build is csrc {csrc_hash()}): rerun tools/profile_passes.sh"
    if t.get("replicas") != n_replicas or t.get("trials_per_launch") != trials_per_launch:
        return None, None, f"other configuration, refused ({tag}: {t.get('replicas')} replicas x {t.get('trials_per_launch')} trials)"
    if t.get("note"):
        tag += "; " + t["note"]
    return float(t["hbm_bytes_per_launch"]), t.get("issue"), tag


def graph_config(graph, n, max_rows, table_log2=None, replicas_per_wave=0):
    import grl_amd
    make = {"pendulum_sarsa": grl_amd.pendulum_sarsa_config, "cart_pole_ac": grl_amd.cart_pole_ac_config,
            "acrobot_q": grl_amd.acrobot_q_config, "compass_walker_q": grl_amd.compass_walker_q_config}[graph]
    cfg = make(n)
    if graph in ("cart_pole_ac", "compass_walker_q"):
        cfg.table_log2_capacity = 18      # pre-sized for the whole run: no re-hash between the timed launches (the defaults, 2^16 / 2^17, grow)
    cfg.max_rows = max_rows
    if table_log2:
        cfg.table_log2_capacity = table_log2
    if replicas_per_wave:
        cfg.replicas_per_wave = replicas_per_wave
    return cfg


def oracle_spec(graph):
    from tests import configs
    make = {"pendulum_sarsa": configs.pendulum, "cart_pole_ac": configs.cart_pole_ac, "acrobot_q": configs.acrobot,
            "compass_walker_q": configs.compass_walker}[graph]
    return make(None, 1)[1]


_ORACLE_RUNS = {}


def oracle_cpu_baseline(graph, budget_s, trials_in_regime, steps_in_regime=0):
    key = (graph, budget_s, trials_in_regime, steps_in_regime)
    if key not in _ORACLE_RUNS:
        _ORACLE_RUNS[key] = _oracle_cpu_baseline(graph, budget_s, trials_in_regime, steps_in_regime)
    return _ORACLE_RUNS[key]


def _oracle_cpu_baseline(graph, budget_s, trials_in_regime, steps_in_regime):
    """The oracle (validated against the reference's golden files) on ONE host core, libm arithmetic = the reference's own:
    replica seed 1 for about `budget_s` seconds.  Also counts the algorithmic bytes per step on the first
    `trials_in_regime` trials (the ones bench.py times on the GPU)."""
    from tests import oracle_binding as ob
    spec = oracle_spec(graph)
    spec.math = ob.MATH_LIBM
    e = ob.Experiment(spec, seed=1)
    rpt = ALGORITHMIC_BYTES[graph]["reads_per_test"]
    t0 = time.perf_counter()
    if steps_in_regime:
        e.set_steps_budget(steps_in_regime)
        e.run(1 << 20)
        e.set_steps_budget(0)
    else:
        e.run(trials_in_regime)
    st = e.stats()
    trials_in_regime = trials_in_regime or e.trials_run()
    counted = dict(learn=(8.0 * (st.weight_reads - st.test_steps * rpt) + 16.0 * st.weight_rmws) / max(st.learn_steps, 1), test=8.0 * rpt,
                   source=f"oracle, seed 1, trials 1-{trials_in_regime} ({int(st.learn_steps)} learning steps)")
    trials = trials_in_regime
    while time.perf_counter() - t0 < budget_s:
        e.run(22)
        trials += 22
    dt = time.perf_counter() - t0
    st = e.stats()
    steps = int(st.learn_steps + st.test_steps)
    e.close()
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"oracle (C restatement of grl's scalar path, libm arithmetic), 1 replica seed 1, {trials} trials = {steps} env-steps in {dt:.1f} s, weight init excluded"}, counted


def all_cores_baseline(steps_per_s):
    """The same oracle on all host cores this job may use, one replica per core (the reference's experiment/multi:
    one thread per clone); spawned processes, they never touch the GPU."""
    import subprocess
    cores = min(len(os.sched_getaffinity(0)), 16)
    if cores <= 1:
        return None
    trials = max(110, int(steps_per_s * 6.0 / STEPS_PER_EPISODE))          # about 6 s per core
    code = "import sys; from tests import oracle_binding as ob; s, t = ob.timed_run((int(sys.argv[1]), int(sys.argv[2]))); print(s, t)"
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(seed), str(trials)], cwd=ROOT, stdout=subprocess.PIPE, text=True)
             for seed in range(1, cores + 1)]
    res = []
    for pr in procs:
        line = pr.communicate(timeout=300)[0].split()
        if pr.returncode == 0 and len(line) == 2:
            res.append((int(line[0]), float(line[1])))
    wall = time.perf_counter() - t0
    if len(res) != cores:
        return None
    busy = max(r[1] for r in res)
    return {"value": sum(r[0] for r in res) / busy, "unit": "env-steps/s", "cores": cores,
            "sample": f"{cores} processes x {trials} trials, slowest {busy:.1f} s (wall {wall:.1f} s incl. start-up and weight init)"}


class Dist:
    """The job as torch.distributed.run launched it (world 1: no process group)."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            if self.world == 1 and args.gpus > 1:
                raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
            raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {self.world}")

    def barrier(self, torch):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()


def run_rollout_workload(name, D, torch, steps, warmup, replicas=None, table_log2=None, cpu_baseline=True, full_cpu_baseline=False):
    """One workload on all ranks: `warmup` untimed launches, then EXACTLY `steps` timed launches of every context this rank
    owns, bracketed by barrier + synchronize, with the workload's one collective (all-reduce of the curve statistics) inside
    the timed region; MAX of the elapsed time over ranks.  HIP events per launch on the stream each kernel is launched on.
    Episode lengths vary (absorbing states), so env-steps are counted by the device and summed over ranks afterwards."""
    import grl_amd
    from grl_amd import parallel
    w = WORKLOADS[name]
    n_rank = replicas or w["replicas"]
    parts = parallel.partition(name, D.rank, D.world, n_rank)
    main_stream = torch.cuda.current_stream()
    ctx = []
    for k, (graph, seeds) in enumerate(parts):
        g = dict(GRAPHS[graph])
        if graph in w.get("budget", {}):
            g["budget"] = w["budget"][graph]
        # test_interval 10: one row per 11 trials; with a steps budget the replicas write rows at their own pace: the curve is reduced
        # over the first 16 rows (every replica has them after the first launches), the row arrays hold up to 1024
        rows_total = ((steps + warmup) * g["trials"]) // TRIALS_PER_STEP if g["trials"] else 16
        cfg = graph_config(graph, len(seeds), rows_total + 1 if g["trials"] else 1024, table_log2, w.get("replicas_per_wave", 0) if len(seeds) >= 8 else 0)
        stream = main_stream if len(parts) == 1 else torch.cuda.Stream()
        ctx.append(dict(graph=graph, g=g, seeds=seeds, runner=grl_amd.Runner(cfg, seeds), stream=stream, rows=rows_total,
                        ev=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]))
    rows_max = max(c["rows"] for c in ctx)
    curve = torch.zeros((len(ctx), rows_max, 3), dtype=torch.float64, device="cuda")

    def launch(c, k):
        """launch k (0-based, warm-up included) of context c: g["trials"] further trials, or whole trials up to a cumulative steps budget"""
        if c["g"]["trials"]:
            c["runner"].run(c["g"]["trials"], c["stream"].cuda_stream)
        else:
            c["runner"].run_steps(1 << 20, (k + 1) * c["g"]["budget"], c["stream"].cuda_stream)

    for c in ctx:
        for k in range(warmup):
            launch(c, k)
    for c in ctx:
        c["runner"].sync(c["stream"].cuda_stream)
        c["count0"] = c["runner"].step_counts()
    # warm the collective too (same shape and dtype as the timed one): communicator set-up and the first
    # launch of the RCCL kernel belong to start-up, not to the job
    parallel.reduce_curve(torch.zeros_like(curve), D.world)

    D.barrier(torch)
    t0 = time.perf_counter()
    # launch order within a step: contexts WITHOUT an environment server first.  The server's small waves fit beside the acrobot's rollout waves
    # (335 + 174 registers) but not beside the walker's (346): launched first they spread over all SIMDs and the walkers' waves wait for a SIMD
    # without one (composite of configs[3]: walker kernel 714 ms per launch with the acrobot first, 520 ms with the servers off)
    order = sorted(ctx, key=lambda c: 0 if c["graph"] == "compass_walker_q" else 1) if len(ctx) > 1 else ctx
    for k in range(steps):
        for c in order:
            c["ev"][k][0].record(c["stream"])
            launch(c, warmup + k)
            c["ev"][k][1].record(c["stream"])
    # the job's only collective: learning-curve statistics over all replicas of all GPUs
    for i, c in enumerate(ctx):
        c["runner"].curve_stats(curve[i].data_ptr(), 0, c["rows"], c["stream"].cuda_stream)
        if c["stream"] is not main_stream:
            main_stream.wait_stream(c["stream"])
    parallel.reduce_curve(curve, D.world)
    D.barrier(torch)
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, D.world, device="cuda")

    learn = test = 0
    per_graph = []
    for c in ctx:
        if c["runner"].last_kernel() != c["g"]["want_kernel"]:
            raise SystemExit(f"bench.py: {c['graph']} did not run its specialised kernel (grlx_last_kernel = {c['runner'].last_kernel()})")
        c["runner"].sync(c["stream"].cuda_stream)               # raises on table overflow etc.
        l1, t1 = c["runner"].step_counts()
        c["learn"], c["test"] = l1 - c["count0"][0], t1 - c["count0"][1]
        learn += c["learn"]
        test += c["test"]
        c["kernel_ms"] = [a.elapsed_time(b) for a, b in c["ev"]]
        c["rpw"] = c["runner"].replicas_per_wave()
        c["env_server"] = c["runner"].env_server_counts()       # (replicas served to the end of the last launch, replicas that fell back)
    all_learn, all_test = parallel.sum_over_ranks([learn, test], D.world, device="cuda")
    curve_host = curve.cpu().numpy()
    for c in ctx:
        c["runner"].close()                                     # frees this workload's tables before the next one allocates
    if D.rank != 0:
        return None

    out = {"workload": name, "value": (all_learn + all_test) / elapsed, "unit": "env-steps/s", "n_gpus": D.world, "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * elapsed / steps, "scaling": "weak", "dtype": "f64", "data": "synthetic",
           "replicas_per_gpu": n_rank, "parallelism": f"replicas x{D.world}" + (" (both halves on every rank, two streams)" if len(ctx) > 1 else ""),
           "env_steps_per_step": (all_learn + all_test) / steps, "learn_steps": int(all_learn), "test_steps": int(all_test),
           "baseline_config": f"BASELINE.json configs[{w['baseline_config']}]"}
    text = []
    for i, c in enumerate(ctx):
        g = c["g"]
        n = len(c["seeds"])
        per_step = f"{g['trials']} trials per replica per step" if g["trials"] else f"whole trials up to {g['budget']} further learning steps per replica per step (experiment/online_learning:steps)"
        text.append(f"{g['text']}, {n} independent-seed replicas per GPU, {per_step}")
        bytes_ = dict(ALGORITHMIC_BYTES[c["graph"]], source="committed figure (bench.py ALGORITHMIC_BYTES; tools/algorithmic_bytes.py)")
        cpu = None
        if cpu_baseline:
            cpu, counted = oracle_cpu_baseline(c["graph"], 12.0 if (full_cpu_baseline and i == 0) else 4.0, (steps + warmup) * g["trials"],
                                               (steps + warmup) * g.get("budget", 0))
            if c["graph"] != "pendulum_sarsa":                  # the headline keeps SURVEY's declared 2228 B
                bytes_.update(counted)
            else:
                bytes_["oracle_this_run"] = counted["learn"]
            if full_cpu_baseline and i == 0:
                ac = all_cores_baseline(cpu["value"])
                if ac:
                    cpu["all_cores"] = ac
        avg_ms = sum(c["kernel_ms"]) / len(c["kernel_ms"])
        alg = (c["learn"] * bytes_["learn"] + c["test"] * bytes_["test"]) / steps
        achieved = alg / (avg_ms * 1e-3) / 1e9
        traffic, issue, src = measured_pmc(g["pmc_key"], n, g["trials"] or g["budget"]) if len(ctx) == 1 else (None, None, "none: concurrent contexts are not profiled separately")
        mean_curve = curve_host[i, :c["rows"], 0] / curve_host[i, :c["rows"], 2].clip(min=1)
        c["report"] = {"graph": c["graph"], "replicas": n, "trials_per_step": g["trials"] or None, "steps_budget_per_step": g.get("budget"), "replicas_per_wave": c["rpw"],
                       "env_steps_per_step_this_rank": (c["learn"] + c["test"]) / steps,
                       "mean_test_return_first_last": [float(mean_curve[0]), float(mean_curve[-1])],
                       "curve_replicas": float(curve_host[i, 0, 2]),
                       "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                    "traffic": traffic, "traffic_source": src, "issue": issue, "kernel": g["kernel"], "kernel_ms_avg": avg_ms,
                                    "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_learn_step": bytes_["learn"],
                                    "algorithmic_bytes_per_test_step": bytes_["test"], "algorithmic_bytes_source": bytes_["source"]}}
        if "oracle_this_run" in bytes_:
            c["report"]["roofline"]["oracle_bytes_per_learn_step_on_these_trials"] = bytes_["oracle_this_run"]
        if c["env_server"] != (0, 0):
            # the environment steps come from a second, co-resident kernel (grl_amd/csrc/grlx_env_server.h, grlx_env_server_wide.h); kernel_ms_avg
            # spans the pair.  The counters behind `traffic` are collected with the server off (rocprofv3 serialises kernels while it reads
            # counters), so the mailboxes' bytes are added here from the protocol: per served env-step, times the steps the served replicas took.
            served_share = c["env_server"][0] / max(sum(c["env_server"]), 1)
            env_steps = (c["learn"] + c["test"]) / steps
            mb = mailbox_bytes_per_step(c["graph"])
            c["report"]["env_server"] = {"replicas_served": c["env_server"][0], "replicas_fell_back": c["env_server"][1]}
            if mb:
                mail = mb["total"] * env_steps * served_share
                c["report"]["env_server"]["mailbox_bytes_per_served_step"] = mb
                r = c["report"]["roofline"]
                r["traffic_mailbox"] = mail
                r["traffic_kernel_alone"] = traffic
                if traffic is not None:
                    r["traffic"] = traffic + mail
                    r["traffic_source"] = src + " (the rollout kernel with the server off) + the mailboxes' bytes by the protocol, polls excluded (traffic_mailbox)"
        if cpu:
            c["report"]["cpu_baseline"] = cpu
    out["config"] = {"workload": "; ".join(text), "replicas_per_gpu": n_rank, "parallelism": out["parallelism"]}
    dom = [c for c in ctx if c["graph"] == w["dominant"]][0]
    out["roofline"] = dom["report"]["roofline"]
    if "cpu_baseline" in dom["report"]:
        out["cpu_baseline"] = dom["report"]["cpu_baseline"]
    out["mean_test_return_first_last"] = dom["report"]["mean_test_return_first_last"]
    out["curve_replicas"] = dom["report"]["curve_replicas"]
    out["replicas_per_wave"] = dom["report"]["replicas_per_wave"]
    if "env_server" in dom["report"]:
        out["env_server"] = dom["report"]["env_server"]
    if len(ctx) > 1:
        out["parts"] = [c["report"] for c in ctx]
    return out


def run_fqi(D, torch, cpu_baseline=True, replicas=None, batch_size=None, epochs=None):
    """The batch path on all ranks: replicas only -- rank g runs the independent-seed experiments
    g*R .. (g+1)*R - 1; the one collective is the all-reduce of the per-batch test-return statistics [batches][3]."""
    import numpy as np
    import grl_amd
    from grl_amd import parallel
    w = FQI
    R = replicas or w["replicas"]
    n = batch_size or w["batch_size"]
    ep = epochs or w["epochs"]
    (_, seeds), = parallel.partition(w["key"], D.rank, D.world, R)
    cfg = grl_amd.pendulum_fqi_config(R, batch_size=n, iterations=w["iterations"], epochs=ep, max_batches=2)
    r = grl_amd.FqiRunner(cfg, seeds)
    stream = torch.cuda.current_stream()
    r.run_batch(stream.cuda_stream)
    r.sync(stream.cuda_stream)
    stats = torch.zeros((2, 3), dtype=torch.float64, device="cuda")
    parallel.reduce_curve(torch.zeros_like(stats), D.world)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    D.barrier(torch)
    t0 = time.perf_counter()
    e0.record(stream)
    r.run_batch(stream.cuda_stream)
    e1.record(stream)
    r.sync(stream.cuda_stream)
    returns = np.array([r.rows(k, 2)[2] for k in range(R)])                 # [replica][batch]
    stats.copy_(torch.tensor(np.stack([returns.sum(0), (returns ** 2).sum(0), np.full(2, float(R))], axis=1)))
    parallel.reduce_curve(stats, D.world)
    D.barrier(torch)
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, D.world, device="cuda")
    its = [r.info(k)["iterations"] for k in range(R)]
    r.close()
    sample_epochs_rank = sum(its) * ep * 2 * n
    sample_epochs, = parallel.sum_over_ranks([sample_epochs_rank], D.world, device="cuda")
    if D.rank != 0:
        return None
    flops = sample_epochs_rank * w["flops_per_sample_epoch"]
    ms = e0.elapsed_time(e1)
    s = stats.cpu().numpy()
    traffic, issue, src = measured_pmc(w["key"], R, 2 * n)
    out = {"workload": w["key"], "value": sample_epochs / elapsed, "unit": "sample-epochs/s", "n_gpus": D.world, "steps": 1, "warmup": 1, "ms_per_step": 1e3 * elapsed,
           "scaling": "weak", "dtype": "f64", "data": "synthetic", "replicas_per_gpu": R, "parallelism": f"replicas only x{D.world} (no data-path collective)",
           "baseline_config": "BASELINE.json configs[4]",
           "config": {"workload": w["workload"].replace("100000 transitions", f"{n} transitions").replace("200000 stored", f"{2 * n} stored").replace("16 independent", f"{R} independent").replace("500 epochs", f"{ep} epochs"),
                      "replicas_per_gpu": R, "transitions_stored": 2 * n, "epochs": ep},
           "iterations_run": its, "mean_test_return": float(s[1, 0] / s[1, 2]), "curve_replicas": float(s[1, 2]),
           "parity": "unpinned by the reference (oracle/fqi.c D1-D4); HIP == oracle bit for bit",
           "roofline": {"bound": "valu", "achieved": flops / (ms * 1e-3) / 1e12, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / (ms * 1e-3) / 1e12 / F64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": src, "issue": issue,
                        "kernel": w["kernel"], "kernel_ms_total": ms, "algorithmic_flops_per_step": flops,
                        "note": "f64 vector ALU, no MFMA is issued (K = 3 and K = 20 contractions with N = 1 and the logistic dominate, DESIGN.md 4.3); "
                                "peak = the f64 vector rate"}}
    if cpu_baseline:
        from tests import oracle_binding as ob
        e = ob.FqiExperiment(ob.pendulum_fqi_spec(math=ob.MATH_LIBM, sum_order=ob.SUM_SEQUENTIAL, batch_size=4000, iterations=4, epochs=100), seed=1)
        t0 = time.perf_counter()
        e.run_batch()
        dt = time.perf_counter() - t0
        se = e.info()["iterations"] * 100 * 4000
        e.close()
        out["cpu_baseline"] = {"value": se / dt, "unit": "sample-epochs/s", "cores": 1, "kind": "port",
                               "sample": f"oracle/fqi.c (libm, the reference's sample-order gradient sum), 1 replica, 4000 transitions x {se // 4000} epochs in {dt:.1f} s "
                                         f"-- a SMALLER regime than the GPU's ({2 * n} transitions x {ep} epochs x {R} replicas): the unit is the same, the store fits the host's cache"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed launches of the chosen workload (default 20 for the headline)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="pendulum_sarsa", choices=sorted(WORKLOADS) + [FQI["key"]])
    ap.add_argument("--replicas", type=int, default=0, help="replicas per GPU of the chosen workload (default: BASELINE.json's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline only: skip the other BASELINE.json configurations")
    ap.add_argument("--no-fqi", action="store_true", help="skip the batch-path entry among the secondaries")
    ap.add_argument("--no-composite", action="store_true", help="profiling: skip acrobot_walker, whose two kernels are the ones of acrobot_q and compass_walker_q")
    ap.add_argument("--only", default="", help="synonym of --workload (kept for the profiling scripts)")
    ap.add_argument("--secondary-replicas", type=int, default=0, help="tests: replicas per GPU of the secondary workloads")
    ap.add_argument("--fqi-replicas", type=int, default=0, help="tests: replicas per GPU of the batch path")
    ap.add_argument("--fqi-batch-size", type=int, default=0, help="tests: transitions per batch of the batch path")
    ap.add_argument("--fqi-epochs", type=int, default=0)
    ap.add_argument("--table-log2", type=int, default=0)
    ap.add_argument("--replicas-per-wave", type=int, default=0, help="experiments: force the wave layout of the chosen workload (grlx_config.replicas_per_wave)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    args = ap.parse_args()
    if args.replicas_per_wave:
        WORKLOADS[args.only or args.workload]["replicas_per_wave"] = args.replicas_per_wave
    if args.only:
        args.workload = args.only
        args.no_secondary = True
    D = Dist(args)
    if os.environ.get("GRLX_DUMP_MAPS"):
        # diagnostic: the process's mappings when the interpreter ends (names the libraries behind the PCs of a crash in a C exit handler)
        import atexit
        atexit.register(lambda: open(os.environ["GRLX_DUMP_MAPS"], "w").write(open("/proc/self/maps").read()))

    import torch
    import torch.distributed as dist
    from grl_amd import parallel

    torch.cuda.set_device(D.local_rank % max(torch.cuda.device_count(), 1))
    parallel.init_distributed(args.backend)
    cpu = not args.no_cpu_baseline and D.world == 1              # cpu_baseline: rank 0 at N = 1 only

    if args.workload == FQI["key"]:
        out = run_fqi(D, torch, cpu, args.replicas or args.fqi_replicas or None, args.fqi_batch_size or None, args.fqi_epochs or None)
        head = FQI["key"]
    else:
        w = WORKLOADS[args.workload]
        steps = args.steps if args.steps is not None else w["steps"]
        warmup = args.warmup if args.warmup is not None else w["warmup"]
        out = run_rollout_workload(args.workload, D, torch, steps, warmup, args.replicas or None, args.table_log2 or None, cpu,
                                   full_cpu_baseline=args.workload == "pendulum_sarsa")
        head = args.workload
    if D.rank == 0:
        metric = {"pendulum_sarsa": "env-steps/sec (batched rollouts), pendulum SARSA-tc", FQI["key"]: "sample-epochs/sec (batch path), pendulum FQI-ANN"}
        line = {"metric": metric.get(head, "env-steps/sec (batched rollouts), " + head), "value": out["value"], "unit": out["unit"], "n_gpus": D.world,
                "steps": out["steps"], "warmup": out["warmup"], "ms_per_step": out["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
        line.update({k: v for k, v in out.items() if k not in line})
        if head == "pendulum_sarsa":
            n = out["replicas_per_gpu"]
            line["config"].update({"trials_per_step": TRIALS_PER_STEP, "env_steps_per_step": int(out["env_steps_per_step"]), "tilings": 16, "memory": 8388608})
            # every pendulum episode lasts exactly 100 steps: the devices' own step counters must say so.  Reported, not asserted: a line
            # with "learn_steps_expected" != "learn_steps" is a finding, a crashed bench is no line at all
            line["learn_steps_expected"] = n * 10 * STEPS_PER_EPISODE * out["steps"] * D.world
            if out["learn_steps"] != line["learn_steps_expected"]:
                print(f"bench.py: learning steps counted by the devices ({out['learn_steps']}) differ from 10 episodes x 100 steps per replica and "
                      f"launch ({line['learn_steps_expected']})", file=sys.stderr)
    if args.workload == "pendulum_sarsa" and not args.no_secondary:
        # the other configurations BASELINE.json names, each sharded over the same ranks and timed the same way
        sec = []
        for name in SECONDARY_ORDER:
            if name.startswith("acrobot_walker") and args.no_composite:
                continue
            w = WORKLOADS[name]
            sec.append(run_rollout_workload(name, D, torch, w["steps"], w["warmup"], args.secondary_replicas or None, None, cpu))
        if not args.no_fqi:
            sec.append(run_fqi(D, torch, cpu, args.fqi_replicas or None, args.fqi_batch_size or None, args.fqi_epochs or None))
        if D.rank == 0:
            line["secondary"] = sec
    if D.rank == 0:
        print(json.dumps(line))
    if D.world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
