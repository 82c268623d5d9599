#!/usr/bin/env python3
"""Launches bounded by trials against launches bounded by a steps budget (grlx_run_steps), same graph, same replicas:
steps_budget_ab.py <acrobot|compass_walker|cart_pole_ac|pendulum> <replicas> <launches> <trials per launch> <steps per launch>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grl_amd
name, n, launches, trials, budget = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
make = {"acrobot": grl_amd.acrobot_q_config, "compass_walker": grl_amd.compass_walker_q_config, "cart_pole_ac": grl_amd.cart_pole_ac_config,
        "pendulum": grl_amd.pendulum_sarsa_config}[name]
for mode in ("trials", "steps"):
    cfg = make(n); cfg.max_rows = 4096
    if name == "cart_pole_ac": cfg.table_log2_capacity = 18
    r = grl_amd.Runner(cfg, np.arange(1, n + 1))
    def launch(k):
        if mode == "trials": r.run(trials)
        else: r.run_steps(1 << 20, (k + 1) * budget)
    launch(0); r.sync()
    l0, t0s = r.step_counts()
    t0 = time.perf_counter()
    for k in range(1, launches + 1): launch(k)
    r.sync(); dt = time.perf_counter() - t0
    l1, t1s = r.step_counts()
    steps = (l1 - l0) + (t1s - t0s)
    print(f"{name} {n} replicas, {launches} launches bounded by {mode} ({trials} trials | {budget} learning steps per launch): "
          f"{steps/1e6:.1f} M env-steps in {dt*1e3:.1f} ms = {steps/dt/1e6:.1f} M env-steps/s, replicas per wave {r.replicas_per_wave()}, kernel {r.last_kernel()}")
    r.close()
