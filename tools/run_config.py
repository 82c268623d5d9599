#!/usr/bin/env python3
"""Run one of the BASELINE.json configurations for profiling: run_config.py <name> <replicas> <warm trials> <trials>
name: pendulum | pendulum_q | cart_pole_ac | acrobot | compass_walker"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grl_amd
from tests import configs
configs.NO_ORACLE = True            # the grlx_config half of the builders only: the oracle is test infrastructure

name, n, warm, trials = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
make = {"pendulum_acc": lambda g, k: configs.pendulum(g, k, trace=2), "pendulum_adv": lambda g, k: configs.pendulum(g, k, agent=4, kappa=0.2),
        "pendulum_qv": configs.pendulum_qv, "pendulum": configs.pendulum, "pendulum_q": lambda g, k: configs.pendulum(g, k, agent=1), "cart_pole_ac": configs.cart_pole_ac,
        "acrobot": configs.acrobot, "compass_walker": configs.compass_walker}[name]
cfg, _ = make(grl_amd, n)
cfg.max_rows = 256
cfg.force_generic = int(os.environ.get("GRLX_FORCE_GENERIC", "0"))     # A/B: generic vs specialised instantiation
r = grl_amd.Runner(cfg, np.arange(1, n + 1))
r.run(warm); r.sync()
l0, t0s = r.step_counts()
t0 = time.perf_counter(); r.run(trials); r.sync(); dt = time.perf_counter() - t0
l1, t1s = r.step_counts()
steps = (l1 - l0) + (t1s - t0s)
print(f"{name} {n} replicas {trials} trials (kernel variant {r.last_kernel()}): {steps/1e6:.1f} M env-steps in {dt*1e3:.1f} ms -> {steps/dt/1e6:.1f} M env-steps/s")
