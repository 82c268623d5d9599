#!/usr/bin/env python3
"""env-steps/s of the other BASELINE.json configurations (not the bench line; recorded in DESIGN.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grl_amd
from tests import configs
configs.NO_ORACLE = True            # the grlx_config half of the builders only: the oracle is test infrastructure

CASES = [("pendulum SARSA-tc", configs.pendulum, 4096, 110), ("pendulum Q-tc", lambda g, n: configs.pendulum(g, n, agent=1), 4096, 110),
         ("cart-pole AC-tc", configs.cart_pole_ac, 16384, 44), ("acrobot Q-tc", configs.acrobot, 8192, 110),
         ("compass walker Q-tc", configs.compass_walker, 8192, 44)]
for name, make, n, trials in CASES:
    cfg, _ = make(grl_amd, n)
    cfg.max_rows = 64
    r = grl_amd.Runner(cfg, np.arange(1, n + 1))
    r.run(11); r.sync()
    l0, t0s = r.step_counts()
    t0 = time.perf_counter(); r.run(trials); r.sync(); dt = time.perf_counter() - t0
    l1, t1s = r.step_counts()
    steps = (l1 - l0) + (t1s - t0s)
    print(f"{name:22s} {n:6d} replicas  {trials:4d} trials  {steps/1e6:9.1f} M env-steps in {dt*1e3:8.1f} ms -> {steps/dt/1e6:8.1f} M env-steps/s")
    r.close()
