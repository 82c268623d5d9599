#!/usr/bin/env python3
"""env-steps/s of the accumulating-trace kernel (pendulum SARSA-tc with trace/enumerated/accumulating, 4096 replicas)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grl_amd
n = 4096
for agent, name in ((0, "SARSA"), (1, "Q")):
    cfg = grl_amd.pendulum_sarsa_config(n, agent=agent, trace=2, max_rows=64)
    r = grl_amd.Runner(cfg, np.arange(1, n + 1))
    r.run(11); r.sync()
    res = []
    for _ in range(3):
        l0, t0s = r.step_counts()
        t0 = time.perf_counter(); r.run(44); r.sync(); dt = time.perf_counter() - t0
        l1, t1s = r.step_counts()
        res.append(((l1 - l0) + (t1s - t0s)) / dt / 1e6)
    print(f"accumulating trace, {name}, {n} replicas: " + " ".join(f"{v:.1f}" for v in res) + " M env-steps/s")
    r.close()
