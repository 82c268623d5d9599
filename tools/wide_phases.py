#!/usr/bin/env python3
"""Environment vs table phase of the wide kernels (stamped build: GRLX_EXTRA_FLAGS=-DGRLX_WIDE_STAMPS):
wide_phases.py <acrobot|compass_walker|pendulum> <replicas> <warm trials> <trials>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grl_amd
name, n, warm, trials = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
make = {"acrobot": grl_amd.acrobot_q_config, "compass_walker": grl_amd.compass_walker_q_config, "pendulum": lambda k: grl_amd.pendulum_sarsa_config(k, agent=1)}[name]
cfg = make(n); cfg.max_rows = 512; cfg.replicas_per_wave = 8
r = grl_amd.Runner(cfg, np.arange(1, n + 1))
r.run(warm); r.sync()
r.set_diag(True)
l0, t0s = r.step_counts()
t0 = time.perf_counter(); r.run(trials); r.sync(); dt = time.perf_counter() - t0
l1, t1s = r.step_counts()
d = r.read_diag().astype(np.float64)[: (n + 7) // 8]
env, tab, passes = d[:, 0].sum(), d[:, 1].sum(), d[:, 2].sum()
steps = (l1 - l0) + (t1s - t0s)
print(f"{name} {n} replicas: {steps/dt/1e6:.1f} M env-steps/s (stamped build); per wave-pass: environment {env/passes:.0f} cycles, table phase (2 sub-batches) {tab/passes:.0f} cycles; "
      f"environment share {env/(env+tab):.3f}; passes per wave {passes/len(d):.0f}; replica-steps per wave-pass {steps/passes:.2f} of 8")
