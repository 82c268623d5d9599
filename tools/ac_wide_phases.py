#!/usr/bin/env python3
"""Phases of the wide actor-critic kernel (stamped build: GRLX_EXTRA_FLAGS=-DGRLX_WIDE_STAMPS): ac_wide_phases.py [replicas] [trials] [replicas per wave]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grl_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 11
rpw = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cfg = grl_amd.cart_pole_ac_config(n, max_rows=8 * trials + 8)
cfg.replicas_per_wave = rpw
r = grl_amd.Runner(cfg, np.arange(1, n + 1))
r.run(trials); r.sync()
r.set_diag(True)
l0, t0s = r.step_counts()
t0 = time.perf_counter(); r.run(trials); r.sync(); dt = time.perf_counter() - t0
l1, t1s = r.step_counts()
waves = min((n + rpw - 1) // rpw, 1024)
d = r.read_diag().astype(np.float64)[:waves]
steps = (l1 - l0) + (t1s - t0s)
passes = d[:, 2].sum()
names = {0: "environment", 1: "table phase (all sub-batches)", 3: "  unpark + hash + issue loads", 4: "  deferred critic update", 5: "  finish lookups, reconcile", 6: "  sums, policy, actor update", 7: "  between trials + park"}
ins = sum(r.table_load(k) for k in range(0, n, 257)) / len(range(0, n, 257))
print("mean critic-table slots of sampled replicas after", 2 * trials, "trials:", ins)
print(f"{n} replicas: {steps/dt/1e6:.1f} M env-steps/s (stamped build); passes per wave {passes/len(d):.0f}; replica-steps per wave-pass {steps/passes:.2f}")
for k in (0, 1, 3, 4, 5, 6, 7):
    print(f"  {names[k]:34s} {d[:, k].sum()/passes:8.0f} cycles per wave-pass")
