#!/usr/bin/env python3
"""Where does a rollout step spend its cycles?  Runs the STAMPED diagnostic build of the
kernel (s_memtime per phase) and prints the share of each phase.  Shares only -- the
stamped build's own run time is not representative."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grl_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 11
logc = int(sys.argv[3]) if len(sys.argv) > 3 else 17
cfg = grl_amd.pendulum_sarsa_config(n, max_rows=512, table_log2_capacity=logc)
r = grl_amd.Runner(cfg, np.arange(1, n + 1))
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 33
r.run(warm); r.sync()                     # warm tables
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 1   # 1: in-place update, 2: deferred update (production ordering)
r.set_diag(mode)
t0 = time.perf_counter(); r.run(trials); r.sync(); dt = time.perf_counter() - t0
d = r.read_diag().astype(np.float64)
names = ["loop/bookkeeping", "env step (RK4)", "tile hashing", "inserts + LDS writes", "LDS sums + sampler", "TD update + trace", "wait for previous stores", "table lookup (loads)"]
steps = trials * 100 + trials
tot = d.sum(1).mean()
print(f"warmup {warm} trials; logC {logc} replicas {n}, {trials} trials, stamped launch {dt*1e3:.1f} ms, mean cycles/step/wave {tot/steps:.0f}")
for k, nm in enumerate(names):
    print(f"  {nm:22s} {d[:,k].mean()/steps:9.0f} cycles/step  {100*d[:,k].mean()/tot:5.1f} %")
r.set_diag(False)
t0 = time.perf_counter(); r.run(trials); r.sync(); dt = time.perf_counter() - t0
print(f"production launch {dt*1e3:.1f} ms -> {n*trials*100/dt/1e6:.1f} M env-steps/s")
