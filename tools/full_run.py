#!/usr/bin/env python3
"""Whole-run throughput of BASELINE config 2: 4096 replicas x 2000 trials (181 900 learning + 18 100 test
steps each, 8.19e8 env-steps), cold tables, launches of 11 trials; prints the rate per 220-trial segment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grl_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = grl_amd.pendulum_sarsa_config(n, max_rows=200)
r = grl_amd.Runner(cfg, np.arange(1, n + 1))
t_all = time.perf_counter()
done = 0
while done < 2000:
    seg = min(220, 2000 - done)
    t0 = time.perf_counter()
    for _ in range(seg // 11):
        r.run(11)
    if seg % 11:
        r.run(seg % 11)
    r.sync()
    dt = time.perf_counter() - t0
    done += seg
    print(f"trials {done - seg:4d}..{done:4d}: {n * seg * 100 / dt / 1e6:7.1f} M env-steps/s", flush=True)
dt = time.perf_counter() - t_all
learn, test = r.step_counts()
print(f"whole run: {(learn + test) / 1e6:.1f} M env-steps in {dt:.2f} s -> {(learn + test) / dt / 1e6:.1f} M env-steps/s; "
      f"table load {r.table_load(0)} slots of {1 << cfg.table_log2_capacity if cfg.table_log2_capacity else 131072}")
rows = r.rows(0)
print("replica 0 first/last test returns:", rows[2][0], rows[2][-1])
import hashlib
h = hashlib.sha256()
for k in range(0, n, max(n // 64, 1)):                # 64 replicas' complete learning curves: compare two builds / GRLX_ENV_SERVER settings
    h.update(np.ascontiguousarray(r.rows(k)[2]).tobytes())
print("sha256 of 64 replicas' returns:", h.hexdigest()[:16], " environment server (served, fell back in the last launch):", r.env_server_counts())
