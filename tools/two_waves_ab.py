#!/usr/bin/env python3
"""A/B for DESIGN.md section 5.2: the headline workload at 8192 replicas (two waves' worth of work per SIMD) through
(a) the 4-replicas-per-wave kernel, 512 registers, one wave per SIMD at a time; (b) the 8-replicas-per-wave kernel;
(c) with GRLX_LIB pointing at a build whose rollout kernels are limited to 256 registers, (a) again: two waves per SIMD."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grl_amd

def run(n, rpw, trials=22, reps=3):
    cfg = grl_amd.pendulum_sarsa_config(n, max_rows=4 * trials + 8, replicas_per_wave=rpw)
    r = grl_amd.Runner(cfg, list(range(1, n + 1)))
    r.run(trials); r.sync()
    best = 0
    for _ in range(reps):
        l0, t0 = r.step_counts()
        t = time.time(); r.run(trials); r.sync(); dt = time.time() - t
        l1, t1 = r.step_counts()
        best = max(best, (l1 + t1 - l0 - t0) / dt)
    k = r.last_kernel(); got = r.replicas_per_wave(); r.close()
    return best, k, got

for n in (4096, 8192):
    for rpw in (4, 8):
        v, k, got = run(n, rpw)
        print(f"replicas {n} requested rpw {rpw} got {got} kernel {k}: {v / 1e6:.1f} M env-steps/s")
