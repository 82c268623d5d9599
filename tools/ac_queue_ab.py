#!/usr/bin/env python3
"""Cart-pole actor-critic, 16384 replicas: the wide kernel with and without the device-side replica queue (wave_limit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grl_amd

def run(n, wave_limit, trials=11, launches=4, **over):
    cfg = grl_amd.cart_pole_ac_config(n, max_rows=trials * (launches + 2) + 8, wave_limit=wave_limit, **over)
    r = grl_amd.Runner(cfg, np.arange(1, n + 1))
    r.run(trials); r.run(trials); r.sync()
    res = []
    for _ in range(launches):
        l0, t0 = r.step_counts()
        t = time.time(); r.run(trials); r.sync(); dt = time.time() - t
        l1, t1 = r.step_counts()
        res.append(((l1 + t1 - l0 - t0) / dt / 1e6, dt * 1e3, (l1 + t1 - l0 - t0) / n / trials))
    r.close()
    return res

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for over in (dict(), dict(end_stop_penalty=1)):
    for wl in (2048, 1024):
        res = run(n, wl, **over)
        print(over, "wave_limit", wl, " ".join(f"{v:.0f}M/{ms:.0f}ms/{spt:.0f}steps-per-trial" for v, ms, spt in res))
