#!/usr/bin/env python3
"""Time per epoch of the batch path: fqi_timing.py <replicas> <batch_size> <iterations> <epochs>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grl_amd
R, n, its, ep = (int(v) for v in sys.argv[1:5])
cfg = grl_amd.pendulum_fqi_config(R, batch_size=n, iterations=its, epochs=ep, max_batches=2)
r = grl_amd.FqiRunner(cfg, np.arange(1, R + 1))
r.run_batch(); r.sync()
t0 = time.perf_counter(); r.run_batch(); r.sync(); dt = time.perf_counter() - t0
it = r.info(0)["iterations"]
se = R * 2 * n * it * ep
print(f"R={R} n={2*n} iterations={it} epochs={ep}: {dt*1e3:.1f} ms, {dt/(it*ep)*1e6:.1f} us/epoch, {se/dt/1e9:.2f} G sample-epochs/s, {444*se/dt/1e12:.3f} TFLOP/s; return {r.rows(0,2)[2]}")
