#!/usr/bin/env python3
"""Where the cycles of the environment server go (stats build: GRLX_EXTRA_FLAGS=-DGRLX_ENV_SERVER_STATS, library named by GRLX_LIB;
run with GRLX_ENV_SERVER=1): env_server_stats.py [replicas] [trials <= 32, one launch]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import grl_amd
from grl_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = grl_amd.pendulum_sarsa_config(n)
r = grl_amd.Runner(cfg, np.arange(1, n + 1))
r.run(trials); r.sync()          # warm tables
l0, t0 = r.step_counts()
r.run(trials); r.sync()
l1, t1 = r.step_counts()
lib = r.lib
lib.grlx_env_server_debug.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
mail = np.zeros((n, 128), dtype=np.uint64)
capi.check(lib.grlx_env_server_debug(r._ctx, mail.ctypes.data_as(C.c_void_p), mail.nbytes))
st = mail[:, 16:32].astype(np.float64)
steps = (l1 - l0) + (t1 - t0)
print(f"{n} replicas, {trials} trials: {steps} env-steps, {steps / n:.0f} per replica")
print(f"server : kernel {st[:,0].mean():.0f} cycles, busy (command seen -> candidates stored) {st[:,1].mean():.0f} = {st[:,1].sum()/st[:,0].sum():.2f}, "
      f"{st[:,2].mean():.0f} commands, {st[:,1].sum()/max(st[:,2].sum(),1):.0f} cycles per command, {st[:,3].mean():.0f} idle polls, "
      f"{st[:,10].mean():.0f} batches of which {st[:,9].mean():.0f} without all four replicas of the wave")
print(f"rollout: kernel {st[:,4].mean():.0f} cycles, waiting for candidates {st[:,5].mean():.0f} = {st[:,5].sum()/st[:,4].sum():.2f}, "
      f"{st[:,6].mean():.0f} fetches, {st[:,5].sum()/max(st[:,6].sum(),1):.0f} cycles per fetch, {st[:,7].sum()/max(st[:,6].sum(),1):.2f} extra polls per fetch, "
      f"{int(st[:,8].sum())} of {n} replicas still served at the end")
print(f"cycles per pass (rollout kernel / fetches): {st[:,4].sum()/max(st[:,6].sum(),1):.0f}   (s_memtime ticks = shader cycles)")
ph = mail[:, 92:100].astype(np.float64)       # EnvMail::pad1[0..7]: the rollout wave's stamps (same slots as tools/diag_phases.py)
names = ["loop/bookkeeping", "environment step: taking the server's answer", "tile hashing", "inserts + LDS writes", "LDS sums + sampler (+ command, load ahead)",
         "TD update + trace", "wait for previous stores", "table lookup (loads)"]
tot = ph.sum()
if tot > 0:
    print("rollout wave, share of the stamped cycles per phase (%.0f cycles per pass):" % (tot / max(st[:, 6].sum(), 1)))
    for k, nm in enumerate(names):
        print(f"  {nm:46s} {ph[:, k].sum() / tot:6.3f}   {ph[:, k].sum() / max(st[:, 6].sum(), 1):8.0f} cycles per pass")
r.close()
