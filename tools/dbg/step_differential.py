#!/usr/bin/env python3
"""Differential run of the 5-action production kernel, k control steps at a time: environment state, RNG positions and the
dense weight vector against the oracle after each prefix (how DESIGN.md section 4.1f located the first wrong update)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import grl_amd
from tests import oracle_binding as ob
from tests.test_gpu_generic_paths import _apply
seeds = [31]
for k in (1, 2, 3, 5):
    over = dict(action_steps=5, test_interval=4, randomization=1.0, timeout=0.03 * k + 0.001)
    cfg = grl_amd.pendulum_sarsa_config(len(seeds), agent=1, max_rows=4)
    _apply(cfg, over)
    r = grl_amd.Runner(cfg, seeds); r.run(1); r.sync()
    spec = ob.pendulum_sarsa_spec(agent=1); _apply(spec, over)
    e = ob.Experiment(spec, seed=seeds[0]); e.run(1)
    gw = np.asarray(r.export_weights(0)); allslots = np.arange(cfg.projector.memory, dtype=np.uint32)
    ow = e.all_weights()
    bad = np.nonzero(gw.view(np.uint64) != ow.view(np.uint64))[0]
    print("k", k, "steps", r.step_counts(), "state", list(r.env_state(0)), list(e.state()), "rng eq", list(r.rng(0))[:3] == list(e.rng())[:3], "weight diffs", bad.size)
    for b in bad[:12]:
        print("    slot", int(b), "gpu", gw[b], "orc", ow[b])
    r.close(); e.close()
