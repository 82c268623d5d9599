#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import grl_amd
from tests import oracle_binding as ob
cap = 250
cfg = grl_amd.pendulum_sarsa_config(1, agent=1, tap_replica=0, tap_capacity=cap)
cfg.action_steps = 5
r = grl_amd.Runner(cfg, [31]); r.run(2); r.sync()
gt = r.taps(); r.close()
spec = ob.pendulum_sarsa_spec(agent=1); spec.action_steps = 5
e = ob.Experiment(spec, seed=31)
_, ot = e.run(2, tap_cap=cap)
print(len(gt), len(ot))
for k, (g, o) in enumerate(zip(gt, ot)):
    same = (list(g.q[:5]) == list(o.q[:5]) and g.action_index == o.action_index and g.reward == o.reward and g.delta == o.delta
            and list(g.p_idx[:16]) == list(o.p_idx[:16]) and list(g.obs[:2]) == list(o.obs[:2]))
    if not same or k < 2:
        print("step", k, "same" if same else "DIFF")
        print("  q  gpu", list(g.q[:5]), "\n  q  orc", list(o.q[:5]))
        print("  a", g.action_index, o.action_index, "action", g.action, o.action, "r", g.reward, o.reward, "delta", g.delta, o.delta)
        print("  obs", list(g.obs[:2]), list(o.obs[:2]))
        print("  idx gpu", list(g.p_idx[:16]), "\n  idx orc", list(o.p_idx[:16]))
        if not same:
            break
