#!/usr/bin/env python3
"""Debug helper: the 5-action pendulum instantiation, deferred vs in-place vs oracle, for parameter subsets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import grl_amd
from tests import oracle_binding as ob

def apply(obj, over):
    for k, v in over.items():
        if k.startswith("representation."):
            setattr(obj.representation, k.split(".", 1)[1], v)
        else:
            setattr(obj, k, v)

CASES = {
  "five": dict(action_steps=5),
  "five+ti": dict(action_steps=5, test_interval=4),
  "five+rand": dict(action_steps=5, randomization=1.0),
  "five+lim": {"action_steps": 5, "representation.output_min": -500.0, "representation.output_max": 10.0},
  "five+decay": dict(action_steps=5, decay_rate=0.97, decay_min=0.2),
  "three+lim": {"representation.output_min": -500.0, "representation.output_max": 10.0},
  "all": {"action_steps": 5, "test_interval": 4, "decay_rate": 0.97, "decay_min": 0.2, "randomization": 1.0,
          "representation.output_min": -500.0, "representation.output_max": 10.0},
}
seeds = [31, 32, 33]
trials = 25
for rep in range(2):
  for name, over in CASES.items():
    for agent in (1, 0):
        o2 = dict(over); o2.setdefault("test_interval", -1)
        res = {}
        for mode in ("deferred", "inplace"):
            cfg = grl_amd.pendulum_sarsa_config(len(seeds), agent=agent, max_rows=trials + 1)
            apply(cfg, o2)
            r = grl_amd.Runner(cfg, seeds)
            if mode == "inplace":
                r.set_diag(True)
            r.run(10); r.run(15); r.sync()
            res[mode] = [r.rows(k)[2].copy() for k in range(len(seeds))]
            r.close()
        bad = []
        for k, seed in enumerate(seeds):
            spec = ob.pendulum_sarsa_spec(agent=agent)
            apply(spec, o2)
            e = ob.Experiment(spec, seed=seed)
            rows, _ = e.run(trials)
            want = np.array([x.reward for x in rows])
            for mode in res:
                d = np.nonzero(res[mode][k].view(np.uint64) != want.view(np.uint64))[0]
                if d.size:
                    bad.append((mode, seed, int(d[0])))
            e.close()
        print(rep, name, "agent", agent, "OK" if not bad else "FIRST DIFFERING ROW per (mode, seed): %s" % bad, flush=True)
