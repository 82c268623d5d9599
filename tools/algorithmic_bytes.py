#!/usr/bin/env python3
"""Algorithmic bytes per env-step of every bench.py workload, counted by the oracle (test infrastructure) on the
trials bench.py times: 8 B per weight read (`weight_reads`) + 16 B per read-modify-write (`weight_rmws`), SURVEY.md 8(d).
A test step reads `reads_per_test_step` weights (NA x T for the greedy Q policy, T for the actor) and writes nothing.

    python3 tools/algorithmic_bytes.py [seeds]      -> the table committed as ALGORITHMIC_BYTES in bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import configs, oracle_binding as ob  # noqa: E402

# graph -> (oracle spec builder, what bench.py runs (warm-up + timed): trials, or -(learning steps) for the steps-budget launches,
#           weights read per test step)
GRAPHS = {"pendulum_sarsa": (lambda: configs.pendulum(None, 1)[1], 23 * 11, 48),
          "cart_pole_ac": (lambda: configs.cart_pole_ac(None, 1)[1], 6 * 11, 16),
          "acrobot_q": (lambda: configs.acrobot(None, 1)[1], -6 * 1100, 48),
          "compass_walker_q": (lambda: configs.compass_walker(None, 1)[1], -6 * 12200, 48)}


def measure(graph, seed, trials=None):
    make, t, rpt = GRAPHS[graph]
    spec = make()
    spec.math = ob.MATH_LIBM
    e = ob.Experiment(spec, seed=seed)
    t = trials or t
    if t < 0:
        e.set_steps_budget(-t)
        e.run(1 << 20)
    else:
        e.run(t)
    st = e.stats()
    e.close()
    learn_reads = st.weight_reads - st.test_steps * rpt
    return dict(learn_steps=int(st.learn_steps), test_steps=int(st.test_steps), reads_per_learn_step=learn_reads / st.learn_steps,
                rmws_per_learn_step=st.weight_rmws / st.learn_steps,
                bytes_per_learn_step=(8 * learn_reads + 16 * st.weight_rmws) / st.learn_steps, bytes_per_test_step=8 * rpt)


if __name__ == "__main__":
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    for g in GRAPHS:
        rows = [measure(g, s) for s in range(1, seeds + 1)]
        mean = sum(r["bytes_per_learn_step"] for r in rows) / len(rows)
        print(g, "bytes/learn-step per seed:", ["%.1f" % r["bytes_per_learn_step"] for r in rows], "mean %.1f" % mean,
              "reads %.2f rmws %.2f" % (sum(r["reads_per_learn_step"] for r in rows) / len(rows), sum(r["rmws_per_learn_step"] for r in rows) / len(rows)),
              "test-step bytes", rows[0]["bytes_per_test_step"], "learn steps", [r["learn_steps"] for r in rows])
