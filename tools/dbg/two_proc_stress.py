#!/usr/bin/env python3
"""Two (or more) processes share one GPU, each running the headline configuration (256 pendulum replicas, 11-trial launches) with the environment
server on: after every launch the device's step counters must be exactly 1000 learning + 100 test steps per replica, and at the end rows and
streams must equal a run of the same seeds with the server off.  Looks for the rare +8 learning steps seen once in the 2-rank bench rehearsal.
   python3 tools/dbg/two_proc_stress.py [processes] [launches]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(rank, launches):
    import numpy as np
    import grl_amd
    n = 256
    seeds = np.arange(1 + rank * n, 1 + (rank + 1) * n)
    cfg = grl_amd.pendulum_sarsa_config(n, max_rows=launches + 2)
    r = grl_amd.Runner(cfg, seeds)
    bad = 0
    for k in range(launches):
        r.run(11); r.sync()
        learn, test = r.step_counts()
        if (learn, test) != (n * 1000 * (k + 1), n * 100 * (k + 1)):
            print(f"rank {rank} launch {k}: step counts {learn} {test} expected {n * 1000 * (k + 1)} {n * 100 * (k + 1)}; server {r.env_server_counts()}", flush=True)
            bad += 1
            break
    served = r.env_server_counts()
    rows = [r.rows(i) for i in range(n)]
    rng = [list(r.rng(i)) for i in range(n)]
    r.close()
    os.environ["GRLX_ENV_SERVER"] = "0"
    r = grl_amd.Runner(cfg, seeds)
    for k in range(launches if not bad else k + 1):
        r.run(11)
    r.sync()
    diff = [i for i in range(n) if not (np.array_equal(rows[i][1], r.rows(i)[1]) and np.array_equal(np.asarray(rows[i][2]).view(np.uint64), np.asarray(r.rows(i)[2]).view(np.uint64)) and rng[i] == list(r.rng(i)))]
    print(f"rank {rank}: {launches} launches, last server counts {served}, replicas that differ from the server-off run: {diff[:16]} ({len(diff)})", flush=True)
    r.close()
    return 1 if (bad or diff) else 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        sys.exit(child(int(sys.argv[2]), int(sys.argv[3])))
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(k), str(launches)]) for k in range(procs)]
    rc = [p.wait() for p in ps]
    print("exit codes", rc)
    sys.exit(max(rc))
