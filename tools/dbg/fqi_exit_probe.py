"""Why does `rocprofv3 -- python3 bench.py --workload pendulum_fqi_ann` end with SIGSEGV (rc 139) after "tool finalization"?
One process per variant, each dumps /proc/self/maps from a Python atexit hook (the C exit handlers, where the fault is, run after
it; the libraries are still mapped), so the PCs of rocprofv3's stack trace can be given a library and a symbol.

  python3 tools/dbg/fqi_exit_probe.py <variant> <maps file>
    fqi          FqiRunner without torch: one small batch, sync, close             (cooperative launch)
    fqi_nocoop   the same with GRLX_FQI_NO_COOP=1                                  (plain launch of the same grid)
    fqi_leak     the same as fqi, but the context is NOT destroyed before exit
    rollout      a pendulum Runner without torch: 22 trials, sync, close           (control: no cooperative launch)
"""
import atexit
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def dump_maps(path):
    with open("/proc/self/maps") as f, open(path, "w") as g:
        g.write(f.read())


def main():
    variant, maps = sys.argv[1], sys.argv[2]
    atexit.register(dump_maps, maps)
    if variant == "fqi_nocoop":
        os.environ["GRLX_FQI_NO_COOP"] = "1"
    import grl_amd
    if variant.startswith("fqi"):
        cfg = grl_amd.pendulum_fqi_config(2, batch_size=4000, iterations=2, epochs=20, max_batches=1)
        r = grl_amd.FqiRunner(cfg, [1, 2])
        r.run_batch()
        r.sync()
        print(variant, "rows", r.rows(0, 1)[2], flush=True)
        if variant == "fqi_leak":
            r._ctx = None          # keep the device context alive to the end of the process
        else:
            r.close()
    else:
        cfg = grl_amd.pendulum_sarsa_config(64)
        r = grl_amd.Runner(cfg, list(range(1, 65)))
        r.run(22)
        r.sync()
        print(variant, "rows", r.rows(0)[2], flush=True)
        r.close()


if __name__ == "__main__":
    main()
