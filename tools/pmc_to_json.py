#!/usr/bin/env python3
"""Turn the rocprofv3 passes of tools/profile_passes.sh into (i) a per-kernel summary (stdout: the text committed as
profiles/<tag>_pmc_raw.txt) and (ii) profiles/pmc_traffic.json, which bench.py reads for roofline.traffic / .issue.

    python3 tools/pmc_to_json.py gpurun_out/<tag> [--write]

Every workload of bench.py runs its own kernel instantiation, so launches are grouped by kernel name; the mean is taken
over the LAST `timed` launches of each (the ones bench.py times; the warm-up launches come first)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_hash  # noqa: E402  (identifies the device code the passes ran on)
# kernel-name fragment -> (bench.py workload key, replicas, trials per launch, timed launches)
# (the headline: rollout_served_kernel beside env_server_kernel in the stats pass; rollout_kernel in the counter passes, which run with
#  GRLX_ENV_SERVER=0 because rocprofv3 serialises kernels while it reads counters and the pair only exists together)
WORKLOADS = [("rollout_kernel<0, 3, false, grlx::SpecPendulumTcA<0>", "pendulum_sarsa", 4096, 11, 20),
             ("rollout_served_kernel<3, grlx::SpecPendulumTcA<0>", "pendulum_sarsa", 4096, 11, 20),
             ("env_server_kernel<0, 3, grlx::SpecPendulumTcA<0>", "pendulum_sarsa_env_server", 4096, 11, 20),
             ("rollout_ac_wide_kernel<1, 4, grlx::SpecCartPoleAc>", "cart_pole_ac", 16384, 11, 5),
             ("rollout_wide_kernel<2, 3, 2, grlx::SpecAcrobotQ>", "acrobot_q", 8192, 1100, 5),          # (steps budget per launch)
             ("rollout_wide_served_kernel<2, grlx::SpecAcrobotQ>", "acrobot_q", 8192, 1100, 5),         # (the stats pass: beside its server)
             ("env_server_acrobot_pinned_kernel<grlx::SpecAcrobotQ>", "acrobot_q_env_server", 8192, 1100, 5),
             ("rollout_wide_kernel<3, 3, 8, grlx::SpecWalkerQ>", "compass_walker_q", 32768, 12200, 4)]


def workload_of(name):
    for frag, key, n, trials, timed in WORKLOADS:
        if frag in name:
            return key, n, trials, timed
    return None


def fqi_entry(out):
    """The batch path (bench.py --workload pendulum_fqi_ann: one warm-up batch, one timed batch).  A "launch" of bench.py's roofline
    object is the whole timed batch: per kernel name the second half of its launches, summed over all fqi kernels."""
    def timed_sum(vals):
        return sum(vals[len(vals) // 2:])
    dur = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(out, "fqi_stats", "**", "*kernel_trace.csv"), recursive=True)):
        with open(path) as f:
            for row in csv.DictReader(f):
                if "fqi_" in row.get("Kernel_Name", ""):
                    dur[row["Kernel_Name"].split("(")[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    for name, v in sorted(dur.items()):
        print(f"fqi trace: {name}: launches={len(v)} timed batch: {len(v) - len(v) // 2} launches, {timed_sum(v):.3f} ms, mean {timed_sum(v) / max(len(v) - len(v) // 2, 1):.4f} ms")
    counters = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(out, "fqi_pmc*"))):
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if "fqi_" in row.get("Kernel_Name", ""):
                        counters[row["Counter_Name"]][row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    tot = {}
    for cname, per_kernel in counters.items():
        tot[cname] = sum(timed_sum(v) for v in per_kernel.values())
        for kname, v in sorted(per_kernel.items()):
            print(f"fqi pmc {cname} {kname}: launches={len(v)} timed-batch sum={timed_sum(v):.6g}")
    if "FETCH_SIZE" not in tot or "WRITE_SIZE" not in tot:
        return None
    e = {"kernel": "fqi_epochs_kernel<20> (+ the other fqi kernels of one batch)", "replicas": 16, "trials_per_launch": 200000,
         "hbm_bytes_per_launch": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0, "fetch_bytes": tot["FETCH_SIZE"] * 1024.0,
         "write_bytes": tot["WRITE_SIZE"] * 1024.0, "source": os.path.basename(out.rstrip("/")),
         "note": "per timed batch (all kernels of the second grlx_fqi_run_batch); trials_per_launch holds the stored transitions"}
    if all(k in tot for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES")) and tot["SQ_WAVE_CYCLES"]:
        e["issue"] = (tot["SQ_ACTIVE_INST_VALU"] + tot["SQ_ACTIVE_INST_SCA"]) / tot["SQ_WAVE_CYCLES"]
    if "SQ_WAIT_ANY" in tot and tot.get("SQ_WAVE_CYCLES"):
        e["wait_any"] = tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"]
    ek = [v for k, v in dur.items() if "fqi_epochs_kernel" in k]
    if ek:
        e["kernel_ms_trace"] = sum(timed_sum(v) for v in dur.values())
        e["epochs_kernel_ms_per_launch"] = timed_sum(ek[0]) / max(len(ek[0]) - len(ek[0]) // 2, 1)
    return e


def main():
    out = sys.argv[1]
    write = "--write" in sys.argv[2:]
    stats = {}
    for path in sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)):
        with open(path) as f:
            for row in csv.DictReader(f):
                w = workload_of(row.get("Name", ""))
                if w:
                    stats[w[0]] = {k: row[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs") if k in row}
                    print("stats:", w[0], stats[w[0]])
    # per-launch durations of the timed launches from the kernel trace of the stats pass
    for path in sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True)):
        dur = defaultdict(list)
        with open(path) as f:
            for row in csv.DictReader(f):
                w = workload_of(row.get("Kernel_Name", ""))
                if w:
                    dur[w[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
        for frag, key, n, trials, timed in WORKLOADS:
            if key in dur:
                v = dur[key][-timed:]
                print(f"trace: {key}: launches={len(dur[key])} mean_timed_ms={sum(v) / len(v):.4f} (last {len(v)})")
                stats.setdefault(key, {})["timed_ms"] = sum(v) / len(v)
    means = defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
        if not os.path.isdir(d):
            continue
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            vals = defaultdict(lambda: defaultdict(list))
            with open(path) as f:
                for row in csv.DictReader(f):
                    w = workload_of(row.get("Kernel_Name", ""))
                    if w:
                        vals[w][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for w, counters in vals.items():
                for name, v in counters.items():
                    timed = v[-w[3]:]
                    means[w[0]][name] = sum(timed) / len(timed)
                    print(f"{os.path.basename(d)} {w[0]} {name}: launches={len(v)} mean_timed={means[w[0]][name]:.6g} last={v[-1]:.6g}")
    doc = {"csrc_sha256": csrc_hash(), "source": os.path.basename(out.rstrip("/")),
           "format": "per workload key of bench.py: HBM-side bytes per launch (FETCH_SIZE + WRITE_SIZE in KB x 1024, separate rocprofv3 --pmc "
                     "passes, mean of the timed launches; 64-B requests of 16-B loads: the gfx950 x2 correction for wide streaming reads is "
                     "not applied, MI355X_MICROARCH.md calls other widths uncalibrated) and issue = (SQ_ACTIVE_INST_VALU + SQ_ACTIVE_INST_SCA) "
                     "/ SQ_WAVE_CYCLES, all in quad-cycles; written by tools/pmc_to_json.py", "workloads": {}}
    for frag, key, n, trials, timed in WORKLOADS:
        m = means.get(key, {})
        if "FETCH_SIZE" not in m or "WRITE_SIZE" not in m:
            continue
        e = {"kernel": frag, "replicas": n, "trials_per_launch": trials, "hbm_bytes_per_launch": (m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0,
             "fetch_bytes": m["FETCH_SIZE"] * 1024.0, "write_bytes": m["WRITE_SIZE"] * 1024.0, "source": os.path.basename(out.rstrip("/"))}
        if all(k in m for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES")):
            e["issue"] = (m["SQ_ACTIVE_INST_VALU"] + m["SQ_ACTIVE_INST_SCA"]) / m["SQ_WAVE_CYCLES"]
        if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m:
            e["wait_any"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
        if key in stats and "timed_ms" in stats[key]:
            e["kernel_ms_trace"] = stats[key]["timed_ms"]
        if "SQ_THREAD_CYCLES_VALU" in m and m.get("SQ_ACTIVE_INST_VALU"):
            # lanes enabled per issued vector instruction (exec mask), of 64: rocprof's VALUUtilization.  (Copies of one replica in several
            # lanes count as enabled: the number of DISTINCT integrations per stream is in DESIGN.md 4.1h.)
            e["valu_lanes_enabled"] = m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 64.0)
        if key == "acrobot_q":
            e["note"] = ("counters collected with GRLX_ENV_SERVER=0 (rollout_wide_kernel integrating itself: the same table accesses); "
                         "kernel_ms_trace is rollout_wide_served_kernel beside env_server_acrobot_pinned_kernel")
        if key == "pendulum_sarsa":
            e["note"] = ("counters collected with GRLX_ENV_SERVER=0 (rollout_kernel integrating itself: the same table accesses); "
                         "kernel_ms_trace is rollout_served_kernel beside env_server_kernel" +
                         (f" ({stats['pendulum_sarsa_env_server']['timed_ms']:.4f} ms)" if "timed_ms" in stats.get("pendulum_sarsa_env_server", {}) else ""))
        doc["workloads"][key] = e
    fqi = fqi_entry(out)
    if fqi:
        doc["workloads"]["pendulum_fqi_ann"] = fqi
    print(json.dumps(doc, indent=1))
    with open(os.path.join(out, "pmc_traffic.json"), "w") as f:      # (gpurun merges gpurun_out/ back: copy this one to profiles/)
        json.dump(doc, f, indent=1)
    if write:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
