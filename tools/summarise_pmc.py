"""Summarise rocprofv3 output directories written by tools/profile_passes.sh:
mean per launch of every counter for the rollout kernel, and the kernel-trace stats line."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for path in sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            if "rollout" in row.get("Name", ""):
                print("stats:", {k: row[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs") if k in row})
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        vals = defaultdict(list)
        with open(path) as f:
            for row in csv.DictReader(f):
                if "rollout" in row.get("Kernel_Name", ""):
                    vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for name, v in vals.items():
            timed = v[3:] if len(v) > 3 else v        # skip the 3 warm-up launches
            print(f"{os.path.basename(d)} {name}: launches={len(v)} mean_timed={sum(timed) / len(timed):.6g} last={v[-1]:.6g}")
