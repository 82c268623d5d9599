#!/bin/bash
# Run on the GPU box from the repo root: the compass walker with 16 replicas per wave at 16384 replicas against 32 per wave at 32768 (eight
# sub-batches share one environment phase), alternating on ONE box.
#   tools/wide32_ab.sh <tag>
TAG=${1:-wide32}
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { # name, bench args...
  local name=$1; shift
  timeout -k 10 290 python bench.py --workload compass_walker_q --no-cpu-baseline --steps 3 "$@" > $OUT/walker_$name.json 2> $OUT/walker_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/walker_$name.json").read().strip().splitlines()[-1])
    print("compass_walker_q %-16s %8.1f M env-steps/s  kernel %.2f ms  rpw %s" % ("$name", d["value"] / 1e6, d["roofline"]["kernel_ms_avg"], d.get("replicas_per_wave")))
except Exception as e:
    print("walker $name: no bench line:", e)
PY
}
run n16384_rpw16 --replicas 16384 --replicas-per-wave 16
run n32768_rpw32 --replicas 32768 --replicas-per-wave 32
run n32768_rpw16 --replicas 32768 --replicas-per-wave 16
run n16384_rpw16_b --replicas 16384 --replicas-per-wave 16
run n32768_rpw32_b --replicas 32768 --replicas-per-wave 32
