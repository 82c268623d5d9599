#!/bin/bash
# Run on the GPU box from the repo root: the TD agents' wave layouts on the acrobot and the compass walker at 16384 replicas per GPU
# (16 per SIMD): 8 replicas per wave (two rounds of waves; the acrobot beside its environment server) against 16 (four sub-batches share
# one environment phase), alternating on ONE box; and the composite with 16 per wave.
#   tools/wide16_ab.sh <tag> [workloads...]
TAG=${1:-wide16}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { # workload, name, bench args...
  local w=$1 name=$2; shift; shift
  timeout -k 10 280 python bench.py --workload $w --no-cpu-baseline "$@" > $OUT/${w}_$name.json 2> $OUT/${w}_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/${w}_$name.json").read().strip().splitlines()[-1])
    print("%-18s %-14s %8.1f M env-steps/s  kernel %.2f ms  rpw %s  %s" % ("$w", "$name", d["value"] / 1e6, d["roofline"]["kernel_ms_avg"], d.get("replicas_per_wave"), d.get("env_server")))
except Exception as e:
    print("$w $name: no bench line:", e)
PY
}
for w in "${@:-acrobot_q compass_walker_q}"; do
  for w1 in $w; do
    run $w1 n8192_rpw8 --replicas 8192 --replicas-per-wave 8
    run $w1 n16384_rpw8 --replicas 16384 --replicas-per-wave 8
    run $w1 n16384_rpw16 --replicas 16384 --replicas-per-wave 16
    run $w1 n16384_rpw8_b --replicas 16384 --replicas-per-wave 8
    run $w1 n16384_rpw16_b --replicas 16384 --replicas-per-wave 16
  done
done
