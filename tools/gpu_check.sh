#!/bin/bash
# Run on the GPU box from the repo root: the GPU suite, then the default bench; a step that timed out stops the rest.
#   tools/gpu_check.sh <tag> [pytest args...]
TAG=${1:-check}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q "$@" > $OUT/tests.log 2>&1
rc=$?
echo "tests rc=$rc" | tee -a $OUT/tests.log
tail -4 $OUT/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: no further GPU step"; exit $rc; fi
timeout -k 10 ${BENCH_TIMEOUT:-600} python bench.py ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err
brc=$?
echo "bench rc=$brc"
tail -c 1200 $OUT/bench.err
python3 - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print("headline %.1f M env-steps/s, frac %.4f" % (d["value"] / 1e6, d["roofline"]["frac"]))
    for s in d.get("secondary", []):
        print("  %-18s %10.2f M %s  kernel_ms %s frac %.4f" % (s["workload"], s["value"] / 1e6, s["unit"], s["roofline"].get("kernel_ms_avg", s["roofline"].get("kernel_ms_total")), s["roofline"]["frac"]))
except Exception as e:
    print("no bench line:", e)
PY
exit $(( rc != 0 ? rc : brc ))
