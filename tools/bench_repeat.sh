#!/bin/bash
# bench.py several times, value and ms per run: bench_repeat.sh <count> [bench args...]
n=$1; shift
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f M env-steps/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"
done
