#!/bin/bash
# A/B timing of two builds of libgrlx.so on the SAME GPU box (box-to-box spread is ~6 %, run-to-run on one
# box ~0.1 %): ab_bench.sh <libA.so> <libB.so> [rounds] [bench args...]   -- alternates A, B, A, B, ...
A=$1; B=$2; N=${3:-3}; shift; shift; shift || true
for i in $(seq 1 $N); do
  for L in "$A" "$B"; do
    GRLX_LIB=$(realpath $L) python bench.py --no-cpu-baseline "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L  %.1f M env-steps/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"
  done
done
