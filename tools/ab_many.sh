#!/bin/bash
# one bench run per library, in the order given, repeated: ab_many.sh <rounds> lib1.so lib2.so ...
N=$1; shift
for i in $(seq 1 $N); do
  for L in "$@"; do
    GRLX_LIB=$(realpath $L) python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L  %.1f M env-steps/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"
  done
done
