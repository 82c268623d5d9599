#!/bin/bash
# A/B of two builds on one box for another configuration: ab_config.sh libA.so libB.so <rounds> <run_config.py args...>
A=$1; B=$2; N=$3; shift; shift; shift
for i in $(seq 1 $N); do
  for L in "$A" "$B"; do
    echo -n "$L  "; GRLX_LIB=$(realpath $L) python tools/run_config.py "$@"
  done
done
