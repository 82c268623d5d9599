#!/bin/bash
# Cart-pole actor-critic throughput by replicas per GPU and slots per wave (bench.py --workload cart_pole_ac --replicas N --replicas-per-wave R),
# one box, one call: the table of DESIGN.md section 4.1d (8 / 12 rotated / 16 slots).
for cfg in "8192 8" "12288 12" "16384 8" "16384 12" "24576 12" "16384 16"; do set -- $cfg; timeout -k 10 200 python bench.py --workload cart_pole_ac --no-cpu-baseline --replicas $1 --replicas-per-wave $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('replicas=$1 rpw=$2  %.1f M env-steps/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"; done
