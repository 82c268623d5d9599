#!/bin/bash
# A/B of the environment server (GRLX_ENV_SERVER) on the headline workload, same box, alternating: env_server_ab.sh [rounds] [bench args...]
N=${1:-2}; shift || true
for i in $(seq 1 $N); do
  for S in 0 1; do
    GRLX_ENV_SERVER=$S python bench.py --no-cpu-baseline --no-secondary "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('server=$S  %.1f M env-steps/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"
  done
done
