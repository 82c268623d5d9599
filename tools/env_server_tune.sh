#!/bin/bash
# Sweep of the environment server's experiment bits (GRLX_ENV_SERVER_TUNE, grlx_internal.h) on the headline workload: env_server_tune.sh <values...>
for T in "$@"; do
  GRLX_ENV_SERVER=1 GRLX_ENV_SERVER_TUNE=$T python bench.py --no-cpu-baseline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tune=$T  %.1f M env-steps/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"
done
