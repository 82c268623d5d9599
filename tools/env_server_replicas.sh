#!/bin/bash
# The headline launch with and without the environment server at 4096 / 2048 / 1024 replicas: with fewer replicas than SIMD slots the
# server and the rollout waves land on different SIMDs, which shows what the contention costs at 4096 (profiles/r03_env_server_ab.md, 4).
mkdir -p gpurun_out/es
for R in 4096 2048 1024; do for S in 0 1; do GRLX_ENV_SERVER=$S python bench.py --no-cpu-baseline --no-secondary --replicas $R | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('replicas=$R server=$S  %.1f M env-steps/s  %.3f ms  %s' % (d['value']/1e6, d['ms_per_step'], d.get('env_server')))"; done; done 2>&1 | tee gpurun_out/es/half.log
