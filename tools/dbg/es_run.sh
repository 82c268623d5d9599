mkdir -p gpurun_out/es
GRLX_TEST_POISON=off timeout -k 10 400 python -m pytest tests/test_gpu_env_server.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/es/parity.log 2>&1; echo parity rc=$?; tail -3 gpurun_out/es/parity.log
for L in grl_amd/lib/libgrlx_es1.so grl_amd/lib/libgrlx.so grl_amd/lib/libgrlx_es1.so grl_amd/lib/libgrlx.so; do GRLX_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L  %.1f M env-steps/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"; done 2>&1 | tee gpurun_out/es/ab.log
timeout -k 10 300 tools/env_server_tune.sh 3 7 16 2>&1 | tee gpurun_out/es/tune.log
GRLX_LIB=$PWD/grl_amd/lib/libgrlx_stats.so timeout -k 10 120 python tools/env_server_stats.py 2>&1 | tee gpurun_out/es/stats.log
