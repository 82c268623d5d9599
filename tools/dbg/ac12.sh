mkdir -p gpurun_out/ac12
GRLX_TEST_POISON=off timeout -k 10 400 python -m pytest tests/test_gpu_generic_paths.py -x -q -m gpu -k "wide_waves_actor_critic" > gpurun_out/ac12/test.log 2>&1; tail -15 gpurun_out/ac12/test.log
for R in 8 12; do timeout -k 10 200 python bench.py --workload cart_pole_ac --no-cpu-baseline --replicas-per-wave $R 2>gpurun_out/ac12/err$R.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rpw=$R  %.1f M env-steps/s  %.3f ms rpw=%s' % (d['value']/1e6, d['ms_per_step'], d.get('replicas_per_wave')))"; done
