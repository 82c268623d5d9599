mkdir -p gpurun_out/es
for i in 1 2; do for L in grl_amd/lib/libgrlx_es2.so grl_amd/lib/libgrlx.so; do GRLX_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-secondary | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L  %.1f M env-steps/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"; done; done 2>&1 | tee gpurun_out/es/ab2.log
