# A/B of two libraries over every rollout workload: ab_all.sh libA libB
mkdir -p gpurun_out/ab
for L in "$1" "$2" "$1" "$2"; do GRLX_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-fqi --no-composite | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$L  headline %.1f' % (d['value']/1e6), ' '.join('%s %.1f' % (s['workload'], s['value']/1e6) for s in d['secondary']))"; done 2>&1 | tee gpurun_out/ab/ab_all.log
