# tests of the served path, the profile passes and the bench line with the fresh counters: final_pass.sh <tag>
TAG=$1
mkdir -p gpurun_out/$TAG
GRLX_TEST_POISON=off timeout -k 10 300 python -m pytest tests/test_gpu_env_server.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/$TAG/es_tests.log 2>&1; tail -3 gpurun_out/$TAG/es_tests.log
tools/profile_passes.sh $TAG > /dev/null 2>&1; tail -2 gpurun_out/$TAG/summary.txt
cp gpurun_out/$TAG/pmc_traffic.json profiles/pmc_traffic.json && python bench.py > gpurun_out/$TAG/bench_final.json 2> gpurun_out/$TAG/bench_final.err
python - <<PY
import json
d=json.loads(open("gpurun_out/$TAG/bench_final.json").read().strip().splitlines()[-1])
print(d["value"]/1e6, d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"], d.get("env_server"))
for s in d["secondary"]: print(s["workload"], s["value"]/1e6, s["roofline"].get("traffic"))
PY
