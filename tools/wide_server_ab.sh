#!/bin/bash
# Run on the GPU box from the repo root: bench.py --workload <w> with the environment server of the wide kernels on / off and with other
# wave priorities (GRLX_ENV_SERVER_TUNE: bits 0-1 s_setprio of the rollout wave, 2-3 of the server wave), alternating on ONE box.
#   tools/wide_server_ab.sh <tag> <workload> [tune values...]
TAG=${1:-wide_ab}; W=${2:-acrobot_q}; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 250 python bench.py --workload $W --no-cpu-baseline > $OUT/${W}_$name.json 2> $OUT/${W}_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/${W}_$name.json").read().strip().splitlines()[-1])
    print("%-18s %-10s %8.1f M env-steps/s  kernel %.2f ms  %s" % ("$W", "$name", d["value"] / 1e6, d["roofline"]["kernel_ms_avg"], d.get("env_server")))
except Exception as e:
    print("$W $name: no bench line:", e)
PY
}
GRLX_ENV_SERVER_DEBUG=1 python -c "
import grl_amd, numpy as np
from grl_amd import runner
for make in (grl_amd.acrobot_q_config, grl_amd.compass_walker_q_config):
    cfg = make(64); cfg.replicas_per_wave = 8
    r = grl_amd.Runner(cfg, np.arange(1, 65)); r.run(2); r.sync(); print(r.env_server_counts()); r.close()
" 2>&1 | grep -v amdgpu.ids
run off GRLX_ENV_SERVER=0
run on GRLX_ENV_SERVER=1
for t in "$@"; do run tune$t GRLX_ENV_SERVER_TUNE=$t; done
run off2 GRLX_ENV_SERVER=0
run on2 GRLX_ENV_SERVER=1
