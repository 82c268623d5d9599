# Run on the GPU box from the repo root: the 2-rank rehearsal of bench.py's headline (two processes on the one GPU, gloo) N times; prints the
# devices' learning-step count of every run beside its expected value (512 replicas x 3 launches x 1000) and the server's served / fell-back counts.
#   bash tools/dbg/two_rank_probe.sh [runs]
N=${1:-4}
mkdir -p gpurun_out/two_rank_probe
for i in $(seq 1 $N); do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500+i)) bench.py --gpus 2 --backend gloo --no-cpu-baseline --steps 3 --warmup 1 --replicas 256 --no-secondary > gpurun_out/two_rank_probe/run$i.json 2> gpurun_out/two_rank_probe/run$i.err
  rc=$?
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/two_rank_probe/run$i.json").read().splitlines() if l.startswith("{")][-1])
    flag = "" if d["learn_steps"] == d["learn_steps_expected"] else "   <-- MISMATCH"
    print("run $i rc=$rc", d["learn_steps"], d["learn_steps_expected"], d["test_steps"], d.get("env_server", {}).get("replicas_served"), d.get("env_server", {}).get("replicas_fell_back"), flag)
except Exception as e:
    print("run $i rc=$rc: no line", e); print(open("gpurun_out/two_rank_probe/run$i.err").read()[-400:])
PY
done
