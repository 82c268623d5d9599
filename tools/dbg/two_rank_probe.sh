for i in 1 2 3 4; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500+i)) bench.py --gpus 2 --backend gloo --no-cpu-baseline --steps 3 --warmup 1 --replicas 256 --no-secondary > gpurun_out/r04_af/run$i.json 2> gpurun_out/r04_af/run$i.err
  echo "run $i rc=$?"
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r04_af/run$i.json").read().splitlines() if l.startswith("{")][-1])
    print(d["learn_steps"], d["test_steps"], d.get("env_server"))
except Exception as e:
    print("no line", e); print(open("gpurun_out/r04_af/run$i.err").read()[-600:])
PY
done
