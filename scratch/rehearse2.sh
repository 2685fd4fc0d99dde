#!/bin/bash
# the 2-rank bench path on ONE GPU (gloo collectives, both ranks share the card): bash scratch/rehearse2.sh
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --rehearse-on-one-gpu --no-cpu-baseline --no-fp32-record 2>gpurun_out/reh.err | tail -1 > gpurun_out/reh.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/reh.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","n_gpus","ms_per_step","scaling")}, d["phase_ms_per_step"])
PY
tail -3 gpurun_out/reh.err
