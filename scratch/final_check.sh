#!/bin/bash
# end-of-round check on the GPU box: full -m gpu suite, then the default bench line
python -m pytest tests -x -q -m gpu > gpurun_out/t_all.log 2>&1; tail -2 gpurun_out/t_all.log
python bench.py 2>gpurun_out/final_bench.err | tail -1 > gpurun_out/final_bench.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/final_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["phase_ms_per_step"]["rollout"], d["phase_ms_per_step"]["update"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["fp32"]["value"], d["cpu_baseline"]["value"])
PY
