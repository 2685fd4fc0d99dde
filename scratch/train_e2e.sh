#!/bin/bash
# End-to-end rate of the shipped CLI on the synthetic env (logging, checkpointing, validation rollouts included): env steps/s between the
# first and the last logged iteration.  bash scratch/train_e2e.sh [extra train.py flags]
cd $GRAFT_REPO_ROOT/train-procgen-pytorch_amd
rm -rf logs/train/synthetic/e2e
python train.py --exp_name e2e --env_name synthetic --param_name hard-500 --num_timesteps 1966080 --precision bf16 --seed 1 "$@" > /tmp/e2e.log 2>&1 || { tail -5 /tmp/e2e.log; exit 1; }
python - <<'PY'
import csv, glob
f = glob.glob("logs/train/synthetic/e2e/*/log-append.csv")[0]
rows = list(csv.DictReader(open(f)))
t = [float(r["wall_time"]) for r in rows]; s = [float(r["timesteps"]) for r in rows]
print(f"{len(rows)} iterations logged; steady state {(s[-1] - s[4]) / (t[-1] - t[4]):.0f} env steps/s ({(t[-1] - t[4]) / (len(rows) - 5) * 1e3:.1f} ms per iteration of {int(s[1] - s[0])} steps)")
PY
