"""Rollout phase alone (257 policy steps at E = 256, bf16) for a kernel trace: python scratch/rollout_only.py [iters]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "train-procgen-pytorch_amd")]
from mi355.engine import Engine
T, E = 256, 256
eng = Engine("impala", T, E, 15, 8192, precision="bf16")
rng = np.random.default_rng(0)
stage = eng.pinned((E, 64, 64, 3), np.uint8)
for t in range(T + 1):
    stage[...] = rng.integers(0, 256, size=stage.shape, dtype=np.uint8)
    eng.put_obs(t, stage); eng.sync()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
USE_RD = len(sys.argv) > 2 and sys.argv[2] == 'rd'
RD = (np.zeros(E, np.float32), np.zeros(E, np.float32))
for it in range(n):
    t0 = time.perf_counter()
    for t in range(T + 1):
        eng.rollout_step(t, (RD[0] if t and USE_RD else None), (RD[1] if t and USE_RD else None), seed=it)
    print("rollout ms", (time.perf_counter() - t0) * 1e3, flush=True)
