"""Policy step issued eagerly vs as a replayed hipGraph of the same launches (mi_debug_step_latency): microseconds per step incl. the stream wait."""
import sys, numpy as np
sys.path[:0] = [".", "train-procgen-pytorch_amd"]
from mi355.engine import Engine
from mi355 import layout
for prec in ("bf16", "fp32"):
    for E in (64, 256):
        eng = Engine("impala", 2, E, 9, E, precision=prec)
        eng.set_params(np.random.default_rng(0).standard_normal(eng.n_params).astype(np.float32) * 0.05)
        fr = np.random.default_rng(1).integers(0, 256, (E, 64, 64, 3), dtype=np.uint8)
        for t in range(3):
            eng.put_obs(t, fr)
        res = []
        for rep in range(3):
            res.append((eng.debug_step_latency(1, 300, False), eng.debug_step_latency(1, 300, True)))
        print(prec, "E =", E, " eager / graph us per step:", " ".join(f"{a:.1f}/{b:.1f}" for a, b in res))
        eng.close()
