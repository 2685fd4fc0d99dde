"""Two 8192-sample minibatch updates (forward, loss, backward, clip+Adam) of the hard-500 shape: the workload the
rocprofv3 --pmc passes are collected on (counters serialise kernels; the full bench would take too long)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "train-procgen-pytorch_amd")]
import torch
from mi355.engine import Engine, F_ACT, F_LOGP, F_VALUE, F_REW, F_DONE
from mi355 import layout
from common.model import ImpalaModel
from common.policy import CategoricalPolicy
T, E, A, B = 32, 256, 15, 8192
torch.manual_seed(6033)
pol = CategoricalPolicy(ImpalaModel(3), False, A)
PREC = sys.argv[2] if len(sys.argv) > 2 else "fp32"
eng = Engine("impala", T, E, A, B, precision=PREC)
eng.set_params(layout.flatten(layout.impala_param_shapes(A), {k: v.detach().numpy() for k, v in pol.state_dict().items()}))
rng = np.random.default_rng(0)
for t in range(T + 1):
    eng.put_obs(t, rng.integers(0, 256, size=(E, 64, 64, 3), dtype=np.uint8)); eng.sync()
eng.write_field(F_ACT, rng.integers(0, A, (T, E)).astype(np.float32))
eng.write_field(F_LOGP, np.full((T, E), np.log(1 / A), np.float32))
eng.write_field(F_VALUE, rng.standard_normal((T + 1, E)).astype(np.float32))
eng.write_field(F_REW, rng.standard_normal((T, E)).astype(np.float32)); eng.write_field(F_DONE, np.zeros((T, E), np.float32))
eng.compute_estimates(0.999, 0.95)
hp = eng.hparams()
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    eng.minibatch(rng.permutation(T * E)[:B], B, hp)
    eng.optimizer_step(5e-4, 0.5, k + 1)
eng.sync()
print("done")
