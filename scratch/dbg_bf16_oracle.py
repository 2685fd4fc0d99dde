"""Per-tensor relative L2 error of the bf16 engine's gradients against oracle/ppo_oracle_bf16.py (and against the fp32 oracle for scale)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "train-procgen-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
torch.set_num_threads(16)
import test_gpu_fullsize as F
from oracle import ppo_oracle_bf16 as OB, ppo_oracle as O
from mi355 import engine as M, layout
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ro = F._rollout(2)
eng, shapes, params = F._engine(ro, "bf16", max(n, 256))
adv, ret = eng.read_field(M.F_ADV), eng.read_field(M.F_RET)
idx = np.random.default_rng(3).permutation(F.T * F.E)[:n]
eng.minibatch(idx, n, eng.hparams())
mine = layout.unflatten(shapes, eng.get_grads())
eng.close()
f = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(-1)[idx])
fr = ro["frames"][:-1].reshape(-1, 64, 64, 3)[idx]
args = (f(ro["act"]), f(ro["logp"]), f(ro["val"][:-1]), f(ret), f(adv))
L, g = OB.loss_and_grads(params, fr, *args)
L0, g0 = OB.loss_and_grads(params, fr, *args, rounding=False)
for k in g:
    print(f"{k:36s} |g| {float(g[k].norm()):.3e}  engine vs bf16-oracle {F._rel(mine[k], g[k].numpy()):.2e}   bf16-oracle vs fp32 {F._rel(g[k].numpy(), g0[k].numpy()):.2e}   engine vs fp32 {F._rel(mine[k], g0[k].numpy()):.2e}")
