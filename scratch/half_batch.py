"""Does a minibatch in SPLIT accumulated passes run faster than in one (a 16-channel 32x32 activation of 8192 samples is 268 MB, more
than the 256 MB memory-side cache; of 4096 samples 134 MB)?  python scratch/half_batch.py <splits...>  -> ms per 8192-sample update"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "train-procgen-pytorch_amd")]
import torch
from mi355.engine import Engine, F_ACT, F_LOGP, F_VALUE, F_REW, F_DONE
from mi355 import layout
from common.model import ImpalaModel
from common.policy import CategoricalPolicy
T, E, A, B = 64, 256, 15, 8192
torch.manual_seed(6033)
pol = CategoricalPolicy(ImpalaModel(3), False, A)
eng = Engine("impala", T, E, A, B, precision="bf16")
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
perms = [rng.permutation(T * E)[:B] for _ in range(12)]
step = 0
for rep in range(2):
    for S in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
        for timed in (False, True):
            eng.sync(); t0 = time.perf_counter()
            for p in perms:
                for q in range(S):
                    eng.minibatch(p[q * B // S:(q + 1) * B // S], B, hp)
                step += 1
                eng.optimizer_step(5e-4, 0.5, step)
            eng.sync(); dt = time.perf_counter() - t0
        print("passes per minibatch %d: %.3f ms per 8192-sample update" % (S, dt / len(perms) * 1e3), flush=True)
eng.loss_log()
