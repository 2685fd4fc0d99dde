"""Where a pipelined rollout step goes: host time inside rollout_submit (copy enqueue + launches), inside rollout_wait (spin
until the group's actions are on the host) and in the Python between them.  python scratch/rollout_pipe.py [G] [E]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "train-procgen-pytorch_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from mi355.engine import Engine
from mi355.numa import pin_to_gpu_node
_pinned = pin_to_gpu_node(0)
print('numa pin:', None if _pinned is None else len(_pinned), 'cpus')
from mi355 import layout
from common.model import ImpalaModel
from common.policy import CategoricalPolicy

G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
E = int(sys.argv[2]) if len(sys.argv) > 2 else 256
h2d = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
T, A = 256, 15
torch.manual_seed(0)
pol = CategoricalPolicy(ImpalaModel(3), False, A)
eng = Engine("impala", T, E, A, E, precision="bf16")
eng.set_params(layout.flatten(layout.impala_param_shapes(A), {k: v.detach().numpy() for k, v in pol.state_dict().items()}))
eng.rollout_groups(G)
if flags:
    eng.debug_flags(flags)
ng = E // G
rng = np.random.default_rng(0)
fr = [[eng.pinned((ng, 64, 64, 3), np.uint8) for _ in range(4)] for _ in range(G)]
for a in fr:
    for b in a:
        b[...] = rng.integers(0, 256, size=b.shape, dtype=np.uint8)
rew = eng.pinned((T, E), np.float32); done = eng.pinned((T, E), np.float32)
rew[:] = 0.1; done[:] = 0
pc = time.perf_counter
for rep in range(3):
    ts = tw = 0.0
    t0 = pc()
    for t in range(T + 1):
        for g in range(G):
            if t:
                a = pc(); eng.rollout_wait(g); tw += pc() - a
            sl = slice(g * ng, (g + 1) * ng)
            a = pc()
            eng.rollout_submit(t, g, fr[g][t & 3] if h2d else None, rew[t - 1, sl] if t else None, done[t - 1, sl] if t else None, seed=rep)
            ts += pc() - a
    for g in range(G):
        a = pc(); eng.rollout_wait(g); tw += pc() - a
    tot = pc() - t0
    n = (T + 1) * G
    print(f"G={G} E={E} h2d={h2d}: rollout {tot * 1e3:.2f} ms = {tot / (T + 1) * 1e6:.1f} us/step ; per group-step: submit {ts / n * 1e6:.1f} us, wait {tw / n * 1e6:.1f} us, "
          f"python between {(tot - ts - tw) / n * 1e6:.1f} us")
eng.close()
