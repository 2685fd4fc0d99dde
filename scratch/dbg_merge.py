"""Diagnostic: one-by-one vs merged gradient accumulation (test_accumulated_minibatches_in_one_pass_equal_one_by_one):
per optimizer step, which tensors differ and by how much (gradients and parameters)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "train-procgen-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from test_gpu_agent import _impala_agent
from mi355 import engine as M, layout

precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
T, E, A = 4, 16, 15
rng = np.random.default_rng(3)
frames = rng.integers(0, 256, size=(T + 1, E, 64, 64, 3), dtype=np.uint8)
act = rng.integers(0, A, (T, E)); logp = (np.log(1 / A) + 0.2 * rng.standard_normal((T, E))).astype(np.float32)
val = rng.standard_normal((T + 1, E)).astype(np.float32); rew = rng.standard_normal((T, E)).astype(np.float32)
done = (rng.random((T, E)) < 0.2).astype(np.float32)
shapes = layout.impala_param_shapes(A)
snaps = []
keep = []
for merge in (False, True):
    agent, policy, storage = _impala_agent(T, E, 8, epoch=2, n_minibatch=2, precision=precision, merge_accumulation=merge)
    keep.append(agent)
    eng = agent.engine
    for t in range(T + 1):
        eng.put_obs(t, frames[t]); eng.sync()
    eng.write_field(M.F_ACT, act.astype(np.float32)); eng.write_field(M.F_LOGP, logp); eng.write_field(M.F_VALUE, val)
    eng.write_field(M.F_REW, rew); eng.write_field(M.F_DONE, done)
    storage.compute_estimates(0.999, 0.95, True, True)
    torch.manual_seed(5)
    rec = []
    orig = agent.optimizer.step
    def step(clip, orig=orig, eng=eng, rec=rec, **kw):
        g = eng.get_grads()
        orig(clip, **kw)
        rec.append((g, eng.get_params()))
    agent.optimizer.step = step
    agent.optimize()
    snaps.append(rec)
for s, ((g0, p0), (g1, p1)) in enumerate(zip(*snaps)):
    G0, G1, P0, P1 = (layout.unflatten(shapes, x) for x in (g0, g1, p0, p1))
    print(f"step {s + 1}: |g| {np.linalg.norm(g0):.4f}  max|dg| {np.abs(g1 - g0).max():.3e}  max|dp| {np.abs(p1 - p0).max():.3e}")
    for k in G0:
        dg, dp = np.abs(G1[k] - G0[k]).max(), np.abs(P1[k] - P0[k]).max()
        if dp > 1e-6 or dg > 1e-5 * (np.abs(G0[k]).max() + 1e-12):
            print(f"    {k:34s} |g|max {np.abs(G0[k]).max():.3e} dg {dg:.3e} dp {dp:.3e} n(dp>4e-6) {(np.abs(P1[k] - P0[k]) > 4e-6).sum()}")
