"""world_size-2 `gloo` test (CPU) of the data-parallel scheme of SURVEY 8(e): same global permutation on every
rank, env-sharded indices, partial sums scaled by 1/B_global, ONE summed all-reduce of the flat gradient per
optimizer step, Chan-merged advantage statistics.  The per-rank arithmetic is played by the CPU oracle (test
infrastructure); what is under test is the host logic in mi355/dist.py that the agent uses unchanged on RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT, load_npz, npz_params

T, E, A, B = 8, 8, 2, 16


def _rollout(seed=3):
    rng = np.random.default_rng(seed)
    return dict(obs=rng.standard_normal((T + 1, E, 9)).astype(np.float32), act=rng.integers(0, A, (T, E)),
                rew=rng.standard_normal((T, E)).astype(np.float32), done=(rng.random((T, E)) < 0.1).astype(np.float32),
                logp=(np.log(0.5) + 0.1 * rng.standard_normal((T, E))).astype(np.float32),
                val=rng.standard_normal((T + 1, E)).astype(np.float32))


def _partial_loss(O, p, obs, act, logp_old, v_old, ret, adv, inv_b):
    lp, v, _ = O.policy_forward(p, "mlp", obs)
    logp = lp.gather(1, act.long().reshape(-1, 1)).reshape(-1)
    ratio = torch.exp(logp - logp_old)
    pi = torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).sum()
    vc = v_old + (v - v_old).clamp(-0.2, 0.2)
    vl = torch.max((v - ret) ** 2, (vc - ret) ** 2).sum()
    ent = (-(torch.softmax(lp, -1) * lp).sum(-1)).sum()
    return (-pi + 0.5 * 0.5 * vl - 0.01 * ent) * inv_b


def _worker(rank, world, port, out):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ppo_oracle as O
    from mi355.dist import Collective, env_range, merge_adv_stats, shard_indices
    torch.set_num_threads(1)
    coll = Collective()
    assert coll.active and coll.world == world and coll.rank == rank
    ro = _rollout()
    e0, e1 = env_range(E, rank, world)
    El = e1 - e0
    # local GAE on the env shard, global normalisation through merged statistics
    adv_l, ret_l = O.compute_estimates(torch.from_numpy(ro["rew"][:, e0:e1]), torch.from_numpy(ro["done"][:, e0:e1]),
                                       torch.from_numpy(ro["val"][:, e0:e1]), 0.99, 0.95, True, False)
    a64 = adv_l.double().numpy().ravel()
    stats = coll.allgather_f64([a64.size, a64.mean(), ((a64 - a64.mean()) ** 2).sum()])
    n, mean, m2 = merge_adv_stats(stats)
    adv_l = (adv_l - np.float32(mean)) / (np.float32(np.sqrt(m2 / (n - 1))) + 1e-8)
    params = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in npz_params(load_npz("g7_mlp_forward.npz")).items()}
    torch.manual_seed(11)                                   # same stream on every rank
    chunk = torch.randperm(T * E).numpy()[:B]
    loc = shard_indices(chunk, E, rank, world)
    t, e = loc // El, loc % El
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    loss = _partial_loss(O, params, f(ro["obs"][:-1, e0:e1][t, e]), f(ro["act"][:, e0:e1][t, e]), f(ro["logp"][:, e0:e1][t, e]),
                         f(ro["val"][:-1, e0:e1][t, e]), ret_l[t, e], adv_l[t, e], 1.0 / B)
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in params.values()])
    coll.allreduce_sum_(flat)                                # the ONE gradient collective
    lsum = torch.tensor([float(loss)], dtype=torch.float64)
    coll.allreduce_sum_(lsum)
    if rank == 0:
        np.savez(out, grad=flat.numpy(), loss=lsum.numpy(), n_local=len(loc))
    dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process(tmp_path):
    from oracle import ppo_oracle as O
    out = str(tmp_path / "r0.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    # single-process reference on the full minibatch (means over B)
    ro = _rollout()
    adv, ret = O.compute_estimates(torch.from_numpy(ro["rew"]), torch.from_numpy(ro["done"]), torch.from_numpy(ro["val"]), 0.99, 0.95, True, True)
    torch.manual_seed(11)
    chunk = torch.randperm(T * E).numpy()[:B]
    ag = O.OraclePPO(npz_params(load_npz("g7_mlp_forward.npz")), "mlp", T, E, epoch=1, n_minibatch=1, mini_batch_size=B)
    ti = torch.from_numpy(chunk)
    f = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(-1))[ti]
    L, g = ag.loss_and_grads(torch.from_numpy(ro["obs"][:-1].reshape(T * E, 9))[ti], f(ro["act"]), f(ro["logp"]), f(ro["val"][:-1]),
                             ret.reshape(-1)[ti], adv.reshape(-1)[ti])
    ref = torch.cat([v.reshape(-1) for v in g.values()]).numpy()
    assert 0 < int(got["n_local"]) < B
    assert abs(float(got["loss"][0]) - L["total"]) < 1e-5
    np.testing.assert_allclose(got["grad"], ref, rtol=1e-4, atol=2e-6)
