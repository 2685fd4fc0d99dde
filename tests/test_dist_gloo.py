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


# ------------------------------------------------------------------------------------------ the update schedule (mi355/dist.py::update_plan)
def _plan_worker(rank, world, port, out, merge, xent):
    """Plays agents/ppo.py::optimize on CPU: the plan's operations drive real gloo collectives and a recording stand-in for the
    engine (per-sample 'gradient' = one-hot of the sample index, so the all-reduced sum says which samples were visited)."""
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mi355.dist import Collective, env_range, update_plan
    coll = Collective()
    Tn, Eg, Bm, n_mini, epochs = 6, 16, 12, 2, 2                 # N = 96, batch_size = 48, 4 accumulated minibatches per step
    N = Tn * Eg
    gas = (N // n_mini) / Bm
    e0, e1 = env_range(Eg, rank, world)
    El = e1 - e0
    max_batch = 30                                                # forces a split inside an accumulation group now and then
    torch.manual_seed(21)                                         # same permutation stream on every rank

    def chunks():
        for _ in range(epochs):
            perm = torch.randperm(N).numpy()
            for k in range(N // Bm):
                yield perm[k * Bm:(k + 1) * Bm]

    grads = torch.zeros(N)                                        # 'flat gradient': visit counts of GLOBAL sample ids
    ring = torch.zeros(64)                                        # per-minibatch statistic: number of samples seen
    visits_per_step, passes, n_stats, log_n, mb_i = [], [], 0, None, 0
    for op in update_plan(chunks(), rank, world, Eg, gas, merge, coll.active and xent, max_batch, max_segments=3):
        if op[0] == "minibatch":
            _, local, seg_n, n_global = op
            assert n_global == Bm and sum(seg_n) == len(local) and len(local) <= max(max_batch, Bm) and len(seg_n) <= 3
            assert merge or len(seg_n) == 1
            t, e = local // El, local % El                        # local flat index -> global sample id
            np.add.at(grads.numpy(), t * Eg + e + e0, 1.0)
            for c in seg_n:
                ring[mb_i] += c
                mb_i += 1
            passes.append(list(seg_n))
        elif op[0] == "stats":
            s = torch.tensor([float(passes[-1][0])])
            coll.allreduce_sum_(s)
            assert s.item() == Bm                                 # the ranks' shares of ONE minibatch
            n_stats += 1
        elif op[0] == "step":
            coll.allreduce_sum_(grads)
            visits_per_step.append(grads.clone())
            grads.zero_()
        else:
            log_n = op[1]
            coll.allreduce_sum_(ring[:log_n])
    if rank == 0:
        np.savez(out, visits=torch.stack(visits_per_step).numpy(), ring=ring.numpy(), log_n=log_n, n_stats=n_stats,
                 n_passes=len(passes), max_seg=max(len(p) for p in passes))
    dist.barrier()
    dist.destroy_process_group()


def _run_plan(tmp_path, merge, xent, port):
    out = str(tmp_path / f"plan_{int(merge)}{int(xent)}.npz")
    mp.spawn(_plan_worker, args=(2, port, out, merge, xent), nprocs=2, join=True)
    return np.load(out)


def test_update_plan_two_ranks_visit_every_sample_once_per_epoch(tmp_path):
    """The accumulation groups of 4 minibatches (48 samples) between optimizer steps: whatever the schedule (one pass per minibatch,
    per-minibatch statistics exchange, or merged passes split by max_batch / max_segments), the all-reduced 'gradient' of every step
    counts each of its 48 samples exactly once, the per-minibatch statistics sum to the global minibatch size, and both ranks issue
    the same collectives (else gloo would hang / mismatch)."""
    port = 29700 + os.getpid() % 200
    base = _run_plan(tmp_path, False, False, port)
    exch = _run_plan(tmp_path, False, True, port + 1)
    merged = _run_plan(tmp_path, True, False, port + 2)
    for r in (base, exch, merged):
        v = r["visits"]
        assert v.shape == (4, 96)                                 # 2 epochs x 2 optimizer steps
        assert set(np.unique(v)) <= {0.0, 1.0} and (v.sum(1) == 48).all()
        assert (v[0] + v[1] == 1).all() and (v[2] + v[3] == 1).all()      # an epoch = a permutation of all samples
        assert int(r["log_n"]) == 16 and (r["ring"][:16] == 12).all() and (r["ring"][16:] == 0).all()
    assert np.array_equal(base["visits"], merged["visits"]) and np.array_equal(base["visits"], exch["visits"])
    assert int(base["n_passes"]) == 16 and int(base["n_stats"]) == 0 and int(exch["n_stats"]) == 16
    assert int(merged["n_passes"]) < 16 and 2 <= int(merged["max_seg"]) <= 3
