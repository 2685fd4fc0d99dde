"""CPU oracle for the PPO hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch restatement (plain PyTorch-CPU fp32 + numpy) of the algorithm the
reference runs on its PPO path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module, and only as the checker /
the reported CPU baseline.  The product path (train-procgen-pytorch_amd/) never
imports it and fails loudly when the HIP library is missing.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
against tests/golden/*.npz, which tests/golden/make_golden.py produced by
running the reference itself (/root/reference, imported in the build container).

Each function cites the reference lines it follows (paths relative to the
reference root).  Tensors are torch CPU fp32 unless noted; parameters live in a
plain dict keyed by the reference's state_dict names
(embedder.block1.conv.weight ... fc_value.bias).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- observation path

def frames_to_obs(frames_u8):
    """uint8 NHWC frames -> fp32 NCHW in [0,1].

    common/env/procgen_wrappers.py:391-404 (TransposeFrame) and :407-419
    (ScaledFloatFrame: obs/255.0 in float64), then agents/ppo.py:74 /
    common/storage.py:40 cast to fp32."""
    a = np.asarray(frames_u8)
    return torch.from_numpy((a.transpose(0, 3, 1, 2) / 255.0).astype(np.float32))


# --------------------------------------------------------------------------- embedders

def _res_block(p, pre, x):
    # common/model.py:141-146  ReLU -> conv1 -> ReLU -> conv2 -> + x
    out = F.conv2d(F.relu(x), p[pre + ".conv1.weight"], p[pre + ".conv1.bias"], padding=1)
    out = F.conv2d(F.relu(out), p[pre + ".conv2.weight"], p[pre + ".conv2.bias"], padding=1)
    return out + x


def _impala_block(p, pre, x, taps=None):
    # common/model.py:156-161  conv -> MaxPool2d(3,2,1) -> res1 -> res2
    c = F.conv2d(x, p[pre + ".conv.weight"], p[pre + ".conv.bias"], padding=1)
    q = F.max_pool2d(c, kernel_size=3, stride=2, padding=1)
    y = _res_block(p, pre + ".res2", _res_block(p, pre + ".res1", q))
    if taps is not None:
        taps[pre + "_conv"], taps[pre + "_pool"], taps[pre] = c, q, y
    return y


def impala_embed(p, obs, taps=None):
    """ImpalaModel.forward_with_attn_indices (common/model.py:181-208).
    Returns (feat (B,256), flat (B,2048) post-ReLU NCHW-flattened, fs scalar)."""
    x = obs
    for b in ("embedder.block1", "embedder.block2", "embedder.block3"):
        x = _impala_block(p, b, x, taps)
    flat = torch.flatten(F.relu(x), start_dim=1)                       # model.py:195, :54-56
    feat = F.relu(F.linear(flat, p["embedder.fc.weight"], p["embedder.fc.bias"]))   # :199-200
    fs = torch.mean(torch.max(torch.tanh(torch.abs(flat * 100)), 0)[0])            # :207
    return feat, flat, fs


def mlp_embed(p, x):
    """MLPModel.forward (common/model.py:954-980): Linear/ReLU stack, no final ReLU.
    nn.Sequential naming: embedder.model.0, embedder.model.2.{0,2,..}, embedder.model.3."""
    h = F.relu(F.linear(x, p["embedder.model.0.weight"], p["embedder.model.0.bias"]))
    i = 0
    while f"embedder.model.2.{i}.weight" in p:
        h = F.relu(F.linear(h, p[f"embedder.model.2.{i}.weight"], p[f"embedder.model.2.{i}.bias"]))
        i += 2
    return F.linear(h, p["embedder.model.3.weight"], p["embedder.model.3.bias"])


def heads(p, feat):
    """CategoricalPolicy.hidden_to_output + distribution (common/policy.py:74-87).
    Returns (logp_all (B,A) = Categorical(logits=log_softmax(.)).logits, value (B,))."""
    logits = F.linear(feat, p["fc_policy.weight"], p["fc_policy.bias"])
    lp = F.log_softmax(logits, dim=1)
    lp = lp - lp.logsumexp(dim=-1, keepdim=True)     # Categorical.__init__ renormalises its logits
    value = F.linear(feat, p["fc_value.weight"], p["fc_value.bias"]).reshape(-1)
    return lp, value


def gru_cell(p, x, h, mask):
    """GRU.forward prediction branch (common/model.py:219-225): one nn.GRU step on h * mask.
    p holds gru.gru.{weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0}; gate order r, z, n."""
    hm = h * mask.reshape(-1, 1)
    gi = F.linear(x, p["gru.gru.weight_ih_l0"], p["gru.gru.bias_ih_l0"])
    gh = F.linear(hm, p["gru.gru.weight_hh_l0"], p["gru.gru.bias_hh_l0"])
    H = h.shape[1]
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * hm


def policy_forward(p, arch, obs):
    """-> (logp_all, value, fs or None)."""
    if arch == "impala":
        feat, _, fs = impala_embed(p, obs)
    else:
        feat, fs = mlp_embed(p, obs), None
    lp, v = heads(p, feat)
    return lp, v, fs


def sample_actions(logp_all, u):
    """Inverse-CDF categorical sampling from uniforms u in [0,1).  This is the BUILD's rollout
    sampler definition (the reference draws torch.multinomial from the torch generator,
    agents/ppo.py:78 -- not bit-reproducible across devices; parity is teacher-forced).
    act = first a with cumsum(p)[a] > u, clamped to A-1.  logp = logp_all[act] (ppo.py:79)."""
    prob = torch.exp(logp_all)
    cdf = torch.cumsum(prob, dim=1)
    act = (cdf <= u.reshape(-1, 1)).sum(dim=1).clamp(max=logp_all.shape[1] - 1)
    return act, logp_all.gather(1, act.reshape(-1, 1)).reshape(-1)


# --------------------------------------------------------------------------- GAE / returns

def compute_estimates(rew, done, value, gamma, lmbda, use_gae=True, normalize_adv=True):
    """Storage.compute_estimates (common/storage.py:56-79).  rew/done (T,E), value (T+1,E).
    Returns (adv (T,E), ret (T,E)).  use_gae=False reproduces the reference's overwrite:
    return = adv(=0) + V (storage.py:68-77)."""
    T = rew.shape[0]
    adv = torch.zeros_like(rew)
    if use_gae:
        A = 0
        for i in reversed(range(T)):
            delta = (rew[i] + gamma * value[i + 1] * (1 - done[i])) - value[i]
            adv[i] = A = gamma * lmbda * A * (1 - done[i]) + delta
    ret = adv + value[:-1]
    if normalize_adv:
        adv = (adv - torch.mean(adv)) / (torch.std(adv) + 1e-8)      # unbiased std over all T*E
    return adv, ret


def compute_estimates_np(rew, done, value, gamma, lmbda):
    """numpy fp32 twin of the GAE scan above (no normalisation) -- independent arithmetic path."""
    rew, done, value = (np.asarray(a, dtype=np.float32) for a in (rew, done, value))
    g, gl, one = np.float32(gamma), np.float32(gamma * lmbda), np.float32(1)
    T = rew.shape[0]
    adv = np.zeros_like(rew)
    A = np.zeros(rew.shape[1], dtype=np.float32)
    for i in range(T - 1, -1, -1):
        nd = one - done[i]
        delta = (rew[i] + g * value[i + 1] * nd) - value[i]
        A = gl * A * nd + delta
        adv[i] = A
    return adv, adv + value[:-1]


# --------------------------------------------------------------------------- minibatch index streams

def minibatch_indices(n_total, mini_batch_size):
    """One epoch of Storage.fetch_train_generator(recurrent=False) (common/storage.py:86-91):
    BatchSampler(SubsetRandomSampler(range(N)), B, drop_last=True) == ONE torch.randperm(N) on the
    global CPU generator, consecutive chunks of B, remainder dropped.  Flat index i = t*E + e."""
    perm = torch.randperm(n_total)
    nb = n_total // mini_batch_size
    return [perm[k * mini_batch_size:(k + 1) * mini_batch_size].numpy().astype(np.int64) for k in range(nb)]


def recurrent_env_batches(n_steps, n_envs, mini_batch_size):
    """recurrent=True branch (common/storage.py:93-110): perm = randperm(E); groups of
    E // (N // B) envs; a minibatch is all T steps of those envs, time-major."""
    n = n_steps * n_envs
    per = n_envs // (n // mini_batch_size)
    perm = torch.randperm(n_envs)
    return [perm[s:s + per].numpy().astype(np.int64) for s in range(0, n_envs, per)]


# --------------------------------------------------------------------------- loss

def ppo_loss(logp_all, value, act, old_logp, old_value, ret, adv, eps_clip=0.2, value_coef=0.5,
             entropy_coef=0.01, x_entropy_coef=0.0, entropy_multiplier=1.0, fs=None, fs_coef=0.0):
    """agents/ppo.py:131-169 + cross_batch_entropy (common/misc_util.py:42-51).
    Returns dict(pi_loss, value_loss, entropy, x_ent, total)."""
    logp = logp_all.gather(1, act.long().reshape(-1, 1)).reshape(-1)           # dist.log_prob(act)
    ratio = torch.exp(logp - old_logp)
    surr1 = ratio * adv
    surr2 = torch.clamp(ratio, 1.0 - eps_clip, 1.0 + eps_clip) * adv
    pi_loss = -torch.min(surr1, surr2).mean()
    clipped = old_value + (value - old_value).clamp(-eps_clip, eps_clip)
    value_loss = 0.5 * torch.max((value - ret).pow(2), (clipped - ret).pow(2)).mean()
    prob = torch.softmax(logp_all, dim=-1)                                      # Categorical.probs
    cond = (-(prob * logp_all).sum(-1)).mean()
    q = prob.mean(0)
    marg = -(q * torch.log(q)).sum()
    x_ent = marg - cond
    total = (pi_loss + value_coef * value_loss - entropy_coef * cond * entropy_multiplier
             - x_entropy_coef * x_ent)
    if fs is not None:
        total = total + fs_coef * fs
    return dict(pi_loss=pi_loss, value_loss=value_loss, entropy=cond, x_ent=x_ent, total=total)


# --------------------------------------------------------------------------- optimiser

def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (agents/ppo.py:174): L2 over all grads,
    coef = min(1, max_norm / (norm + 1e-6)).  In place.  Returns (norm, coef)."""
    norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return float(norm), float(coef)


def adam_step(params, grads, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-5):
    """optim.Adam(lr, eps=1e-5) single step (agents/ppo.py:58,175), torch's update order:
    m.lerp_(g, 1-b1); v = b2 v + (1-b2) g^2; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = lr / bc1
    bc2_sqrt = math.sqrt(bc2)
    for k in params:
        g = grads[k]
        m[k].lerp_(g, 1 - beta1)
        v[k].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v[k].sqrt() / bc2_sqrt).add_(eps)
        params[k].addcdiv_(m[k], denom, value=-step_size)


class OraclePPO:
    """PPO.optimize (agents/ppo.py:96-208) over a stored rollout, on the restated pieces above.

    rollout: dict(obs (T+1,E,...) fp32 in the reference's layout, act, rew, done, logp (T,E),
    val (T+1,E)).  Keeps the reference's quirks: unscaled gradient accumulation (ppo.py:170-177),
    sign-flipped logged pi/value losses (:178-179), mini_batch_size = min(given, T*E//n_minibatch).
    """

    def __init__(self, params, arch, n_steps, n_envs, epoch=3, n_minibatch=8, mini_batch_size=256,
                 gamma=0.99, lmbda=0.95, learning_rate=2.5e-4, grad_clip_norm=0.5, eps_clip=0.2,
                 value_coef=0.5, entropy_coef=0.01, x_entropy_coef=0.0, normalize_adv=True,
                 use_gae=True, fs_coef=0.0, **_ignored):
        self.p = OrderedDict((k, torch.as_tensor(np.array(v, dtype=np.float32)).clone()) for k, v in params.items())
        self.arch = arch
        self.T, self.E = n_steps, n_envs
        self.hp = dict(epoch=epoch, n_minibatch=n_minibatch, mini_batch_size=mini_batch_size, gamma=gamma,
                       lmbda=lmbda, lr=learning_rate, clip=grad_clip_norm, eps_clip=eps_clip,
                       value_coef=value_coef, entropy_coef=entropy_coef, x_entropy_coef=x_entropy_coef,
                       normalize_adv=normalize_adv, use_gae=use_gae, fs_coef=fs_coef)
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in self.p.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in self.p.items())
        self.step = 0
        self.lr = learning_rate
        self.grad_log = []
        self.param_norm_log = []

    def loss_and_grads(self, obs, act, old_logp, old_value, ret, adv):
        for t in self.p.values():
            t.requires_grad_(True)
            t.grad = None
        lp, value, fs = policy_forward(self.p, self.arch, obs)
        hp = self.hp
        L = ppo_loss(lp, value, act, old_logp, old_value, ret, adv, hp["eps_clip"], hp["value_coef"],
                     hp["entropy_coef"], hp["x_entropy_coef"], 1.0, fs, hp["fs_coef"])
        L["total"].backward()
        grads = OrderedDict((k, t.grad.detach().clone()) for k, t in self.p.items())
        for t in self.p.values():
            t.requires_grad_(False)
            t.grad = None
        out = {k: float(x.detach()) for k, x in L.items()}
        out["fs"] = float(fs.detach()) if fs is not None else float("nan")
        return out, grads

    def optimize(self, ro, index_stream=None):
        hp, T, E = self.hp, self.T, self.E
        N = T * E
        batch_size = N // hp["n_minibatch"]
        B = min(hp["mini_batch_size"], batch_size)
        acc_steps = batch_size / B
        obs = torch.as_tensor(ro["obs"][:-1], dtype=torch.float32).reshape(N, *ro["obs"].shape[2:])
        flat = {k: torch.as_tensor(ro[k], dtype=torch.float32).reshape(-1) for k in ("act", "logp", "ret", "adv")}
        oldv = torch.as_tensor(ro["val"][:-1], dtype=torch.float32).reshape(-1)
        acc = None
        cnt = 1
        logs = {k: [] for k in ("pi_loss", "value_loss", "entropy", "x_ent", "total", "fs")}
        for e in range(hp["epoch"]):
            chunks = minibatch_indices(N, B) if index_stream is None else index_stream[e]
            for idx in chunks:
                idx = torch.as_tensor(idx)
                L, g = self.loss_and_grads(obs[idx], flat["act"][idx], flat["logp"][idx], oldv[idx],
                                           flat["ret"][idx], flat["adv"][idx])
                acc = g if acc is None else OrderedDict((k, acc[k] + g[k]) for k in g)
                if cnt % acc_steps == 0:
                    self.grad_log.append(OrderedDict((k, t.clone()) for k, t in acc.items()))
                    clip_grad_norm(acc, hp["clip"])
                    self.step += 1
                    adam_step(self.p, acc, self.m, self.v, self.step, self.lr)
                    self.param_norm_log.append({k: float(torch.sqrt((t.double() ** 2).sum()))
                                                for k, t in self.p.items()})
                    acc = None
                cnt += 1
                for k in logs:
                    logs[k].append(L[k])
        return {"Loss/pi": -np.mean(logs["pi_loss"]), "Loss/v": -np.mean(logs["value_loss"]),
                "Loss/entropy": np.mean(logs["entropy"]), "Loss/x_entropy": np.mean(logs["x_ent"]),
                "Loss/feature_sparsity": np.mean(logs["fs"]), "Loss/total": np.mean(logs["total"])}
