#!/usr/bin/env python3
"""Generate the golden vectors G1..G8 by RUNNING the reference.

This script is the only place in the repo that imports /root/reference.  It runs
in the build container only (the reference never travels to the GPU box); what
it writes -- small .npz/.json fixtures of inputs and expected outputs -- is data,
committed next to it, and is what tests/ and oracle/ are pinned against.

Import recipe: common/storage.py needs torch+numpy only;
agents.ppo / common.policy / common.model need three third-party module *names*
that are absent offline (gym, gymnasium, vector_quantize_pytorch).  None of the
symbols taken from them is executed on the PPO path, so empty module objects
are registered for the names before the import.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Every fixture carries the inputs it was produced from, so the tests never need
the reference again.
"""
import hashlib
import json
import os
import sys
import types
import zlib

import numpy as np

sys.dont_write_bytecode = True
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    _stub("gym").logger = types.SimpleNamespace(set_level=lambda lvl: None)
    sp = _stub("gymnasium.spaces", Box=type("Box", (), {}), Discrete=type("Discrete", (), {}))
    _stub("gymnasium").spaces = sp
    _stub("vector_quantize_pytorch",
          VectorQuantize=type("VectorQuantize", (nn.Module,), {}),
          FSQ=type("FSQ", (nn.Module,), {}))
    sys.path.insert(0, REF)
    from common.storage import Storage
    from common.model import ImpalaModel, MLPModel
    from common.policy import CategoricalPolicy
    from agents.ppo import PPO
    return Storage, ImpalaModel, MLPModel, CategoricalPolicy, PPO


Storage, ImpalaModel, MLPModel, CategoricalPolicy, PPO = _import_reference()
CPU = torch.device("cpu")


class _NullLogger:
    """PPO.optimize only reads .episode_reward_buffer (agents/ppo.py:97)."""
    episode_reward_buffer = [0.0]
    logdir = "/tmp"


def sd_numpy(module):
    return {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def flat_sha(module):
    flat = np.concatenate([p.detach().cpu().numpy().ravel() for p in module.parameters()]).astype(np.float32)
    return hashlib.sha256(flat.tobytes()).hexdigest()


def frames_to_ref_obs(frames_u8):
    """uint8 NHWC -> what the reference's wrapper chain hands the agent:
    TransposeFrame (NHWC->NCHW) then ScaledFloatFrame (/255.0, float64)
    (common/env/procgen_wrappers.py:391-419)."""
    return frames_u8.transpose(0, 3, 1, 2) / 255.0


# --------------------------------------------------------------------------- G1
def g1_gae():
    out = {}
    cases = [(8, 4, 0.0, False), (8, 4, 0.25, False), (64, 8, 0.02, False),
             (64, 8, 0.0, True), (256, 64, 0.02, False), (256, 64, 0.01, True)]
    meta = []
    for ci, (T, E, rate, last_done) in enumerate(cases):
        rng = np.random.default_rng(100 + ci)
        rew = rng.standard_normal((T, E)).astype(np.float32)
        done = (rng.random((T, E)) < rate).astype(np.float32)
        if last_done:
            done[T - 1, :] = 1.0
        val = rng.standard_normal((T + 1, E)).astype(np.float32)
        for norm in (False, True):
            st = Storage((3,), 4, T, E, CPU)
            st.rew_batch = torch.from_numpy(rew.copy())
            st.done_batch = torch.from_numpy(done.copy())
            st.value_batch = torch.from_numpy(val.copy())
            st.compute_estimates(0.999, 0.95, True, norm)
            tag = "norm" if norm else "raw"
            out[f"c{ci}_adv_{tag}"] = st.adv_batch.numpy().copy()
            out[f"c{ci}_ret"] = st.return_batch.numpy().copy()
        # the reference's use_gae=False branch (storage.py:68-77: result overwritten by adv(=0)+V)
        st = Storage((3,), 4, T, E, CPU)
        st.rew_batch = torch.from_numpy(rew.copy())
        st.done_batch = torch.from_numpy(done.copy())
        st.value_batch = torch.from_numpy(val.copy())
        st.compute_estimates(0.999, 0.95, False, False)
        out[f"c{ci}_ret_nogae"] = st.return_batch.numpy().copy()
        out[f"c{ci}_adv_nogae"] = st.adv_batch.numpy().copy()
        out[f"c{ci}_rew"], out[f"c{ci}_done"], out[f"c{ci}_val"] = rew, done, val
        meta.append(dict(T=T, E=E, gamma=0.999, lmbda=0.95))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "g1_gae.npz"), **out)


# --------------------------------------------------------------------------- G2
def g2_perm():
    """Index streams of Storage.fetch_train_generator (storage.py:81-110), captured by
    letting the reference generator call a collate_data that only records indices."""
    rec = {}
    for seed in (0, 123, 6033):
        for (T, E, B) in ((64, 8, 32), (256, 64, 2048), (256, 256, 8192), (256, 2048, 8192)):
            st = Storage((1,), 1, 1, 1, CPU)          # tiny tensors; only the sampler is exercised
            st.num_steps, st.num_envs = T, E
            chunks = []
            st.collate_data = lambda idx: chunks.append(np.asarray(idx, dtype=np.int64))
            torch.manual_seed(seed)
            for _ in range(2):                          # two epochs -> stream position matters
                for _ in st.fetch_train_generator(B, recurrent=False):
                    pass
            allidx = np.concatenate(chunks)
            key = f"s{seed}_T{T}_E{E}_B{B}"
            rec[key] = dict(n_chunks=len(chunks), first16=allidx[:16].tolist(), last16=allidx[-16:].tolist(),
                            crc_all=zlib.crc32(allidx.tobytes()),
                            crc_chunks=[zlib.crc32(c.tobytes()) for c in chunks])
    # recurrent: randperm(E) (storage.py:96)
    for seed in (0, 123):
        for E in (16, 1024):
            torch.manual_seed(seed)
            p = torch.randperm(E).numpy()
            rec[f"rec_s{seed}_E{E}"] = dict(first16=p[:16].tolist(), crc_all=zlib.crc32(p.astype(np.int64).tobytes()))
    with open(os.path.join(OUT, "g2_perm.json"), "w") as f:
        json.dump(rec, f, indent=0, sort_keys=True)


# --------------------------------------------------------------------------- helpers for G3..G6
def build_impala_policy(seed, A):
    torch.manual_seed(seed)
    model = ImpalaModel(in_channels=3)
    policy = CategoricalPolicy(model, False, A)
    policy.device = CPU
    return policy


def synth_rollout(rng, T, E, A, obs_kind):
    if obs_kind == "frames":
        frames = rng.integers(0, 256, size=(T + 1, E, 64, 64, 3), dtype=np.uint8)
    else:
        frames = rng.standard_normal((T + 1, E, obs_kind)).astype(np.float32)
    r = dict(
        frames=frames,
        act=rng.integers(0, A, size=(T, E)).astype(np.int64),
        rew=rng.standard_normal((T, E)).astype(np.float32),
        done=(rng.random((T, E)) < 0.05).astype(np.float32),
        logp=(np.log(1.0 / A) + 0.1 * rng.standard_normal((T, E))).astype(np.float32),
        val=(0.5 * rng.standard_normal((T + 1, E))).astype(np.float32),
    )
    return r


def fill_storage(st, r, T, E, is_frames):
    hid = np.zeros((E, st.hidden_state_size), dtype=np.float32)
    for t in range(T):
        obs = frames_to_ref_obs(r["frames"][t]) if is_frames else r["frames"][t]
        st.store(obs, hid, r["act"][t], r["rew"][t], r["done"][t], [{}] * E, r["logp"][t], r["val"][t])
    obs = frames_to_ref_obs(r["frames"][T]) if is_frames else r["frames"][T]
    st.store_last(obs, hid, r["val"][T])


def run_optimize(policy, st, T, E, hp, capture):
    agent = PPO(None, policy, _NullLogger(), st, CPU, 1, n_steps=T, n_envs=E, **hp)
    orig_step = agent.optimizer.step
    names = [k for k, _ in policy.named_parameters()]

    def step_hook(*a, **k):
        capture.setdefault("grads", []).append(
            {n: p.grad.detach().numpy().copy() for n, p in zip(names, policy.parameters()) if p.grad is not None})
        r = orig_step(*a, **k)
        capture.setdefault("params", []).append(sd_numpy(policy))
        return r

    agent.optimizer.step = step_hook
    summary = agent.optimize()
    return agent, summary


def tensor_stats(d):
    return {k: [float(np.sqrt((v.astype(np.float64) ** 2).sum())), float(v.astype(np.float64).sum())]
            for k, v in d.items()}


# --------------------------------------------------------------------------- G3 (+ params fixture)
def g3_forward():
    out = {}
    obs_u8 = np.random.default_rng(7).integers(0, 256, size=(8, 64, 64, 3), dtype=np.uint8)
    # a frame with flat colour regions: exercises max-pool tie-breaking
    obs_u8[1, :, :32] = 17
    obs_u8[1, :, 32:] = 200
    obs_u8[2] = 0
    out["obs_u8"] = obs_u8
    x = torch.FloatTensor(frames_to_ref_obs(obs_u8))
    shas = {}
    for A in (15, 9):
        policy = build_impala_policy(6033, A)
        shas[f"A{A}"] = flat_sha(policy)
        sd = sd_numpy(policy)
        if A == 15:
            for k, v in sd.items():
                out["p/" + k] = v
        else:
            for k in ("fc_policy.weight", "fc_policy.bias", "fc_value.weight", "fc_value.bias"):
                out["p9/" + k] = sd[k]
        m = policy.embedder
        with torch.no_grad():
            b1 = m.block1(x); b2 = m.block2(b1); b3 = m.block3(b2)
            c1 = m.block1.conv(x)
            p1 = torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)(c1)
            feat, _, fs, _ = m.forward_with_attn_indices(x)
            dist, value = policy.hidden_to_output(feat)
        if A == 15:
            out["act/block1_conv"] = c1.numpy()[:2]
            out["act/block1_pool"] = p1.numpy()[:2]
            out["act/block1"] = b1.numpy()[:2]
            out["act/block2"] = b2.numpy()[:2]
            out["act/block3"] = b3.numpy()
            out["act/feat"] = feat.numpy()
            out["act/fs"] = np.float32(fs.item())
        out[f"A{A}/logits"] = dist.logits.numpy()
        out[f"A{A}/value"] = value.numpy()
    out["sha"] = np.frombuffer(json.dumps(shas).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "g3_impala_forward.npz"), **out)


# --------------------------------------------------------------------------- G4/G5/G6
BASE_HP = dict(epoch=3, n_minibatch=8, mini_batch_size=8192, gamma=0.999, lmbda=0.95, learning_rate=5e-4,
               grad_clip_norm=0.5, eps_clip=0.2, value_coef=0.5, entropy_coef=0.01,
               normalize_adv=True, normalize_rew=True, use_gae=True)


def g4_loss_grad(arch):
    """One minibatch (B = T*E = 32): losses + raw grads.  Run twice through the reference's
    PPO.optimize: clip 1e9 (grads unclipped) and clip 0.5 (the clip coefficient)."""
    out = {}
    T, E = 4, 8
    A = 15 if arch == "impala" else 2
    rng = np.random.default_rng(11)
    r = synth_rollout(rng, T, E, A, "frames" if arch == "impala" else 9)
    for k, v in r.items():
        out["in/" + k] = v
    for tag, clip, xc in (("raw", 1e9, 0.0), ("clip", 0.5, 0.0), ("xent", 1e9, 0.05)):
        if arch == "impala":
            policy = build_impala_policy(6033, A)
            st = Storage((3, 64, 64), 256, T, E, CPU)
        else:
            policy = build_mlp_policy(6033, A)
            st = Storage((9,), 64, T, E, CPU)
        fill_storage(st, r, T, E, arch == "impala")
        st.compute_estimates(0.999, 0.95, True, True)
        hp = dict(BASE_HP, epoch=1, n_minibatch=1, mini_batch_size=T * E, grad_clip_norm=clip, x_entropy_coef=xc)
        cap = {}
        torch.manual_seed(5)
        agent, summary = run_optimize(policy, st, T, E, hp, cap)
        g = cap["grads"][0]
        out[f"{tag}/summary"] = np.frombuffer(json.dumps({k: float(v) for k, v in summary.items()}).encode(), np.uint8)
        out[f"{tag}/grad_stats"] = np.frombuffer(json.dumps(tensor_stats(g)).encode(), np.uint8)
        total = float(np.sqrt(sum((v.astype(np.float64) ** 2).sum() for v in g.values())))
        out[f"{tag}/grad_total_norm"] = np.float64(total)
        small = [k for k in g if g[k].size <= 4608]
        for k in small:
            out[f"{tag}/g/{k}"] = g[k]
        if arch != "impala":
            for k in g:
                out[f"{tag}/g/{k}"] = g[k]
        if tag == "raw":
            out["adv"] = st.adv_batch.numpy().copy()
            out["ret"] = st.return_batch.numpy().copy()
            if arch != "impala":
                for k, v in sd_numpy(build_mlp_policy(6033, A)).items():
                    out["p/" + k] = v
    np.savez_compressed(os.path.join(OUT, f"g4_{arch}_lossgrad.npz"), **out)


def g4_feature_sparsity():
    """G4 with the feature-sparsity term switched on (agents/ppo.py:148-169, common/model.py:207): fs_coef = 0.5.  DARK frames
    (pixel values 0..3): the network is bias-free at initialisation, so its activations scale with the input and tanh(100 h) stays
    out of saturation -- with ordinary frames almost every column's maximum sits at tanh = 1 and the term's gradient is exactly 0."""
    out = {}
    T, E, A = 4, 8, 15
    rng = np.random.default_rng(13)
    r = synth_rollout(rng, T, E, A, "frames")
    r["frames"] = (r["frames"] % 4).astype(np.uint8)
    for k, v in r.items():
        out["in/" + k] = v
    for tag, fs in (("fs0", 0.0), ("fs", 0.5)):
        policy = build_impala_policy(6033, A)
        st = Storage((3, 64, 64), 256, T, E, CPU)
        fill_storage(st, r, T, E, True)
        st.compute_estimates(0.999, 0.95, True, True)
        hp = dict(BASE_HP, epoch=1, n_minibatch=1, mini_batch_size=T * E, grad_clip_norm=1e9, fs_coef=fs)
        cap = {}
        torch.manual_seed(5)
        agent, summary = run_optimize(policy, st, T, E, hp, cap)
        g = cap["grads"][0]
        out[f"{tag}/summary"] = np.frombuffer(json.dumps({k: float(v) for k, v in summary.items()}).encode(), np.uint8)
        out[f"{tag}/grad_stats"] = np.frombuffer(json.dumps(tensor_stats(g)).encode(), np.uint8)
        out[f"{tag}/grad_total_norm"] = np.float64(np.sqrt(sum((v.astype(np.float64) ** 2).sum() for v in g.values())))
        for k in g:
            if g[k].size <= 9216:
                out[f"{tag}/g/{k}"] = g[k]
        if tag == "fs":
            out["adv"] = st.adv_batch.numpy().copy()
            out["ret"] = st.return_batch.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g4_impala_feature_sparsity.npz"), **out)


def g56_optimize(arch):
    """Full PPO.optimize on a stored (T=16,E=8) rollout: G5 (params after optimizer steps 1,2,8,
    accumulation case batch_size/B = 2) and G6 (3-epoch summary + final params)."""
    out = {}
    T, E = 16, 8
    A = 15 if arch == "impala" else 2
    rng = np.random.default_rng(13)
    r = synth_rollout(rng, T, E, A, "frames" if arch == "impala" else 9)
    for k, v in r.items():
        out["in/" + k] = v
    for tag, mbs in (("acc1", 16), ("acc2", 8)):
        if arch == "impala":
            policy = build_impala_policy(6033, A)
            st = Storage((3, 64, 64), 256, T, E, CPU)
        else:
            policy = build_mlp_policy(6033, A)
            st = Storage((9,), 64, T, E, CPU)
        fill_storage(st, r, T, E, arch == "impala")
        st.compute_estimates(0.999, 0.95, True, True)
        hp = dict(BASE_HP, epoch=3, n_minibatch=8, mini_batch_size=mbs)
        cap = {}
        torch.manual_seed(21)
        agent, summary = run_optimize(policy, st, T, E, hp, cap)
        out[f"{tag}/summary"] = np.frombuffer(json.dumps({k: float(v) for k, v in summary.items()}).encode(), np.uint8)
        out[f"{tag}/n_steps"] = np.int64(len(cap["params"]))
        for s in (1, 2, 8, len(cap["params"])):
            out[f"{tag}/param_stats_step{s}"] = np.frombuffer(
                json.dumps(tensor_stats(cap["params"][s - 1])).encode(), np.uint8)
        for k in ("fc_policy.weight", "fc_policy.bias", "fc_value.weight", "fc_value.bias"):
            for s in (1, 8, len(cap["params"])):
                out[f"{tag}/p_step{s}/{k}"] = cap["params"][s - 1][k]
        osd = agent.optimizer.state_dict()
        out[f"{tag}/adam_step"] = np.float64(float(osd["state"][0]["step"]))
        out[f"{tag}/adam_m_fc_value"] = osd["state"][len(osd["state"]) - 2]["exp_avg"].numpy().copy()
        out[f"{tag}/adam_v_fc_value"] = osd["state"][len(osd["state"]) - 2]["exp_avg_sq"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"g56_{arch}_optimize.npz"), **out)


# --------------------------------------------------------------------------- G7 (MLP / cartpole)
def build_mlp_policy(seed, A):
    torch.manual_seed(seed)
    model = MLPModel(9, 4, 256, 64)       # cartpole hparams: depth 4, mid_weight 256, latent 64 (config.yml:989-993)
    policy = CategoricalPolicy(model, False, A)
    policy.device = CPU
    return policy


def g7_mlp_forward():
    out = {}
    policy = build_mlp_policy(6033, 2)
    for k, v in sd_numpy(policy).items():
        out["p/" + k] = v
    x = np.random.default_rng(3).standard_normal((16, 9)).astype(np.float32)
    out["x"] = x
    with torch.no_grad():
        feat, _, fs, _ = policy.embedder.forward_with_attn_indices(torch.from_numpy(x))
        dist, value = policy.hidden_to_output(feat)
    out["feat"], out["logits"], out["value"] = feat.numpy(), dist.logits.numpy(), value.numpy()
    out["sha"] = np.frombuffer(flat_sha(policy).encode(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "g7_mlp_forward.npz"), **out)


# --------------------------------------------------------------------------- G8 (recurrent generator)
def g8_recurrent():
    T, E, B = 8, 16, 32            # N=128 -> 4 minibatches/epoch, 4 envs each
    rng = np.random.default_rng(17)
    st = Storage((5,), 6, T, E, CPU)
    obs = rng.standard_normal((T + 1, E, 5)).astype(np.float32)
    st.obs_batch = torch.from_numpy(obs.copy())
    st.hidden_states_batch = torch.from_numpy(rng.standard_normal((T + 1, E, 6)).astype(np.float32))
    st.act_batch = torch.from_numpy(rng.integers(0, 3, (T, E)).astype(np.float32))
    st.value_batch = torch.from_numpy(rng.standard_normal((T + 1, E)).astype(np.float32))
    torch.manual_seed(9)
    shapes, first = [], []
    for sample in st.fetch_train_generator(B, recurrent=True):
        shapes.append([list(s.shape) for s in sample])
        first.append(dict(obs00=sample[0][:, 0].numpy().tolist(), hid=sample[1][:, 0].numpy().tolist(),
                          act=sample[2].numpy().tolist(), val=sample[5].numpy().tolist()))
    # non-recurrent quirk: hidden_state_batch is the whole (N,H) tensor (storage.py:114-116)
    torch.manual_seed(9)
    s0 = next(iter(st.fetch_train_generator(B, recurrent=False)))
    np.savez_compressed(os.path.join(OUT, "g8_recurrent.npz"),
                        obs=obs, hid=st.hidden_states_batch.numpy(), act=st.act_batch.numpy(),
                        val=st.value_batch.numpy(),
                        meta=np.frombuffer(json.dumps(dict(T=T, E=E, B=B, seed=9, shapes=shapes, first=first,
                                                           nonrec_shapes=[list(s.shape) for s in s0])).encode(),
                                           np.uint8))


# --------------------------------------------------------------------------- G9 (recurrent predict path)
def g9_recurrent_predict():
    """CategoricalPolicy(recurrent=True).forward for three consecutive rollout steps (policy.py:61-72 ->
    GRU.forward prediction branch, model.py:219-225): hidden carried, masked by 1 - done."""
    torch.manual_seed(6033)
    model = ImpalaModel(in_channels=3)
    policy = CategoricalPolicy(model, True, 15)
    out = {}
    for k, v in sd_numpy(policy).items():
        if k.startswith("gru."):
            out["p/" + k] = v
    out["sha_all"] = np.frombuffer(flat_sha(policy).encode(), np.uint8)
    out["keys"] = np.frombuffer(json.dumps(list(policy.state_dict().keys())).encode(), np.uint8)
    rng = np.random.default_rng(23)
    E = 8
    frames = rng.integers(0, 256, size=(3, E, 64, 64, 3), dtype=np.uint8)
    done = np.stack([np.zeros(E), (rng.random(E) < 0.4).astype(np.float64), (rng.random(E) < 0.4).astype(np.float64)])
    hx = torch.zeros(E, 256)
    out["frames"], out["done"] = frames, done.astype(np.float32)
    with torch.no_grad():
        for t in range(3):
            x = torch.FloatTensor(frames_to_ref_obs(frames[t]))
            mask = torch.FloatTensor(1 - done[t])
            dist, value, hx = policy(x, hx, mask)
            out[f"logits{t}"], out[f"value{t}"], out[f"hx{t}"] = dist.logits.numpy(), value.numpy(), hx.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g9_recurrent_predict.npz"), **out)


def g10_logger():
    """Logger.feed / dump (common/logger.py:98-174) and Storage.fetch_log_data (common/storage.py:130-162) over three iterations of
    a (T=64, E=8) reward / done / info stream, training + validation: every CSV column of every dumped row (wall_time aside).
    Episodes run across iteration boundaries, one env never finishes, one episode is exactly max_steps long (timeout flag), and
    more than 40 episodes finish (the 40-deep deques roll over).  A second run feeds rewards / dones WITHOUT info dicts (the
    cartpole-style path: rew_batch / done_batch tensors are logged)."""
    import tempfile
    from common.logger import Logger
    T, E, K = 64, 8, 3
    rng = np.random.default_rng(31)
    out = {}
    for variant in ("info", "plain"):
        logdir = tempfile.mkdtemp()
        logger = Logger(E, logdir)
        logger.max_steps = 37
        st, stv = Storage((9,), 64, T, E, CPU), Storage((9,), 64, T, E, CPU)
        rows = []
        for k in range(K):
            streams = []
            for which, storage in (("t", st), ("v", stv)):
                rew = rng.standard_normal((T, E)).astype(np.float32)
                raw = (rng.integers(0, 3, size=(T, E)) * 5).astype(np.float32)           # env_reward: what Procgen reports (0 / 5 / 10)
                done = rng.random((T, E)) < (0.09 if which == "t" else 0.05)
                done[:, 3] = False                                                         # an env that never finishes
                if which == "t" and k == 0:
                    done[:, 5] = False; done[36, 5] = True                                 # first episode of env 5: exactly max_steps = 37 steps
                seeds = rng.integers(0, 6, size=(T, E))
                out[f"{variant}/{k}/{which}/rew"], out[f"{variant}/{k}/{which}/raw"] = rew, raw
                out[f"{variant}/{k}/{which}/done"], out[f"{variant}/{k}/{which}/seed"] = done, seeds
                z9, h = np.zeros((E, 9), np.float32), np.zeros((E, 64), np.float32)
                for t in range(T):
                    info = [{"env_reward": raw[t, e], "prev_level_seed": int(seeds[t, e])} for e in range(E)] if variant == "info" else [{} for _ in range(E)]
                    storage.store(z9, h, np.zeros(E), rew[t], done[t], info, np.zeros(E), np.zeros(E))
                storage.store_last(z9, h, np.zeros(E))
                if variant == "plain":
                    # no 'prev_level_seed': the reference's fetch_log_data averages an empty list (nan + warning)
                    pass
                streams.append(storage.fetch_log_data())
            (rb, db, tm), (rbv, dbv, tmv) = streams
            out[f"{variant}/{k}/fetch_t_rew"], out[f"{variant}/{k}/fetch_t_done"] = np.asarray(rb, np.float64), np.asarray(db, np.float64)
            logger.feed(rb, db, tm, rbv, dbv, tmv)
            summary = {'Loss/pi': 0.1 * k, 'Loss/v': -0.2, 'Loss/entropy': 2.7, 'Loss/x_entropy': 0.0, 'Loss/atn_entropy': float("nan"),
                       'Loss/atn_entropy2': float("nan"), 'Loss/sparsity': float("nan"), 'Loss/feature_sparsity': 0.8, 'Loss/total': 1.5 - k}
            logger.dump(summary, 5e-4 * (1 - k / K))
            rows.append([float(x) if x is not None else None for x in logger.log.loc[len(logger.log) - 1].tolist()])
        out[f"{variant}/rows"] = np.frombuffer(json.dumps(rows).encode(), np.uint8)
        out[f"{variant}/columns"] = np.frombuffer(json.dumps(list(logger.log.columns)).encode(), np.uint8)
        out[f"{variant}/csv"] = np.frombuffer(open(os.path.join(logdir, "log-append.csv")).read().encode(), np.uint8)
    np.savez_compressed(os.path.join(OUT, "g10_logger.npz"), **out)


def g11_checkpoint_structure():
    """What the reference WRITES as model_<t>.pth (agents/ppo.py:271-276) after real optimizer steps, as structure: top-level keys, the
    model state dict's keys / shapes / dtypes in order, the optimizer state dict's `state` (index -> step / exp_avg / exp_avg_sq with
    shapes and dtypes) and its `param_groups` list verbatim.  Two policies: IMPALA A=15 non-recurrent, and recurrent (the GRU's four
    tensors are in the model dict and in param_groups[0]['params'] but -- never receiving a gradient, SURVEY 8(a) A9 -- have NO
    optimizer state).  The file is written with torch.save and read back with weights_only=True, as train.py:257-263 would."""
    import io
    out = {}
    for tag, rec in (("impala", False), ("impala_rec", True)):
        torch.manual_seed(6033)
        policy = CategoricalPolicy(ImpalaModel(3), rec, 15)
        policy.device = CPU
        T, E = 4, 4
        st = Storage((3, 64, 64), 256, T, E, CPU)
        rng = np.random.default_rng(3)
        frames = rng.integers(0, 256, size=(T + 1, E, 64, 64, 3), dtype=np.uint8)
        hid = np.zeros((E, 256), np.float32)
        for t in range(T):
            st.store(frames_to_ref_obs(frames[t]), hid, rng.integers(0, 15, E), rng.standard_normal(E).astype(np.float32),
                     (rng.random(E) < 0.2), [{} for _ in range(E)], np.full(E, np.log(1 / 15), np.float32), rng.standard_normal(E).astype(np.float32))
        st.store_last(frames_to_ref_obs(frames[T]), hid, rng.standard_normal(E).astype(np.float32))
        st.compute_estimates(0.999, 0.95, True, True)
        agent = PPO(None, policy, _NullLogger(), st, CPU, 1, n_steps=T, n_envs=E, epoch=1, n_minibatch=2, mini_batch_size=8, learning_rate=5e-4)
        agent.optimize()                                                     # 2 optimizer steps
        buf = io.BytesIO()
        torch.save({'model_state_dict': agent.policy.state_dict(), 'optimizer_state_dict': agent.optimizer.state_dict()}, buf)
        buf.seek(0)
        ck = torch.load(buf, map_location="cpu", weights_only=True)
        desc = lambda v: [list(v.shape), str(v.dtype)]
        osd = ck["optimizer_state_dict"]
        out[tag] = {
            "top_keys": list(ck.keys()),
            "model": [[k, *desc(v)] for k, v in ck["model_state_dict"].items()],
            "opt_keys": list(osd.keys()),
            "opt_state": [[int(i), [[k, *desc(v)] for k, v in s.items()], float(s["step"])] for i, s in osd["state"].items()],
            "param_groups": osd["param_groups"],
            "n_parameters": len(list(agent.policy.parameters())),
        }
    json.dump(out, open(os.path.join(OUT, "g11_checkpoint_structure.json"), "w"), indent=1)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                      # regenerate selected fixtures only: python make_golden.py g4_feature_sparsity ...
        for name in sys.argv[1:]:
            globals()[name](); print(name, "done")
        sys.exit(0)
    g1_gae(); print("G1 done")
    g2_perm(); print("G2 done")
    g3_forward(); print("G3 done")
    g7_mlp_forward(); print("G7 fwd done")
    for arch in ("mlp", "impala"):
        g4_loss_grad(arch); print("G4", arch, "done")
        g56_optimize(arch); print("G5/6", arch, "done")
    g4_feature_sparsity(); print("G4 feature sparsity done")
    g10_logger(); print("G10 logger done")
    g8_recurrent(); print("G8 done")
    g9_recurrent_predict(); print("G9 done")
    g11_checkpoint_structure(); print("G11 done")
