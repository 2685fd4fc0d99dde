#!/usr/bin/env python3
"""CLI entry of the PPO path (reference: train.py:20-326 `train_ppo`, flags helper_local.py:562-664).

    python train.py --exp_name x --env_name coinrun --param_name hard-500 --num_timesteps 200000000
    python -m torch.distributed.run --nproc-per-node 8 train.py ...          # (new) one rank per GPU, n_envs sharded

Kept from the reference: every flag of add_training_args with its default (flags that only other agents / architectures read are
accepted and ignored), the hyper-parameter merge rule of train.py:46-105 (a CLI value that is not None overrides config.yml -- so the
argparse defaults of normalize_rew / use_gae / output_dim / fs_coef / sparsity_coef / clip_value / anneal_temp beat the file, as
there), the log directory layout logs/train/<env>/<exp>/<time>__seed_<seed> with hyperparameters.npy + config.npy, the checkpoint
format, `--model_file auto` (create_logdir_train, train.py:277-294: the single run directory under the experiment that holds
model_*.pth is reused and its newest checkpoint loaded -- the reference finds that file and then still hands 'auto' to torch.load),
`--detect_nan`.  Different on purpose: `--device` defaults to gpu and `cpu` is refused (there is no CPU fallback); config.yml is
read relative to this file instead of the hard-coded GLOBAL_DIR of helper_local.py:28; new flags --precision, --rollout_groups."""
import os
import sys

if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:       # before anything initialises HIP: RCCL needs dmabuf IPC on this driver
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
# the pipelined rollout runs up to 4 env-group streams beside the main stream: with the runtime's default of 4 hardware queues two
# of them share a queue and serialise (measured: 4 groups 194 us per step with 4 queues, 134 us with 8)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8" if "--no-use_valid_env" in sys.argv else "16")
# (validation rollouts -- on by default -- run on a second engine with env-group streams of its own: with 8 queues two of ITS streams share
#  a queue with the training engine's and its rollout takes 45.7 instead of 29 ms, scratch/valid_rollout_time.py; without the second
#  engine 16 queues cost the update phase 0.6 %)

import argparse
import random
import time

import numpy as np
import torch
import yaml

from agents.ppo import PPO
from common.env.vec_envs import EnvGroups, SyntheticFrames, create_cartpole, create_procgen_env
from common.logger import Logger
from common.misc_util import set_global_seeds
from common.model import ImpalaModel, MLPModel
from common.policy import CategoricalPolicy
from common.storage import Storage

HERE = os.path.dirname(os.path.abspath(__file__))

# train.py:46-102: the names looked up first; afterwards EVERY key of the hyper-parameter set is looked up the same way
OVERRIDE_FIRST = ["n_envs", "n_steps", "n_minibatch", "mini_batch_size", "levels", "n_impala_blocks", "eps_clip", "increasing_lr",
                  "sparsity_coef", "normalize_rew", "gamma", "lmbda", "learning_rate", "t_learning_rate", "dr_learning_rate",
                  "entropy_coef", "fs_coef", "output_dim", "n_epochs", "n_rollouts", "temperature", "use_gae", "clip_value",
                  "done_coef", "dyn_epochs", "val_epochs", "dr_epochs", "rew_coef", "anneal_temp", "epoch", "value_coef", "t_coef",
                  "num_timesteps", "learned_gamma", "accumulate_all_grads", "learned_temp", "reward_incentive", "adv_incentive",
                  "alpha_learning_rate", "target_entropy_coef", "alpha", "n_imagined_actions", "n_transition_guesses", "beta",
                  "zv_loss_coef", "novelty_loss_coef", "separate_icm", "logsumexp_logits_is_v", "entropy_modified", "depth", "mid_weight"]


def get_hyperparams(param_name):
    with open(os.path.join(HERE, "hyperparams", "procgen", "config.yml")) as f:      # repo-relative (helper_local.py:207-210 used GLOBAL_DIR)
        sets = yaml.safe_load(f)
    if param_name not in sets:
        raise KeyError(f"--param_name {param_name}: not one of the shipped PPO sets {sorted(sets)} (other agents / architectures of the "
                       "reference's config.yml are out of scope)")
    return dict(sets[param_name])


def add_training_args(p):
    """helper_local.py:562-664, same names and defaults (see the module docstring for --device)."""
    p.add_argument('--exp_name', type=str, default='test')
    p.add_argument('--env_name', type=str, default='coinrun')
    p.add_argument('--val_env_name', type=str, default=None)
    p.add_argument('--start_level', type=int, default=0)
    p.add_argument('--num_levels', type=int, default=500)
    p.add_argument('--distribution_mode', type=str, default='easy')
    p.add_argument('--param_name', type=str, default='easy-200')
    p.add_argument('--device', type=str, default='gpu')
    p.add_argument('--gpu_device', type=int, default=0)
    p.add_argument('--num_timesteps', type=int, default=25000000)
    p.add_argument('--seed', type=int, default=random.randint(0, 9999))
    p.add_argument('--log_level', type=int, default=40)
    p.add_argument('--num_checkpoints', type=int, default=1)
    p.add_argument('--model_file', type=str)
    p.add_argument('--mut_info_alpha', type=float, default=None)
    for name, typ in (("gamma", float), ("lmbda", float), ("learning_rate", float), ("t_learning_rate", float), ("dr_learning_rate", float),
                      ("entropy_coef", float), ("n_envs", int), ("n_steps", int), ("n_minibatch", int), ("n_epochs", int), ("dyn_epochs", int),
                      ("val_epochs", int), ("dr_epochs", int), ("n_rollouts", int), ("temperature", float), ("done_coef", float),
                      ("rew_coef", float), ("mini_batch_size", int)):
        p.add_argument('--' + name, type=typ, default=None)
    p.add_argument('--wandb_name', type=str, default=None)
    p.add_argument('--wandb_group', type=str, default=None)
    p.add_argument('--wandb_tags', type=str, nargs='+')
    p.add_argument('--minibatches', type=int, nargs='+')
    p.add_argument('--levels', type=int, nargs='+', default=None)
    p.add_argument('--sparsity_coef', type=float, default=0.)
    p.add_argument('--output_dim', type=int, default=256)
    p.add_argument('--fs_coef', type=float, default=0.)
    p.add_argument('--random_percent', type=int, default=0)
    p.add_argument('--key_penalty', type=int, default=0)
    p.add_argument('--step_penalty', type=int, default=0)
    p.add_argument('--rand_region', type=int, default=0)
    p.add_argument('--num_threads', type=int, default=8)
    for flag in ("detect_nan", "use_valid_env", "normalize_rew", "render", "paint_vel_info", "reduce_duplicate_actions", "use_wandb",
                 "real_procgen", "mirror_env", "use_gae", "clip_value", "anneal_temp", "use_greedy_env", "learned_gamma"):
        p.add_argument('--' + flag, action="store_true")
    # helper_local.py:633-634 point --no-learned_gamma / --no-use_greedy_env at dest='detect_nan': kept (they switch detect_nan off)
    p.add_argument('--no-learned_gamma', dest='detect_nan', action="store_false")
    p.add_argument('--no-use_greedy_env', dest='detect_nan', action="store_false")
    for flag in ("detect_nan", "use_valid_env", "normalize_rew", "render", "paint_vel_info", "reduce_duplicate_actions", "use_wandb",
                 "real_procgen", "mirror_env", "use_gae", "clip_value", "anneal_temp"):
        p.add_argument('--no-' + flag, dest=flag, action="store_false")
    p.set_defaults(detect_nan=False, use_valid_env=True, normalize_rew=True, render=False, paint_vel_info=True,
                   reduce_duplicate_actions=True, use_wandb=False, real_procgen=True, mirror_env=False, use_gae=True, clip_value=True,
                   anneal_temp=False, use_greedy_env=False, learned_gamma=False)
    # ---- new flags
    p.add_argument('--precision', type=str, default=None, choices=['fp32', 'bf16'],
                   help="IMPALA activation storage / matrix-core type: fp32 = parity mode (default), bf16 = BASELINE config 3 (what bench.py measures)")
    p.add_argument('--rollout_groups', type=int, default=0,
                   help="env groups of the pipelined rollout (one group's frame upload + forward beside the host's env.step of another); 0 = auto (from 128 envs per rank: 4, or 2 + 2 when a validation env runs beside the training env; else 2); 1 = the reference's serial step")
    p.add_argument('--x_entropy_coef', type=float, default=None)
    return p


def merge_hyperparameters(hp, args):
    """train.py:37-105: mut_info_alpha split, the explicit override list, then every key already in the set."""
    a = vars(args)
    if a.get("mut_info_alpha") is not None:
        alpha, ent = a["mut_info_alpha"], hp["entropy_coef"]
        hp["entropy_coef"], hp["x_entropy_coef"] = ent * alpha, ent * (1 - alpha)
    for k in OVERRIDE_FIRST + list(hp.keys()):
        if a.get(k) is not None:
            hp[k] = a[k]
    for k in ("precision", "x_entropy_coef"):                       # (new flags follow the same rule)
        if a.get(k) is not None:
            hp[k] = a[k]
    return hp


def create_logdir_train(model_file, env_name, exp_name, seed, rank_suffix=""):
    """train.py:273-300.  -> (logdir, model_file): with 'auto' the one run directory that holds checkpoints and its newest model_<t>.pth."""
    logdir = os.path.join('logs', 'train', env_name, exp_name)
    if model_file == "auto":
        is_ckpt = lambda f: f.startswith("model_") and f.endswith(".pth") and f[6:-4].isdigit()
        runs = [os.path.join(logdir, d) for d in os.listdir(logdir)] if os.path.isdir(logdir) else []
        with_model = [d for d in runs if os.path.isdir(d) and any(is_ckpt(f) for f in os.listdir(d))]
        if len(with_model) > 1:
            raise ValueError(f"Received args.model_file = 'auto', but there are multiple experiments with saved models under experiment_name {exp_name}.")
        if len(with_model) == 0:
            raise ValueError(f"Received args.model_file = 'auto', but there are no saved models under experiment_name {exp_name}.")
        logdir = with_model[0]                                       # reuse logdir
        files = [f for f in os.listdir(logdir) if is_ckpt(f)]
        model_file = os.path.join(logdir, max(files, key=lambda f: int(f[6:-4])))
        if rank_suffix:                                               # only rank 0 writes checkpoints: the other ranks load rank 0's file
            logdir = logdir + rank_suffix                             # but log into a sibling directory of their own
    else:
        logdir = os.path.join(logdir, time.strftime("%Y-%m-%d__%H-%M-%S") + f'__seed_{seed}' + rank_suffix)
    os.makedirs(logdir, exist_ok=True)
    return logdir, model_file


def _one_env(env_name, n_envs, seed, A, args, hp, is_valid, ret_rms=None, num_threads=None):
    if env_name == "synthetic":
        return SyntheticFrames(n_envs, A, seed)
    if env_name.startswith("cartpole") or env_name == "mountain_car":
        if env_name == "mountain_car":
            raise NotImplementedError("mountain_car: only the cart-pole numpy env is built on the host side")
        return create_cartpole(hp, is_valid, seed=seed, n_envs=n_envs)        # 9 observations -> MLPModel(9, ...), BASELINE config 1
    return create_procgen_env(env_name=env_name, n_envs=n_envs, is_valid=is_valid, val_env_name=args.val_env_name,
                              start_level=args.start_level, num_levels=args.num_levels, distribution_mode=args.distribution_mode,
                              num_threads=num_threads or args.num_threads, paint_vel_info=hp.get("paint_vel_info", args.paint_vel_info),
                              normalize_rew=hp.get("normalize_rew", True), mirror_env=hp.get("mirror_env", args.mirror_env),
                              reduce_duplicate_actions=args.reduce_duplicate_actions, ret_rms=ret_rms)


def make_env(env_name, n_envs, seed, A, args, hp, is_valid=False):
    """get_env_constructor(env_name)(args, hp, is_valid) (common/env/env_constructor.py:13-31) for the envs of the PPO path.  With
    --rollout_groups G > 1 (and a non-recurrent policy) the n_envs environments are G independent sub-envs behind one VecEnv
    (EnvGroups): same protocol outwards, and the agent pipelines the groups.  Procgen groups share ONE running return variance, so
    reward normalisation stays a single statistic over all envs (procgen_wrappers.py:316-355) -- but it is UPDATED per group step:
    group g's rewards of step t are scaled by a variance that already holds groups < g of step t and not yet groups > g, where the
    reference's VecNormalize folds all n_envs returns in before scaling any.  The statistic converges to the same value; individual
    scaled rewards differ in the last digits early on.  `--rollout_groups 1` gives the reference's exact scaling (and its serial step)."""
    G = int(getattr(args, "rollout_groups", 1))
    if G > 4:
        raise ValueError(f"--rollout_groups {G}: the engine pipelines at most 4 env groups (mi_rollout_groups; more busy streams than that serialise)")
    if G <= 0:
        # auto: 4 chains keep the GPU's stream slots busy; with a validation env the agent runs both rollouts as lanes of one loop, 2 + 2
        G = (2 if getattr(args, "use_valid_env", False) else 4) if n_envs >= 128 and n_envs % 8 == 0 else 2
    if G == 1 or hp.get("recurrent", False) or n_envs % G or (n_envs // G) % 2 or env_name.startswith("cartpole"):
        return _one_env(env_name, n_envs, seed, A, args, hp, is_valid)
    rms = None
    if env_name != "synthetic" and hp.get("normalize_rew", True):
        from common.env.procgen_pipeline import RunningMoments
        rms = RunningMoments()
    return EnvGroups([_one_env(env_name, n_envs // G, seed + 7919 * g, A, args, hp, is_valid, rms, max(1, args.num_threads // G)) for g in range(G)])


def initialize_model(device, env, hp):
    """helper_local.py:213-236,296-300,439-450 for architecture: impala | mlpmodel."""
    obs_shape = env.observation_space.shape
    arch = hp.get('architecture', 'impala')
    if arch == 'impala':
        if hp.get("output_dim", 256) != 256:
            raise NotImplementedError("output_dim != 256: the IMPALA kernels are built for the reference's 2048 -> 256 embedder")
        model = ImpalaModel(in_channels=obs_shape[0], output_dim=256)
    elif arch == 'mlpmodel':
        model = MLPModel(obs_shape[0], hp.get("depth", 4), hp.get("mid_weight", 64), hp.get("latent_size", 256))
    else:
        raise NotImplementedError(f"Architecture:{arch} is not on the accelerated path")
    policy = CategoricalPolicy(model, hp.get('recurrent', False), env.action_space.n)
    policy.device = device
    return model, obs_shape, policy


def train_ppo(args):
    hp = merge_hyperparameters(get_hyperparams(args.param_name), args)
    env_name = hp.get("env_name", args.env_name)
    for key, value in hp.items():
        print(key, ':', value)
    if hp.get("algo", "ppo") != "ppo":
        raise NotImplementedError("only algo: ppo is accelerated")
    if hp.get("continuous", False):
        raise NotImplementedError("continuous actions are not part of the accelerated PPO path")
    if args.device != 'gpu':
        raise NotImplementedError(f"'device' must be 'gpu', not {args.device}: the MI355X path has no CPU fallback")
    # (new) one process per GPU under `python -m torch.distributed.run --nproc-per-node R train.py ...`: every rank owns
    # n_envs / R environments (its own env instances, seeded per rank) and the agent shards the update
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    if world > 1:
        args.gpu_device = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(args.gpu_device)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", args.gpu_device))
        if hp.get("n_envs", 256) % world:
            raise ValueError(f"n_envs={hp.get('n_envs', 256)} is not divisible by WORLD_SIZE={world}")
        hp["n_envs"] = hp.get("n_envs", 256) // world
        seed_t = torch.tensor([args.seed], dtype=torch.int64, device=torch.device("cuda", args.gpu_device))
        torch.distributed.broadcast(seed_t, 0)              # same initial weights and minibatch permutations on every rank
        args.seed = int(seed_t.item())
    set_global_seeds(args.seed)
    from mi355.numa import pin_to_gpu_node
    pin_to_gpu_node(int(getattr(args, "gpu_device", 0) or 0))      # (new) host threads + pinned buffers on the GPU's NUMA node
    device = torch.device("cuda", args.gpu_device)
    n_envs, n_steps = hp.get("n_envs", 256), hp.get("n_steps", 256)
    # the synthetic env's action count follows Procgen's: 15 combos, 9 after ActionWrapper merges duplicates (default, helper_local.py:653)
    A = (9 if args.reduce_duplicate_actions else 15) if hp.get("architecture", "impala") == "impala" else 2
    env = make_env(env_name, n_envs, args.seed + 2 * rank, A, args, hp)
    env_valid = make_env(env_name, n_envs, args.seed + 2 * rank + 1, A, args, hp, is_valid=True) if args.use_valid_env else None
    logdir, model_file = create_logdir_train(args.model_file, env_name, args.exp_name, args.seed, f'__rank_{rank}' if world > 1 and rank > 0 else '')
    np.save(os.path.join(logdir, "hyperparameters.npy"), hp)
    print(f'Logging to {logdir}')
    cfg = dict(vars(args)); cfg.update(hp)
    np.save(os.path.join(logdir, "config.npy"), cfg)
    model, obs_shape, policy = initialize_model(device, env, hp)
    logger = Logger(n_envs, logdir, use_wandb=args.use_wandb)
    logger.max_steps = hp.get("max_steps", 10 ** 3)
    storage = Storage(obs_shape, model.output_dim, n_steps, n_envs, device)
    storage_valid = Storage(obs_shape, model.output_dim, n_steps, n_envs, device) if args.use_valid_env else None
    agent = PPO(env, policy, logger, storage, device, args.num_checkpoints, env_valid=env_valid, storage_valid=storage_valid,
                seed=args.seed + rank, detect_nan=args.detect_nan, **hp)
    if model_file is not None:
        print("Loading agent from %s" % model_file)
        ck = torch.load(model_file, map_location="cpu", weights_only=True)
        agent.policy.load_state_dict(ck["model_state_dict"])
        agent.optimizer.load_state_dict(ck["optimizer_state_dict"])
        if "t" in ck:                                       # (new) what the reference's checkpoint forgets: continue the step count / LR schedule
            agent.t = int(ck["t"])
            agent.optimizer, _ = agent.adjust_lr(agent.optimizer, agent.learning_rate, agent.t, int(args.num_timesteps))
        rs = ck.get("reward_norm")
        for e in (getattr(env, "env_groups", None) or [env]):
            if rs and getattr(e, "_rew", None) is not None:
                e._rew.load_state(rs)
    print('START TRAINING...')
    agent.train(int(args.num_timesteps))
    return agent


if __name__ == '__main__':
    train_ppo(add_training_args(argparse.ArgumentParser()).parse_args())
