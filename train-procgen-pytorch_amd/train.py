#!/usr/bin/env python3
"""CLI entry of the PPO path (reference: train.py:20-326, flags from helper_local.py:562-664 that matter to
algo: ppo).  `python train.py --exp_name x --env_name synthetic --param_name hard-500 --num_timesteps 1000000`."""
import argparse
import os
import random
import time

import numpy as np
import torch
import yaml

from agents.ppo import PPO
from common.env.vec_envs import CartPoleVec, SyntheticFrames, create_procgen_env
from common.logger import Logger
from common.misc_util import set_global_seeds
from common.model import ImpalaModel, MLPModel
from common.policy import CategoricalPolicy
from common.storage import Storage

HERE = os.path.dirname(os.path.abspath(__file__))


def get_hyperparams(param_name):
    with open(os.path.join(HERE, "hyperparams", "procgen", "config.yml")) as f:      # repo-relative (helper_local.py:207-210 used GLOBAL_DIR)
        return yaml.safe_load(f)[param_name]


def add_training_args(p):
    p.add_argument('--exp_name', type=str, default='test')
    p.add_argument('--env_name', type=str, default='coinrun')
    p.add_argument('--param_name', type=str, default='easy-200')
    p.add_argument('--device', type=str, default='gpu', choices=['gpu'])
    p.add_argument('--gpu_device', type=int, default=0)
    p.add_argument('--num_timesteps', type=int, default=int(25000000))
    p.add_argument('--seed', type=int, default=random.randint(0, 9999))
    p.add_argument('--num_checkpoints', type=int, default=1)
    p.add_argument('--model_file', type=str, default=None)
    p.add_argument('--n_envs', type=int, default=None)
    p.add_argument('--n_steps', type=int, default=None)
    p.add_argument('--n_minibatch', type=int, default=None)
    p.add_argument('--mini_batch_size', type=int, default=None)
    p.add_argument('--learning_rate', type=float, default=None)
    p.add_argument('--entropy_coef', type=float, default=None)
    p.add_argument('--x_entropy_coef', type=float, default=None)
    p.add_argument('--use_valid_env', action="store_true", default=True)
    p.add_argument('--no-use_valid_env', dest='use_valid_env', action="store_false")
    p.add_argument('--use_wandb', action="store_true")
    p.add_argument('--val_env_name', type=str, default=None)
    p.add_argument('--start_level', type=int, default=0)
    p.add_argument('--num_levels', type=int, default=500)
    p.add_argument('--distribution_mode', type=str, default='hard')
    p.add_argument('--num_threads', type=int, default=8)
    p.add_argument('--reduce_duplicate_actions', action="store_true", default=True)
    p.add_argument('--no-reduce_duplicate_actions', dest='reduce_duplicate_actions', action="store_false")
    # (new) IMPALA activation storage / matrix-core type: fp32 = the parity mode, bf16 = BASELINE config 3 (what bench.py measures)
    p.add_argument('--precision', type=str, default=None, choices=['fp32', 'bf16'])
    return p


def make_env(env_name, n_envs, seed, A, args=None, hp=None, is_valid=False):
    if env_name == "synthetic":
        return SyntheticFrames(n_envs, A, seed)
    if env_name.startswith("cartpole"):
        return CartPoleVec(n_envs, seed=seed)
    hp = hp or {}
    return create_procgen_env(env_name=env_name, n_envs=n_envs, is_valid=is_valid, val_env_name=getattr(args, "val_env_name", None),
                              start_level=getattr(args, "start_level", 0), num_levels=getattr(args, "num_levels", 500),
                              distribution_mode=getattr(args, "distribution_mode", "hard"), num_threads=getattr(args, "num_threads", 8),
                              paint_vel_info=hp.get("paint_vel_info", True), normalize_rew=hp.get("normalize_rew", True),
                              mirror_env=hp.get("mirror_env", False),
                              reduce_duplicate_actions=getattr(args, "reduce_duplicate_actions", True))


def initialize_model(device, env, hp):
    """helper_local.py:213-236,296-300,439-450 for architecture: impala | mlpmodel."""
    obs_shape = env.observation_space.shape
    arch = hp.get('architecture', 'impala')
    if arch == 'impala':
        model = ImpalaModel(in_channels=obs_shape[0], output_dim=hp.get("output_dim", 256), latent_dim=hp.get("latent_dim", 32))
    elif arch == 'mlpmodel':
        model = MLPModel(obs_shape[0], hp.get("depth", 4), hp.get("mid_weight", 64), hp.get("latent_size", 256))
    else:
        raise NotImplementedError(f"Architecture:{arch} is not on the accelerated path")
    policy = CategoricalPolicy(model, hp.get('recurrent', False), env.action_space.n)
    policy.device = device
    return model, obs_shape, policy


def train_ppo(args):
    hp = get_hyperparams(args.param_name)
    for k in ("n_envs", "n_steps", "n_minibatch", "mini_batch_size", "learning_rate", "entropy_coef", "x_entropy_coef", "precision"):
        if getattr(args, k, None) is not None:
            hp[k] = getattr(args, k)
    if hp.get("algo", "ppo") != "ppo":
        raise NotImplementedError("only algo: ppo is accelerated")
    # (new) one process per GPU under `python -m torch.distributed.run --nproc-per-node R train.py ...`: every rank owns
    # n_envs / R environments (its own env instances, seeded per rank) and the agent shards the update
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    if world > 1:
        args.gpu_device = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(args.gpu_device)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", args.gpu_device))
        if hp.get("n_envs", 256) % world:
            raise ValueError(f"n_envs={hp.get('n_envs', 256)} is not divisible by WORLD_SIZE={world}")
        hp["n_envs"] = hp.get("n_envs", 256) // world
        seed_t = torch.tensor([args.seed], dtype=torch.int64, device=torch.device("cuda", args.gpu_device))
        torch.distributed.broadcast(seed_t, 0)              # same initial weights and minibatch permutations on every rank
        args.seed = int(seed_t.item())
    set_global_seeds(args.seed)
    device = torch.device("cuda", args.gpu_device)
    n_envs, n_steps = hp.get("n_envs", 256), hp.get("n_steps", 256)
    A = 15 if hp.get("architecture", "impala") == "impala" else 2
    env = make_env(args.env_name, n_envs, args.seed + 2 * rank, A, args, hp)
    env_valid = make_env(args.env_name, n_envs, args.seed + 2 * rank + 1, A, args, hp, is_valid=True) if args.use_valid_env else None
    logdir = os.path.join('logs', 'train', args.env_name, args.exp_name, time.strftime("%Y-%m-%d__%H-%M-%S") + f'__seed_{args.seed}'
                          + (f'__rank_{rank}' if world > 1 else ''))
    os.makedirs(logdir, exist_ok=True)
    np.save(os.path.join(logdir, "hyperparameters.npy"), hp)
    model, obs_shape, policy = initialize_model(device, env, hp)
    logger = Logger(n_envs, logdir, use_wandb=args.use_wandb)
    logger.max_steps = hp.get("max_steps", 10 ** 3)
    storage = Storage(obs_shape, model.output_dim, n_steps, n_envs, device)
    storage_valid = Storage(obs_shape, model.output_dim, n_steps, n_envs, device) if args.use_valid_env else None
    agent = PPO(env, policy, logger, storage, device, args.num_checkpoints, env_valid=env_valid, storage_valid=storage_valid,
                seed=args.seed + rank, **hp)
    if args.model_file is not None:
        ck = torch.load(args.model_file, map_location="cpu", weights_only=True)
        agent.policy.load_state_dict(ck["model_state_dict"])
        agent.optimizer.load_state_dict(ck["optimizer_state_dict"])
        if ck.get("reward_norm") and getattr(env, "_rew", None) is not None:
            env._rew.load_state(ck["reward_norm"])
    agent.train(args.num_timesteps)
    return agent


if __name__ == '__main__':
    train_ppo(add_training_args(argparse.ArgumentParser()).parse_args())
