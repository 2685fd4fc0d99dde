"""train.py's CLI surface against the reference's (helper_local.py:562-664 flags, train.py:37-105 merge rule, train.py:273-300 log
directory / --model_file auto, hyperparams/procgen/config.yml sets).  CPU: argument / config / path logic only (the run itself is
tests/test_gpu_agent.py::test_train_cli_runs_and_resumes)."""
import argparse
import os
import re

import pytest
import yaml

from conftest import PKG


def _args(argv):
    import train
    return train.add_training_args(argparse.ArgumentParser()).parse_args(argv)


def test_every_reference_flag_parses_with_its_default():
    a = _args([])
    # defaults of helper_local.py:563-661 (device aside: there is no CPU fallback here)
    want = dict(exp_name='test', env_name='coinrun', val_env_name=None, start_level=0, num_levels=500, distribution_mode='easy',
                param_name='easy-200', gpu_device=0, num_timesteps=25000000, log_level=40, num_checkpoints=1, model_file=None,
                mut_info_alpha=None, gamma=None, lmbda=None, learning_rate=None, entropy_coef=None, n_envs=None, n_steps=None,
                n_minibatch=None, n_epochs=None, mini_batch_size=None, levels=None, sparsity_coef=0., output_dim=256, fs_coef=0.,
                random_percent=0, num_threads=8, detect_nan=False, use_valid_env=True, normalize_rew=True, render=False,
                paint_vel_info=True, reduce_duplicate_actions=True, use_wandb=False, real_procgen=True, mirror_env=False, use_gae=True,
                clip_value=True, anneal_temp=False, use_greedy_env=False, learned_gamma=False)
    for k, v in want.items():
        assert getattr(a, k) == v, k
    b = _args("--no-use_valid_env --no-normalize_rew --no-reduce_duplicate_actions --no-use_gae --detect_nan --gamma 0.9 --n_envs 32 "
              "--levels 1 2 3 --wandb_tags a b --model_file auto --mirror_env".split())
    assert (b.use_valid_env, b.normalize_rew, b.reduce_duplicate_actions, b.use_gae, b.detect_nan, b.mirror_env) == (False, False, False, False, True, True)
    assert b.gamma == 0.9 and b.n_envs == 32 and b.levels == [1, 2, 3] and b.model_file == "auto"
    # the reference's quirk: --no-learned_gamma / --no-use_greedy_env write to detect_nan (helper_local.py:633-634)
    assert _args(["--detect_nan", "--no-learned_gamma"]).detect_nan is False


def test_flag_names_cover_the_reference_parser():
    """Every `--flag` the reference's add_training_args declares exists here (names are listed, not read from /root/reference at test time)."""
    ref_flags = """exp_name env_name val_env_name start_level num_levels distribution_mode param_name device gpu_device num_timesteps seed
        log_level num_checkpoints model_file mut_info_alpha gamma lmbda learning_rate t_learning_rate dr_learning_rate entropy_coef n_envs
        n_steps n_minibatch n_epochs dyn_epochs val_epochs dr_epochs n_rollouts temperature done_coef rew_coef mini_batch_size wandb_name
        wandb_group wandb_tags minibatches levels sparsity_coef output_dim fs_coef random_percent key_penalty step_penalty rand_region
        num_threads detect_nan use_valid_env normalize_rew render paint_vel_info reduce_duplicate_actions use_wandb real_procgen mirror_env
        use_gae clip_value anneal_temp use_greedy_env learned_gamma no-learned_gamma no-use_greedy_env no-detect_nan no-use_valid_env
        no-normalize_rew no-render no-paint_vel_info no-reduce_duplicate_actions no-use_wandb no-real_procgen no-mirror_env no-use_gae
        no-clip_value no-anneal_temp""".split()
    import train
    p = train.add_training_args(argparse.ArgumentParser())
    have = {s.lstrip("-") for a in p._actions for s in a.option_strings}
    assert not [f for f in ref_flags if f not in have]


def test_merge_rule_cli_defaults_beat_the_file():
    import train
    hp = train.get_hyperparams("hard-500")
    hp["normalize_rew"] = False; hp["use_gae"] = False; hp["fs_coef"] = 0.3      # what a file could say
    out = train.merge_hyperparameters(dict(hp), _args([]))
    # argparse defaults that are not None win over the file (train.py:46-105): normalize_rew / use_gae / fs_coef / output_dim / sparsity_coef ...
    assert out["normalize_rew"] is True and out["use_gae"] is True and out["fs_coef"] == 0. and out["output_dim"] == 256
    assert out["n_envs"] == 256 and out["gamma"] == 0.999 and out["num_timesteps"] == 25000000        # None on the CLI: the file's value stays
    out = train.merge_hyperparameters(dict(hp), _args("--n_envs 64 --gamma 0.9 --lmbda 0.8 --n_minibatch 4 --mini_batch_size 512 --learning_rate 1e-3 "
                                                      "--entropy_coef 0.02 --no-use_gae --fs_coef 0.1 --precision bf16".split()))
    assert (out["n_envs"], out["gamma"], out["lmbda"], out["n_minibatch"], out["mini_batch_size"], out["learning_rate"], out["entropy_coef"]) == \
        (64, 0.9, 0.8, 4, 512, 1e-3, 0.02)
    assert out["use_gae"] is False and out["fs_coef"] == 0.1 and out["precision"] == "bf16"
    # --n_epochs does NOT set `epoch` (different key, as in the reference); mut_info_alpha splits the entropy coefficient (train.py:38-42)
    out = train.merge_hyperparameters(dict(hp), _args("--n_epochs 7 --mut_info_alpha 0.25".split()))
    assert out["epoch"] == 3 and out["n_epochs"] == 7
    assert out["entropy_coef"] == pytest.approx(0.01 * 0.25) and out["x_entropy_coef"] == pytest.approx(0.01 * 0.75)


def test_shipped_config_sets():
    import train
    sets = yaml.safe_load(open(os.path.join(PKG, "hyperparams", "procgen", "config.yml")))
    assert len(sets) == 18 and _args([]).param_name in sets            # the default --param_name loads
    for name, hp in sets.items():
        assert hp["algo"] == "ppo" and hp["architecture"] in ("impala", "mlpmodel"), name
    # the values BASELINE.json / SURVEY 8(d) quote
    h = sets["hard-500"]
    assert (h["n_envs"], h["n_steps"], h["epoch"], h["mini_batch_size"], h["gamma"], h["lmbda"], h["learning_rate"]) == \
        (256, 256, 3, 8192, 0.999, 0.95, 0.0005)
    # hard-500 says `mini_batch_per_epoch: 8`, a key PPO.__init__ ignores: n_minibatch keeps its default 8 (agents/ppo.py:22,39)
    assert h["mini_batch_per_epoch"] == 8 and "n_minibatch" not in h
    e = sets["easy"]
    assert (e["n_envs"], e["mini_batch_size"]) == (64, 2048) and sets["hard-rec"]["recurrent"] is True
    c = sets["cartpole"]
    assert (c["architecture"], c["depth"], c["mid_weight"], c["latent_size"]) == ("mlpmodel", 4, 256, 64)
    with pytest.raises(KeyError):
        train.get_hyperparams("hard-500-impalavq")                     # other architectures: not shipped, said so


def test_logdir_and_model_file_auto(tmp_path, monkeypatch):
    import train
    monkeypatch.chdir(tmp_path)
    logdir, mf = train.create_logdir_train(None, "coinrun", "exp", 7)
    assert re.fullmatch(r"logs/train/coinrun/exp/\d{4}-\d\d-\d\d__\d\d-\d\d-\d\d__seed_7", logdir) and os.path.isdir(logdir) and mf is None
    with pytest.raises(ValueError):
        train.create_logdir_train("auto", "coinrun", "exp", 7)         # no saved model yet
    for t in (4096, 20480, 8192):
        open(os.path.join(logdir, f"model_{t}.pth"), "wb").close()
    logdir2, mf = train.create_logdir_train("auto", "coinrun", "exp", 99)
    assert logdir2 == logdir and mf == os.path.join(logdir, "model_20480.pth")        # newest by step count, run directory reused
    other, _ = train.create_logdir_train(None, "coinrun", "exp", 8)
    open(os.path.join(other, "model_1.pth"), "wb").close()
    with pytest.raises(ValueError):
        train.create_logdir_train("auto", "coinrun", "exp", 7)         # two runs with models: ambiguous, as in the reference
    assert train.create_logdir_train("x/y.pth", "coinrun", "exp2", 1)[1] == "x/y.pth"
