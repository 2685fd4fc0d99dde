"""Host-side helpers of the PPO path (reference: common/misc_util.py:67-96).  The loss helpers of
that file (cross_batch_entropy etc.) live inside the fused HIP loss kernel (csrc/misc.hip)."""
import numpy as np
import torch
import torch.nn as nn


def set_global_seeds(seed):
    """misc_util.py:67-71 -- only torch is seeded by the reference (numpy / random are not)."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def set_global_log_levels(level):     # misc_util.py:74-75 touches gym's logger only; nothing to do here
    return level


def _init_linear_or_conv(module, fn, gain):
    if isinstance(module, (nn.Linear, nn.Conv2d)):
        fn(module.weight.data, gain)
        nn.init.constant_(module.bias.data, 0)
    return module


def orthogonal_init(module, gain=nn.init.calculate_gain('relu')):
    """misc_util.py:78-82 (a no-op on nn.GRU: the type check excludes it)."""
    return _init_linear_or_conv(module, nn.init.orthogonal_, gain)


def xavier_uniform_init(module, gain=1.0):
    """misc_util.py:85-89."""
    return _init_linear_or_conv(module, nn.init.xavier_uniform_, gain)


def adjust_lr(optimizer, init_lr, timesteps, max_timesteps):
    """Linear decay to zero (misc_util.py:92-96)."""
    lr = init_lr * (1 - (timesteps / max_timesteps))
    for group in optimizer.param_groups:
        group['lr'] = lr
    return optimizer, lr


def adjust_lr_grok(optimizer, init_lr, timesteps, max_timesteps):
    """misc_util.py:98-102."""
    lr = init_lr * (1.1 ** (timesteps / 1e6))
    for group in optimizer.param_groups:
        group['lr'] = lr
    return optimizer, lr


def get_n_params(model):
    return str(np.round(np.array([p.numel() for p in model.parameters()]).sum() / 1e6, 3)) + ' M params'
