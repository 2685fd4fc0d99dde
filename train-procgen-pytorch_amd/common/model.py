"""Embedders of the PPO path as parameter containers for the device engine.

The reference classes (common/model.py:134-209 ImpalaModel/ImpalaBlock/ResidualBlock, :954-980
MLPModel, :212-279 GRU) are torch modules whose forward runs cuDNN/cuBLAS kernels.  Here the same
module tree exists only to (a) draw the initial parameters with the reference's RNG consumption
(same constructors, same init calls, same order => bit-identical weights for a seed) and (b) give
``state_dict()`` the reference's key names and shapes for checkpoints.  All arithmetic happens in
libmi355ppo.so; ``forward`` routes through the owning policy's engine and there is no torch/CPU
compute path.
"""
import weakref

import numpy as np
import torch
import torch.nn as nn

from .misc_util import orthogonal_init, xavier_uniform_init


def as_device_obs(x, arch):
    """Whatever the caller holds -> what the engine stores.
    impala: uint8 NHWC frames.  Accepts uint8 (n,64,64,3) as is, or the reference's wrapper output
    (n,3,64,64) float in [0,1] (TransposeFrame + ScaledFloatFrame, procgen_wrappers.py:391-419),
    which is k/255 exactly and converts back losslessly.  mlp: fp32 rows."""
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    if arch != "impala":
        return np.ascontiguousarray(x, dtype=np.float32)
    if x.dtype == np.uint8:
        if x.shape[-1] != 3:
            x = x.transpose(0, 2, 3, 1)
        return np.ascontiguousarray(x)
    if x.ndim != 4 or x.shape[1] != 3:
        raise ValueError(f"expected (n,3,64,64) float frames or (n,64,64,3) uint8, got {x.shape} {x.dtype}")
    return np.ascontiguousarray(np.rint(x * 255.0).astype(np.uint8).transpose(0, 2, 3, 1))


class _EngineBacked(nn.Module):
    arch = None

    def _policy(self):
        ref = getattr(self, "_owner", None)
        pol = ref() if ref is not None else None
        if pol is None or pol.engine is None:
            raise NotImplementedError(
                f"{type(self).__name__} has no compute path of its own: wrap it in CategoricalPolicy and hand both to "
                "agents.ppo.PPO (which creates the MI355X engine), or call policy.attach_engine(...)")
        return pol

    def _bind(self, policy):
        object.__setattr__(self, "_owner", weakref.ref(policy))

    def forward(self, x):
        pol = self._policy()
        _, _, feat = pol.engine.forward(as_device_obs(x, self.arch), want_feat=True)
        return torch.from_numpy(feat)

    def forward_with_attn_indices(self, x):
        """(feat, atn_list, fs_loss, codes) like the reference (model.py:203-208 / :976-977); the
        feature-sparsity value itself is produced inside mi_minibatch (loss log column 'fs')."""
        return self.forward(x), [], None, None


class ResidualBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)
        self.conv2 = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)


class ImpalaBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.res1 = ResidualBlock(out_channels)
        self.res2 = ResidualBlock(out_channels)


class ImpalaModel(_EngineBacked):
    arch = "impala"

    def __init__(self, in_channels, output_dim=256, latent_dim=32, **kwargs):
        super().__init__()
        if in_channels != 3 or output_dim != 256 or latent_dim != 32:
            raise NotImplementedError("the HIP IMPALA-CNN is built for 3x64x64 frames, latent_dim 32, output_dim 256 "
                                      "(every procgen param set in hyperparams/procgen/config.yml)")
        self.block1 = ImpalaBlock(in_channels, 16)
        self.block2 = ImpalaBlock(16, 32)
        self.block3 = ImpalaBlock(32, latent_dim)
        self.encoded_dim = latent_dim * 8 * 8
        self.fc = nn.Linear(self.encoded_dim, output_dim)
        self.output_dim = output_dim
        self.apply(xavier_uniform_init)


class MLPModel(_EngineBacked):
    arch = "mlp"

    def __init__(self, in_channels, depth, mid_weight, latent_size, normalize=False):
        super().__init__()
        if normalize:
            raise NotImplementedError("MLPModel(normalize=True) (LayerNorm) is not on the accelerated path")
        self.input_size, self.depth, self.mid_weight, self.output_dim = in_channels, depth, mid_weight, latent_size
        mid = []
        for _ in range(depth - 2):
            mid += [nn.Linear(mid_weight, mid_weight), nn.ReLU()]
        self.model = nn.Sequential(nn.Linear(in_channels, mid_weight), nn.ReLU(), nn.Sequential(*mid),
                                   nn.Linear(mid_weight, latent_size))
        self.apply(xavier_uniform_init)


class GRU(nn.Module):
    """Weights of the recurrent wrapper (model.py:212-216; orthogonal_init is a no-op on nn.GRU, so torch's
    default uniform init stays).  In ``algo: ppo`` the reference runs the GRU in predict only and never trains it
    (SURVEY 8(a) A9): the cell runs inside the engine's policy step (csrc/misc.hip gru_gates_kernel)."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.gru = orthogonal_init(nn.GRU(input_size, hidden_size), gain=1.0)

    def forward(self, x, hxs, masks):
        raise NotImplementedError("the GRU cell runs inside the engine's policy step; call policy(obs, hx, masks)")
