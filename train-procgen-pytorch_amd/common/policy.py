"""CategoricalPolicy (reference: common/policy.py:18-87) over the MI355X engine.

Same constructor, attributes and state_dict keys; ``forward`` / ``hidden_to_output`` return
``torch.distributions.Categorical`` + value tensors on the host, computed by libmi355ppo.so.  The
nn.Parameters held here are the host VIEW of the weights: the working copy (and the Adam state)
lives in the engine's flat device buffers; ``state_dict()`` pulls, ``load_state_dict()`` pushes.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Categorical

from mi355 import layout
from .misc_util import orthogonal_init
from .model import GRU, as_device_obs


class CategoricalPolicy(nn.Module):
    def __init__(self, embedder, recurrent, action_size, has_vq=False, continuous_actions=False,
                 logsumexp_logits_is_v=False, extra_params=False):
        super().__init__()
        if has_vq or continuous_actions or logsumexp_logits_is_v or extra_params:
            raise NotImplementedError("only the discrete-action CategoricalPolicy of algo: ppo is accelerated")
        self.embedder = embedder
        self.has_vq = False
        self.continuous_actions = False
        self.action_size = action_size
        self.logsumexp_logits_is_v = False
        self.fc_policy = orthogonal_init(nn.Linear(embedder.output_dim, action_size), gain=0.01)
        self.fc_value = orthogonal_init(nn.Linear(embedder.output_dim, 1), gain=1.0)
        self.target_entropy = np.log(action_size)
        self.recurrent = recurrent
        if recurrent:
            self.gru = GRU(embedder.output_dim, embedder.output_dim)
        object.__setattr__(self, "engine", None)
        object.__setattr__(self, "_device_is_newer", False)
        self.device = None
        embedder._bind(self)

    # ---------------------------------------------------------------- engine plumbing
    @property
    def arch(self):
        return self.embedder.arch

    def param_shapes(self):
        if self.arch == "impala":
            return layout.impala_param_shapes(self.action_size)
        e = self.embedder
        return layout.mlp_param_shapes(self.action_size, e.input_size, e.depth, e.mid_weight, e.output_dim)

    def _host_tensors(self):
        sd = super().state_dict()
        return {k: v.detach().cpu().numpy() for k, v in sd.items() if not k.startswith("gru.")}

    def attach_engine(self, engine):
        object.__setattr__(self, "engine", engine)
        object.__setattr__(self, "_aux_engines", [])
        self.sync_to_device()
        self._upload_gru(engine)

    def attach_aux_engine(self, engine):
        """An inference-only twin (PPO's validation engine): it receives the frozen GRU now and again on load_state_dict."""
        self._aux_engines.append(engine)
        self._upload_gru(engine)

    def _upload_gru(self, engine):
        if self.recurrent:
            g = self.gru.gru
            engine.set_gru(*(t.detach().cpu().numpy() for t in (g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)))

    def sync_to_device(self):
        self.engine.set_params(layout.flatten(self.param_shapes(), self._host_tensors()))
        object.__setattr__(self, "_device_is_newer", False)

    def sync_from_device(self):
        if self.engine is None:
            return
        fresh = layout.unflatten(self.param_shapes(), self.engine.get_params())
        own = dict(self.named_parameters())
        with torch.no_grad():
            for k, v in fresh.items():
                own[k].copy_(torch.from_numpy(v))
        object.__setattr__(self, "_device_is_newer", False)

    def mark_device_updated(self):
        object.__setattr__(self, "_device_is_newer", True)

    def state_dict(self, *args, **kwargs):
        if self._device_is_newer:
            self.sync_from_device()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        if self.engine is not None:
            self.sync_to_device()
            # the GRU is not part of the flat parameter vector (never trained, A9): a checkpoint's GRU replaces the freshly
            # initialised one the engines were given at construction (train.py:257-263 loads AFTER building the agent)
            for e in [self.engine] + list(getattr(self, "_aux_engines", [])):
                self._upload_gru(e)
        return out

    # ---------------------------------------------------------------- reference API
    def is_recurrent(self):
        return self.recurrent

    def forward(self, x, hx, masks):
        if self.engine is None:
            self.embedder._policy()        # raises with the explanation
        if self.recurrent:
            to_np = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
            self.engine.rec_state(to_np(hx), 1.0 - to_np(masks))
            lp, value, hid = self.engine.forward_rec(as_device_obs(x, self.arch))
            return Categorical(logits=torch.from_numpy(lp)), torch.from_numpy(value), torch.from_numpy(hid)
        lp, value = self.engine.forward(as_device_obs(x, self.arch))
        return Categorical(logits=torch.from_numpy(lp)), torch.from_numpy(value), hx

    def hidden_to_output(self, hidden):
        raise NotImplementedError("hidden_to_output on host features is not exposed; use policy(obs, hx, masks)")

    def distribution(self, logits):
        return Categorical(logits=torch.log_softmax(torch.as_tensor(logits), dim=1))
