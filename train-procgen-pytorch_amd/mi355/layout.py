"""Parameter naming / ordering of the two in-scope policies, as the reference's
``policy.state_dict()`` / ``policy.parameters()`` enumerate them:
common/model.py:167-179 (ImpalaModel), :954-971 (MLPModel), common/policy.py:39-40 (heads).
The flat fp32 vectors that cross the C ABI (mi_set_params, mi_get_grads, ...) use this order and
the reference's tensor shapes."""
from collections import OrderedDict

import numpy as np


def impala_param_shapes(n_actions, in_channels=3, output_dim=256):
    s = OrderedDict()
    chans = [in_channels, 16, 32, 32]
    for b in range(3):
        ci, co = chans[b], chans[b + 1]
        pre = f"embedder.block{b + 1}"
        s[f"{pre}.conv.weight"] = (co, ci, 3, 3)
        s[f"{pre}.conv.bias"] = (co,)
        for r in ("res1", "res2"):
            for c in ("conv1", "conv2"):
                s[f"{pre}.{r}.{c}.weight"] = (co, co, 3, 3)
                s[f"{pre}.{r}.{c}.bias"] = (co,)
    s["embedder.fc.weight"] = (output_dim, 32 * 8 * 8)
    s["embedder.fc.bias"] = (output_dim,)
    _heads(s, output_dim, n_actions)
    return s


def mlp_param_shapes(n_actions, obs_dim, depth, mid_weight, latent_size):
    s = OrderedDict()
    s["embedder.model.0.weight"] = (mid_weight, obs_dim)
    s["embedder.model.0.bias"] = (mid_weight,)
    for k in range(depth - 2):
        s[f"embedder.model.2.{2 * k}.weight"] = (mid_weight, mid_weight)
        s[f"embedder.model.2.{2 * k}.bias"] = (mid_weight,)
    s["embedder.model.3.weight"] = (latent_size, mid_weight)
    s["embedder.model.3.bias"] = (latent_size,)
    _heads(s, latent_size, n_actions)
    return s


def _heads(s, hidden, n_actions):
    s["fc_policy.weight"] = (n_actions, hidden)
    s["fc_policy.bias"] = (n_actions,)
    s["fc_value.weight"] = (1, hidden)
    s["fc_value.bias"] = (1,)


def flatten(shapes, tensors):
    """dict name -> array  ==> flat fp32 vector in parameters() order (shape-checked)."""
    parts = []
    for k, shp in shapes.items():
        a = np.asarray(tensors[k], dtype=np.float32)
        if tuple(a.shape) != tuple(shp):
            raise ValueError(f"{k}: shape {a.shape} != {shp}")
        parts.append(a.reshape(-1))
    return np.concatenate(parts)


def unflatten(shapes, flat):
    out, o = OrderedDict(), 0
    flat = np.asarray(flat)
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        out[k] = flat[o:o + n].reshape(shp).copy()
        o += n
    if o != flat.size:
        raise ValueError(f"flat vector has {flat.size} elements, layout needs {o}")
    return out
