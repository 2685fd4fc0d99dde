"""bf16-storage variant of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The engine's bf16 mode (mi_config.precision = 1, BASELINE config 3) is the fp32 algorithm of oracle/ppo_oracle.py with
roundings to bf16 at fixed points: what is STORED in HBM between kernels (activations, activation gradients) and what the
bf16 matrix cores take as operands (filter banks, fc.weight images) is bf16; every accumulation, bias, loss, parameter,
parameter gradient and the optimizer are fp32.  This module restates the IMPALA forward / backward pass with exactly those
rounding points (each one cites the kernel that rounds), so that the bf16 engine can be held to the oracle as tightly as
the fp32 engine is held to ppo_oracle.py -- instead of "the same direction as the fp32 gradient".

Parity status: PINNED through ppo_oracle.py: with the rounding switched off (`rounding=False`) every function here must
reproduce ppo_oracle.py's autograd results (tests/test_oracle_golden.py::test_bf16_oracle_without_rounding_is_the_fp32_oracle),
which in turn is pinned to the reference's golden vectors G3 / G4.  The rounding points themselves are the build's own
design (the reference has no bf16 path); they are checked kernel by kernel in tests/test_gpu_bf16.py.

Reference lines followed: common/model.py:134-209 (ImpalaModel / ImpalaBlock / ResidualBlock), common/policy.py:74-87,
agents/ppo.py:119-170 (loss + backward)."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import ppo_oracle as O


def _r16(t):
    """round to bf16 (nearest even), keep the tensor's dtype (fp32, or fp64 for the accumulation-order experiment below)"""
    return t.float().bfloat16().to(t.dtype)


def _ident(t):
    return t


def frames_to_bf16_obs(frames_u8):
    """uint8 NHWC -> NCHW float holding bf16(k/255): the uint8 -> bf16 table of block1.conv's staging (engine.hip lut16:
    (float)((double)k / 255.0) rounded to nearest even)."""
    return _r16(O.frames_to_obs(frames_u8))


def impala_forward(p, frames_u8, rounding=True, dtype=torch.float32):
    """Training-mode forward of the bf16 engine.  Returns (feat fp32 (B,256), cache).  Rounding points:
      * block{1,2,3}.conv: bf16 filter bank, fp32 accumulate + bias, output rounded to bf16 BEFORE the max pool (the fused
        conv+pool kernels pool bf16 keys: conv_bf16.hip conv1_pool_fwd_bf16_kernel, convpool_bf16.hip);
      * residual convs: bf16 banks; conv1 output a and block output y stored as bf16 (resblock_bf16.hip rb_pack);
      * embedder.fc: bf16 weight image, bf16 activations, fp32 accumulate + bias + ReLU, fp32 out (fc_bf16.hip)."""
    r = _r16 if rounding else _ident
    p = {k: v.to(dtype) for k, v in p.items()}
    x = r(O.frames_to_obs(frames_u8)).to(dtype)
    cache = {"x0": x, "blocks": []}
    for b in ("embedder.block1", "embedder.block2", "embedder.block3"):
        c = r(F.conv2d(x, r(p[b + ".conv.weight"]), p[b + ".conv.bias"], padding=1))
        q, idx = F.max_pool2d(c, kernel_size=3, stride=2, padding=1, return_indices=True)
        a1 = r(F.conv2d(F.relu(q), r(p[b + ".res1.conv1.weight"]), p[b + ".res1.conv1.bias"], padding=1))
        y1 = r(F.conv2d(F.relu(a1), r(p[b + ".res1.conv2.weight"]), p[b + ".res1.conv2.bias"], padding=1) + q)
        a2 = r(F.conv2d(F.relu(y1), r(p[b + ".res2.conv1.weight"]), p[b + ".res2.conv1.bias"], padding=1))
        y2 = r(F.conv2d(F.relu(a2), r(p[b + ".res2.conv2.weight"]), p[b + ".res2.conv2.bias"], padding=1) + y1)
        cache["blocks"].append(dict(name=b, xin=x, cshape=c.shape, idx=idx, q=q, a1=a1, y1=y1, a2=a2, y2=y2))
        x = y2
    flat = torch.flatten(F.relu(x), start_dim=1)
    feat = F.relu(F.linear(flat, r(p["embedder.fc.weight"]), p["embedder.fc.bias"]))
    cache["flat"], cache["x3"] = flat, x
    fs = torch.mean(torch.max(torch.tanh(torch.abs(flat * 100)), 0)[0])
    return feat, cache, fs


def impala_backward(p, cache, dfeat_post, feat, rounding=True, dtype=torch.float32, fs_coef=0.0):
    """Backward of the bf16 engine from d loss / d feat (after the fc ReLU).  Rounding points:
      * d feat (pre-ReLU, fp32) is rounded to bf16 only by the matrix-core fc kernels, which run for n >= 1024 (fc_bf16.hip
        fc_tn_kernel / fc_nt_kernel<true>); smaller batches take the fp32 GEMM with the fp32 fc.weight (engine.hip net_backward);
        fc.bias's gradient is the column sum of the unrounded d feat in both cases;
      * every activation gradient written to HBM or handed on through LDS is bf16: d flat, the gradient of a residual conv1's
        output (da), of a block's input (dx), of the pre-pool conv output rebuilt from the pooled gradient (dc; not block1's, see below);
      * data-gradient convs use the bf16 banks; weight gradients multiply bf16 operands exactly and accumulate in fp32."""
    r = _r16 if rounding else _ident
    p = {k: v.to(dtype) for k, v in p.items()}
    n = feat.shape[0]
    g = OrderedDict()
    dpre = (dfeat_post * (feat > 0)).to(dtype)
    big = rounding and n >= 1024
    dq = _r16(dpre) if big else dpre
    g["embedder.fc.weight"] = dq.t() @ cache["flat"]
    g["embedder.fc.bias"] = dpre.sum(0)
    wfc = _r16(p["embedder.fc.weight"]) if big else p["embedder.fc.weight"]
    gy = r((dq @ wfc).reshape(cache["x3"].shape) * (cache["x3"] > 0))
    if fs_coef:
        # + fs_coef * d mean_j max_b tanh(|100 h_bj|) / dh (common/model.py:207): one element per column, the row torch.max picks;
        # the engine adds it to the stored (bf16) gradient element and rounds again (misc.hip fs_apply_kernel)
        h = cache["flat"]
        t, am = torch.max(torch.tanh(torch.abs(h * 100)), 0)
        cols = torch.arange(h.shape[1])
        gf = gy.reshape(n, -1).clone()
        gf[am, cols] = r(gf[am, cols] + (fs_coef * 100.0 / h.shape[1]) * (1 - t * t).to(dtype) * torch.sign(h[am, cols]).to(dtype))
        gy = gf.reshape(gy.shape)
    ci = lambda shape, w, d: torch.nn.grad.conv2d_input(shape, w, d, padding=1)
    cw = lambda x, w, d: torch.nn.grad.conv2d_weight(x, w.shape, d, padding=1)
    for k in (2, 1, 0):
        B = cache["blocks"][k]
        b = B["name"]
        for res, xin, a in ((".res2", B["y1"], B["a2"]), (".res1", B["q"], B["a1"])):
            w1, w2 = p[b + res + ".conv1.weight"], p[b + res + ".conv2.weight"]
            da = r(ci(a.shape, r(w2), gy) * (a > 0))                       # resblock_bwd_full*: d(conv1 output), LDS only
            g[b + res + ".conv2.weight"] = cw(F.relu(a), w2, gy)
            g[b + res + ".conv2.bias"] = gy.sum(dim=(0, 2, 3))
            g[b + res + ".conv1.weight"] = cw(F.relu(xin), w1, da)
            g[b + res + ".conv1.bias"] = da.sum(dim=(0, 2, 3))
            gy = r(ci(xin.shape, r(w1), da) * (xin > 0) + gy)              # d(block input) incl. the skip connection
        # MaxPool2d(3,2,1) backward: every pooled gradient goes to its window's (first) maximum; the <= 4 contributions of a
        # conv-output pixel are summed in fp32 and the sum is rounded (PoolStage::gather in conv_bf16.hip)
        dc = torch.zeros(B["cshape"][0], B["cshape"][1], B["cshape"][2] * B["cshape"][3], dtype=dtype)
        dc.scatter_add_(2, B["idx"].reshape(dc.shape[0], dc.shape[1], -1), gy.reshape(dc.shape[0], dc.shape[1], -1))
        # block1.conv (k == 0) takes its weight gradient from the pooled gradients in one-hot form (conv_bf16.hip
        # conv1_wgrad_onehot_bf16_kernel): every g * x product is summed in fp32, the gathered sum is never rounded
        dc = dc.reshape(B["cshape"]) if k == 0 else r(dc.reshape(B["cshape"]))
        w = p[b + ".conv.weight"]
        g[b + ".conv.weight"] = cw(B["xin"], w, dc)
        g[b + ".conv.bias"] = dc.sum(dim=(0, 2, 3))
        if k > 0:
            gy = r(ci(B["xin"].shape, r(w), dc))
    return g


def cache_from_engine(frames_u8, acts, feat):
    """The forward tensors the ENGINE stored (mi_debug_read: acts[b] = dict(q, a1, y1, a2, y2 as NHWC arrays, arg = window position
    of each pooled element's maximum), feat (n,256)) in the layout impala_backward consumes: a teacher-forced backward pass sees
    exactly the engine's ReLU masks and max-pool routes, so what is compared is the backward arithmetic alone."""
    nchw = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32).transpose(0, 3, 1, 2)))
    x = frames_to_bf16_obs(frames_u8)
    cache = {"x0": x, "blocks": []}
    for b, A in zip(("embedder.block1", "embedder.block2", "embedder.block3"), acts):
        q = nchw(A["q"])
        n, c, ho, wo = q.shape
        arg = nchw(A["arg"]).long()
        oy = torch.arange(ho).reshape(1, 1, ho, 1); ox = torch.arange(wo).reshape(1, 1, 1, wo)
        iy, ix = 2 * oy - 1 + arg // 3, 2 * ox - 1 + arg % 3
        assert int(iy.min()) >= 0 and int(ix.min()) >= 0 and int(iy.max()) < 2 * ho and int(ix.max()) < 2 * wo
        cache["blocks"].append(dict(name=b, xin=x, cshape=(n, c, 2 * ho, 2 * wo), idx=iy * (2 * wo) + ix, q=q, a1=nchw(A["a1"]),
                                    y1=nchw(A["y1"]), a2=nchw(A["a2"]), y2=nchw(A["y2"])))
        x = cache["blocks"][-1]["y2"]
    cache["flat"], cache["x3"] = torch.flatten(F.relu(x), start_dim=1), x
    return torch.from_numpy(np.asarray(feat, np.float32)), cache


def loss_and_grads(params, frames_u8, act, old_logp, old_value, ret, adv, eps_clip=0.2, value_coef=0.5, entropy_coef=0.01,
                   x_entropy_coef=0.0, rounding=True, dtype=torch.float32, forward=None, fs_coef=0.0):
    """One minibatch of PPO.optimize (agents/ppo.py:119-170) with the bf16 engine's rounding points.
    -> (dict of loss terms incl. 'fs', OrderedDict of gradients keyed like the reference's state_dict).
    dtype=torch.float64 accumulates every contraction in fp64 (same rounding points): the distance between that run and the fp32
    one is the noise floor of the ALGORITHM under a change of summation order -- what two correct implementations may differ by.
    forward=(feat, cache) from cache_from_engine: teacher-forced backward on the engine's own forward tensors."""
    p = OrderedDict((k, torch.as_tensor(np.asarray(v, dtype=np.float32))) for k, v in params.items())
    with torch.no_grad():
        if forward is None:
            feat, cache, fs = impala_forward(p, frames_u8, rounding, dtype)
            feat = feat.float()
        else:
            feat, cache = forward
            fs = torch.mean(torch.max(torch.tanh(torch.abs(cache["flat"] * 100)), 0)[0])
    leaf = feat.clone().requires_grad_(True)
    hp = {k: p[k].clone().requires_grad_(True) for k in ("fc_policy.weight", "fc_policy.bias", "fc_value.weight", "fc_value.bias")}
    lp, value = O.heads(hp, leaf)
    L = O.ppo_loss(lp, value, act, old_logp, old_value, ret, adv, eps_clip, value_coef, entropy_coef, x_entropy_coef, 1.0, fs, fs_coef)
    L["total"].backward()
    with torch.no_grad():
        g = impala_backward(p, cache, leaf.grad, feat, rounding, dtype, fs_coef)
        g = OrderedDict((k, v.float()) for k, v in g.items())
    for k, t in hp.items():
        g[k] = t.grad.detach()
    out = {k: float(v.detach()) for k, v in L.items()}
    out["fs"] = float(fs)
    return out, OrderedDict((k, g[k]) for k in p)
