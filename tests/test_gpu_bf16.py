"""bf16-storage mode (mi_config.precision = 1; BASELINE config 3 dtype) on the MI355X.

Activations / activation gradients are bf16 in HBM, the 16/32-channel forward and data-gradient convs run on the
bf16 matrix cores with fp32 accumulation (weights rounded to bf16 when staged), weight gradients / block1.conv /
linear layers keep fp32 arithmetic on the bf16-stored operands.  Tolerances are bf16's: one rounding is 2^-9 = 0.2 %
relative, so kernel outputs are held to 1e-2 of the tensor's scale against an fp32 torch reference fed the SAME
bf16-rounded inputs (and bf16-rounded weights where the kernel rounds them), weight gradients (exact products,
fp32 accumulation) to 1e-4, and the end-to-end losses / gradients against the fp32 golden vectors to a few 1e-2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_npz, npz_json, npz_params

pytestmark = pytest.mark.gpu
SHAPES = [(3, 16, 64), (16, 16, 32), (16, 32, 32), (32, 32, 16), (32, 32, 8)]


def r16(t):
    return t.bfloat16().float()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.fixture(scope="module")
def eng():
    from mi355.engine import Engine
    e = Engine("impala", n_steps=4, n_envs=4, n_actions=15, max_batch=16, precision="bf16")
    yield e
    e.close()


def _inputs(cin, cout, hw, n, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    if cin == 3:
        x_u8 = torch.randint(0, 256, (n, hw, hw, 3), generator=g, dtype=torch.uint8).numpy()
        # block1.conv in bf16 mode stages the frame as bf16(k/255) (uint8 -> bf16 table)
        return w, b, x_u8, r16(torch.from_numpy((x_u8.transpose(0, 3, 1, 2) / 255.0).astype(np.float32)))
    x = r16(torch.randn(n, cin, hw, hw, generator=g))
    return w, b, nhwc(x), x


@pytest.mark.parametrize("cin,cout,hw", SHAPES)
@pytest.mark.parametrize("n", [1, 5])
def test_conv_forward_bf16(eng, cin, cout, hw, n):
    w, b, x_dev, x = _inputs(cin, cout, hw, n, 1)
    relu = cin != 3
    wq = r16(w)                                          # every conv of the bf16 mode rounds its filter bank to bf16
    res = r16(torch.randn(n, cout, hw, hw, generator=torch.Generator().manual_seed(2)))
    ref = F.conv2d(F.relu(x) if relu else x, wq, b, padding=1) + (res if relu else 0)       # block1.conv has no residual input
    out = eng.op_conv3x3(0, cin, cout, hw, w.numpy(), inp=x_dev, relu_in=relu, bias=b.numpy(), res=nhwc(res) if relu else None)
    assert relerr(out, nhwc(ref)) < 1e-2
    assert np.array_equal(out, r16(torch.from_numpy(out)).numpy())          # outputs are bf16 values


@pytest.mark.parametrize("cin,cout,hw", SHAPES[1:])
@pytest.mark.parametrize("n", [1, 5])
def test_conv_dgrad_bf16(eng, cin, cout, hw, n):
    w, _, _, x = _inputs(cin, cout, hw, n, 3)
    g = torch.Generator().manual_seed(4)
    dout = r16(torch.randn(n, cout, hw, hw, generator=g))
    skip = r16(torch.randn(n, cin, hw, hw, generator=g))
    din = torch.nn.grad.conv2d_input(x.shape, r16(w), dout, padding=1)
    ref = din * (x > 0) + skip
    out = eng.op_conv3x3(1, cin, cout, hw, w.numpy(), dout=nhwc(dout), mask=nhwc(x), res=nhwc(skip))
    assert relerr(out, nhwc(ref)) < 1e-2


@pytest.mark.parametrize("cin,cout,hw", SHAPES)
@pytest.mark.parametrize("n", [1, 37])
def test_conv_wgrad_bf16_inputs(eng, cin, cout, hw, n):
    w, _, x_dev, x = _inputs(cin, cout, hw, n, 5)
    relu = cin != 3
    dout = r16(torch.randn(n, cout, hw, hw, generator=torch.Generator().manual_seed(6)))
    xin = F.relu(x) if relu else x
    ref_w = torch.nn.grad.conv2d_weight(xin, w.shape, dout, padding=1)
    gw, gb = eng.op_conv3x3(2, cin, cout, hw, w.numpy(), inp=x_dev, relu_in=relu, dout=nhwc(dout))
    assert relerr(gw, ref_w.numpy()) < 1e-4
    assert relerr(gb, dout.sum(dim=(0, 2, 3)).numpy()) < 1e-4


@pytest.mark.parametrize("n", [1, 5, 37])
def test_block1_conv_pool_fused_bf16(eng, n):
    """block1.conv + MaxPool2d(3,2,1) in one launch (the conv output lives in LDS only), and the weight gradient taken
    straight from the POOLED gradient + arg-max bytes (pool backward fused into the operand staging).  The unfused
    kernels are the oracle for the forward (bit-identical: same conv arithmetic, same tie rule); torch autograd on the
    bf16-rounded conv output for the gradient."""
    w, b, x_u8, x = _inputs(3, 16, 64, n, 11)
    conv = eng.op_conv3x3(0, 3, 16, 64, w.numpy(), inp=x_u8, bias=b.numpy())              # unfused conv (bf16 values)
    pooled = eng.op_conv3x3(3, 3, 16, 64, w.numpy(), inp=x_u8, bias=b.numpy())
    assert np.array_equal(pooled, eng.op_maxpool(0, conv))
    c = torch.from_numpy(conv).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y = F.max_pool2d(c, kernel_size=3, stride=2, padding=1)
    assert np.array_equal(pooled, nhwc(y.detach()))
    dy = r16(torch.randn(y.shape, generator=torch.Generator().manual_seed(12)))
    y.backward(dy)
    dc = c.grad            # one-hot form (conv1_wgrad_onehot_bf16_kernel): the pooled gradients multiply the frame pixels directly, the <= 4
    ref_w = torch.nn.grad.conv2d_weight(x, w.shape, dc, padding=1)      # contributions of a conv pixel are never summed into a bf16 first
    gw, gb = eng.op_conv3x3(4, 3, 16, 64, w.numpy(), inp=x_u8, bias=b.numpy(), dout=nhwc(dy))
    assert relerr(gw, ref_w.numpy()) < 2e-5
    assert relerr(gb, dc.sum(dim=(0, 2, 3)).numpy()) < 2e-5


@pytest.mark.parametrize("cin,cout,hw", [(16, 32, 32), (32, 32, 16)])
@pytest.mark.parametrize("n", [1, 5, 12])
def test_block_conv_pool_fused_bf16(eng, cin, cout, hw, n):
    """block2.conv / block3.conv + MaxPool2d(3,2,1) in one launch, and both gradients of the conv taken from the POOLED
    gradient + arg-max bytes (max-pool backward fused into the operand staging of the weight- and data-gradient
    kernels).  Forward oracle: the unfused kernels (bit-identical: same conv arithmetic, same tie rule).  Backward
    oracle: torch autograd through max_pool2d on the kernel's own bf16 conv output."""
    w, b, x_dev, x = _inputs(cin, cout, hw, n, 21)
    conv = eng.op_conv3x3(0, cin, cout, hw, w.numpy(), inp=x_dev, bias=b.numpy())
    pooled = eng.op_conv3x3(3, cin, cout, hw, w.numpy(), inp=x_dev, bias=b.numpy())
    assert np.array_equal(pooled, eng.op_maxpool(0, conv))
    c = torch.from_numpy(conv).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y = F.max_pool2d(c, kernel_size=3, stride=2, padding=1)
    assert np.array_equal(pooled, nhwc(y.detach()))
    dy = r16(torch.randn(y.shape, generator=torch.Generator().manual_seed(22)))
    y.backward(dy)
    dc = r16(c.grad)                                                    # the staging rounds the gathered sum to bf16
    ref_w = torch.nn.grad.conv2d_weight(x, w.shape, dc, padding=1)
    gw, gb = eng.op_conv3x3(4, cin, cout, hw, w.numpy(), inp=x_dev, bias=b.numpy(), dout=nhwc(dy))
    assert relerr(gw, ref_w.numpy()) < 1e-4
    assert relerr(gb, dc.sum(dim=(0, 2, 3)).numpy()) < 1e-4
    ref_x = torch.nn.grad.conv2d_input(x.shape, r16(w), dc, padding=1)
    gx = eng.op_conv3x3(5, cin, cout, hw, w.numpy(), inp=x_dev, bias=b.numpy(), dout=nhwc(dy))
    assert relerr(gx, nhwc(ref_x)) < 1e-2
    # block2.conv and block3.conv: both gradients from ONE launch (one max-pool backward gather; block2_conv_bwd_bf16_kernel,
    # block3_conv_bwd_bf16_kernel); the data gradient is bit-identical to the generic kernel's
    gw2, gb2 = eng.op_conv3x3(6, cin, cout, hw, w.numpy(), inp=x_dev, bias=b.numpy(), dout=nhwc(dy))
    assert relerr(gw2, ref_w.numpy()) < 1e-4 and relerr(gb2, dc.sum(dim=(0, 2, 3)).numpy()) < 1e-4
    assert np.array_equal(eng.op_conv3x3(7, cin, cout, hw, w.numpy(), inp=x_dev, bias=b.numpy(), dout=nhwc(dy)), gx)


@pytest.mark.parametrize("cin,cout,hw", [(16, 32, 32), (32, 32, 16)])
def test_block_conv_pool_rolling_rows_equal_the_item_kernel(eng, cin, cout, hw):
    """Update-sized launches (n >= 1024) of conv + max pool walk down whole images with the last key row carried over in LDS
    (conv_pool_fwd_roll_bf16_kernel); smaller launches take 4-pooled-row items with one conv row recomputed.  Same arithmetic per conv
    output and pooling window: 1030 images in one launch == the same images in two launches of 515, bit for bit -- the pooled map, and
    (through the data gradient taken from a pooled gradient and the forward's arg-max bytes) the arg-max routes as well."""
    n = 1030
    w, b, x_dev, _ = _inputs(cin, cout, hw, n, 31)
    whole = eng.op_conv3x3(3, cin, cout, hw, w.numpy(), inp=x_dev, bias=b.numpy())
    halves = np.concatenate([eng.op_conv3x3(3, cin, cout, hw, w.numpy(), inp=x_dev[k:k + 515], bias=b.numpy()) for k in (0, 515)])
    assert np.array_equal(whole, halves) and np.abs(whole).max() > 0
    dy = nhwc(r16(torch.randn(n, cout, hw // 2, hw // 2, generator=torch.Generator().manual_seed(32))))
    gx = eng.op_conv3x3(5, cin, cout, hw, w.numpy(), inp=x_dev, bias=b.numpy(), dout=dy)
    gx2 = np.concatenate([eng.op_conv3x3(5, cin, cout, hw, w.numpy(), inp=x_dev[k:k + 515], bias=b.numpy(), dout=dy[k:k + 515]) for k in (0, 515)])
    assert np.array_equal(gx, gx2)


@pytest.mark.parametrize("ch,hw", [(16, 32), (32, 16), (32, 8)])
@pytest.mark.parametrize("n", [1, 6])
def test_fused_residual_block_bf16(eng, ch, hw, n):
    """ResidualBlock (common/model.py:141-146) forward and its two data gradients, one launch each.  The torch
    reference rounds where the kernels round: filters to bf16, conv1's output (forward) / its gradient (backward) to
    bf16 before the second conv consumes it."""
    g = torch.Generator().manual_seed(100 + ch + hw)
    w1, w2 = torch.randn(ch, ch, 3, 3, generator=g) * 0.15, torch.randn(ch, ch, 3, 3, generator=g) * 0.15
    b1, b2 = torch.randn(ch, generator=g), torch.randn(ch, generator=g)
    x = r16(torch.randn(n, ch, hw, hw, generator=g))
    a = r16(F.conv2d(F.relu(x), r16(w1), b1, padding=1))
    y = F.conv2d(F.relu(a), r16(w2), b2, padding=1) + x
    oa, oy = eng.op_resblock(0, nhwc(x), w1.numpy(), w2.numpy(), b1=b1.numpy(), b2=b2.numpy())
    assert relerr(oa, nhwc(a)) < 1e-2 and relerr(oy, nhwc(y)) < 1e-2
    dy = r16(torch.randn(n, ch, hw, hw, generator=g))
    da = r16(torch.nn.grad.conv2d_input(a.shape, r16(w2), dy, padding=1) * (a > 0))
    dx = torch.nn.grad.conv2d_input(x.shape, r16(w1), da, padding=1) * (x > 0) + dy
    ga, gx = eng.op_resblock(1, nhwc(dy), w1.numpy(), w2.numpy(), a_fwd=nhwc(a), x_fwd=nhwc(x))
    assert relerr(ga, nhwc(da)) < 1e-2 and relerr(gx, nhwc(dx)) < 1e-2


@pytest.mark.parametrize("ch,hw", [(16, 32), (32, 16), (32, 8)])
@pytest.mark.parametrize("n", [1, 3, 6])
def test_residual_pair_kernel_equals_two_single_launches(eng, ch, hw, n):
    """net_forward runs res1 + res2 of a block in ONE launch (res1's output reaches res2 through LDS and registers).  Same
    arithmetic as two launches of the single-block kernel, so with the same weights for both blocks: bit-identical."""
    g = torch.Generator().manual_seed(300 + ch + hw)
    w1, w2 = torch.randn(ch, ch, 3, 3, generator=g) * 0.1, torch.randn(ch, ch, 3, 3, generator=g) * 0.1
    b1, b2 = torch.randn(ch, generator=g), torch.randn(ch, generator=g)
    x = nhwc(r16(torch.randn(n, ch, hw, hw, generator=g)))
    _, y1 = eng.op_resblock(0, x, w1.numpy(), w2.numpy(), b1=b1.numpy(), b2=b2.numpy())
    a2, y2 = eng.op_resblock(0, y1, w1.numpy(), w2.numpy(), b1=b1.numpy(), b2=b2.numpy())
    pa, py = eng.op_resblock(3, x, w1.numpy(), w2.numpy(), b1=b1.numpy(), b2=b2.numpy())
    assert np.array_equal(pa, a2) and np.array_equal(py, y2)


def test_residual_pair_role_pipelined_kernel_equals_the_lds_bank_kernel(eng):
    """32 channels @16x16: launches of >= 1024 images run res1 and res2 as two wave roles pipelined over images, each role's filter banks
    in registers (resblock_pair32r_bf16_kernel); smaller launches keep the four banks in LDS (resblock_pair_bf16_kernel).  Same arithmetic
    per output element: 1030 images in one launch == the same images in two launches of 515, bit for bit (res2.conv1 output and block output)."""
    g = torch.Generator().manual_seed(811)
    w1, w2 = torch.randn(32, 32, 3, 3, generator=g) * 0.1, torch.randn(32, 32, 3, 3, generator=g) * 0.1
    b1, b2 = torch.randn(32, generator=g), torch.randn(32, generator=g)
    x = nhwc(r16(torch.randn(1030, 32, 16, 16, generator=g)))
    pa, py = eng.op_resblock(3, x, w1.numpy(), w2.numpy(), b1=b1.numpy(), b2=b2.numpy())
    halves = [eng.op_resblock(3, x[k:k + 515], w1.numpy(), w2.numpy(), b1=b1.numpy(), b2=b2.numpy()) for k in (0, 515)]
    assert np.array_equal(pa, np.concatenate([h[0] for h in halves])) and np.array_equal(py, np.concatenate([h[1] for h in halves]))


@pytest.mark.parametrize("ch,hw", [(16, 32), (32, 16), (32, 8)])
@pytest.mark.parametrize("n", [1, 7])
def test_residual_block_whole_backward_bf16(eng, ch, hw, n):
    """Residual block (16 channels @32x32, 32 channels @16x16): data gradients and both weight / bias gradients in ONE
    launch (the gradient of conv1's output only exists in LDS).  torch reference with the kernel's rounding points:
    filters bf16 for the data path, d(conv1 output) rounded to bf16 before conv1's transposed conv and weight gradient
    consume it."""
    g = torch.Generator().manual_seed(77)
    w1, w2 = torch.randn(ch, ch, 3, 3, generator=g) * 0.15, torch.randn(ch, ch, 3, 3, generator=g) * 0.15
    x = r16(torch.randn(n, ch, hw, hw, generator=g))
    a = r16(F.conv2d(F.relu(x), r16(w1), torch.randn(ch, generator=g), padding=1))
    dy = r16(torch.randn(n, ch, hw, hw, generator=g))
    da = r16(torch.nn.grad.conv2d_input(a.shape, r16(w2), dy, padding=1) * (a > 0))
    dx = torch.nn.grad.conv2d_input(x.shape, r16(w1), da, padding=1) * (x > 0) + dy
    dw2 = torch.nn.grad.conv2d_weight(F.relu(a), w2.shape, dy, padding=1)
    dw1 = torch.nn.grad.conv2d_weight(F.relu(x), w1.shape, da, padding=1)
    flat, gx = eng.op_resblock(2, nhwc(dy), w1.numpy(), w2.numpy(), a_fwd=nhwc(a), x_fwd=nhwc(x))
    flat = flat.ravel()
    wl = ch * ch * 9
    assert relerr(gx, nhwc(dx)) < 1e-2
    assert relerr(flat[:wl].reshape(ch, ch, 3, 3), dw1.numpy()) < 1e-4
    assert relerr(flat[wl:wl + ch], da.sum(dim=(0, 2, 3)).numpy()) < 1e-4
    assert relerr(flat[wl + ch:2 * wl + ch].reshape(ch, ch, 3, 3), dw2.numpy()) < 1e-4
    assert relerr(flat[2 * wl + ch:2 * wl + 2 * ch], dy.sum(dim=(0, 2, 3)).numpy()) < 1e-4


@pytest.mark.parametrize("hw,c", [(64, 16), (32, 32), (16, 32)])
def test_maxpool_bf16(eng, hw, c):
    g = torch.Generator().manual_seed(hw)
    x = r16(torch.randn(3, c, hw, hw, generator=g))
    x[1] = torch.round(x[1])
    x[2, :, : hw // 2] = 0.25
    x.requires_grad_(True)
    y = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    dy = r16(torch.randn(y.shape, generator=g))
    y.backward(dy)
    assert np.array_equal(eng.op_maxpool(0, nhwc(x.detach())), nhwc(y.detach()))
    dx = eng.op_maxpool(1, nhwc(x.detach()), dout=nhwc(dy))
    np.testing.assert_allclose(dx, nhwc(r16(x.grad)), rtol=1e-2, atol=1e-3)


def engine_forward_tensors(eng, n):
    """What the engine's last minibatch pass stored: per block the pooled map, the residual tensors and the arg-max; the features."""
    acts = [dict(q=eng.debug_read(8 * b + 0, n), a1=eng.debug_read(8 * b + 1, n), y1=eng.debug_read(8 * b + 2, n),
                 a2=eng.debug_read(8 * b + 3, n), y2=eng.debug_read(8 * b + 4, n), arg=eng.debug_read(8 * b + 5, n)) for b in range(3)]
    return acts, eng.debug_read(100, n)


def check_bf16_minibatch_against_oracle(eng, shapes, params, frames, idx, scal, hp_kw, tf_tol=5e-3):
    """One bf16 minibatch of the engine against oracle/ppo_oracle_bf16.py, three ways:
      (a) forward: every tensor the engine stored vs the oracle's, relative L2 <= 3e-3 (same rounding points; fp32 summation order
          differs, so a few values per thousand land on the neighbouring bf16 number -- one ulp = 0.4-0.8 %);
      (b) backward, teacher-forced: the oracle's backward pass run on the ENGINE's forward tensors (its ReLU masks, max-pool routes,
          features) -- compares the backward arithmetic alone: every gradient tensor <= tf_tol = 5e-3 relative L2 (measured 2.1e-3 at n = 1024, on a
          bias gradient: a sum of ~1 M bf16-rounded terms that cancels);
      (c) end to end: losses 1e-4 (or 3x the same floor); gradient tensors within 3x the ALGORITHM's own noise floor, measured here as the distance between
          the oracle accumulating in fp32 and in fp64 (same rounding points).  That floor is 3-6e-2 for every conv tensor on the G4
          inputs: a 1e-4 perturbation of the features flips ~1e-3 of the fc / conv ReLU masks, and the batch-summed gradients
          cancel to ~1/sqrt(B) of their terms, so single flips move them by per cents.  No two correct implementations of this
          bf16 algorithm agree better than that end to end; (b) is where the kernels' arithmetic is held tight."""
    from mi355 import layout
    from oracle import ppo_oracle_bf16 as OB
    n = len(idx)
    rec = eng.loss_log()[0]
    g = layout.unflatten(shapes, eng.get_grads())
    acts, feat = engine_forward_tensors(eng, n)
    fr = frames[idx]
    args = tuple(torch.from_numpy(np.asarray(a, np.float32).reshape(-1)[idx]) for a in scal)
    L, go = OB.loss_and_grads(params, fr, *args, **hp_kw)
    rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64).ravel() - np.asarray(b, np.float64).ravel()) / (np.linalg.norm(np.asarray(b, np.float64).ravel()) + 1e-12))
    # (a)
    with torch.no_grad():
        p = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()}
        f_o, cache, _ = OB.impala_forward(p, fr)
    for b in range(3):
        for k in ("q", "a1", "y1", "a2", "y2"):
            e = rel(acts[b][k], cache["blocks"][b][k].permute(0, 2, 3, 1).numpy())
            assert e < 3e-3, (b, k, e)
    assert rel(feat, f_o.numpy()) < 3e-3
    # (b)
    Ltf, gtf = OB.loss_and_grads(params, fr, *args, forward=OB.cache_from_engine(fr, acts, feat), **hp_kw)
    for j, k in enumerate(("pi_loss", "value_loss", "entropy", "x_ent", "total", "fs")):
        assert abs(rec[j] - Ltf[k]) < 2e-6 * max(1.0, abs(Ltf[k])), (k, rec[j], Ltf[k])      # same features in: fp32 loss arithmetic only
    worst_tf = max((rel(g[k], v.numpy()), k) for k, v in gtf.items())
    # (c)
    L64, g64 = OB.loss_and_grads(params, fr, *args, dtype=torch.float64, **hp_kw)
    for j, k in enumerate(("pi_loss", "value_loss", "entropy", "x_ent", "total", "fs")):
        # 1e-4, or 3x what the oracle itself moves by when it accumulates in fp64 (G4 inputs: value loss 2.31929 vs 2.31883)
        assert abs(rec[j] - L[k]) < max(1e-4 * max(1.0, abs(L[k])), 3.0 * abs(L[k] - L64[k])), (k, rec[j], L[k], L64[k])
    worst_e2e = (0.0, "", 0.0)
    for k, r in go.items():
        floor = rel(r.numpy(), g64[k].numpy())
        e = rel(g[k], r.numpy())
        worst_e2e = max(worst_e2e, (e, k, floor))
        assert e < 3.0 * floor + 1e-3, (k, e, floor)
    print(f"n={n}: teacher-forced backward worst {worst_tf}; end to end worst (error, tensor, algorithm noise floor) {worst_e2e}")
    assert worst_tf[0] < tf_tol, worst_tf


def test_end_to_end_bf16_against_golden_and_bf16_oracle():
    """Forward (G3) and one minibatch of losses + gradients (G4 inputs) in bf16 mode.  Outer reference: the reference's fp32
    numbers (what bf16 storage costs: forward 3e-2 of scale, losses 2e-3).  Parity proper: oracle/ppo_oracle_bf16.py, the fp32
    oracle with the kernels' rounding points (see check_bf16_minibatch_against_oracle)."""
    from mi355 import engine as M, layout
    from mi355.engine import Engine
    z = load_npz("g3_impala_forward.npz")
    shapes = layout.impala_param_shapes(15)
    params = npz_params(z)
    flat = layout.flatten(shapes, params)
    eng = Engine("impala", 2, 8, 15, 8, precision="bf16")
    eng.set_params(flat)
    lp, val, feat = eng.forward(z["obs_u8"], want_feat=True)
    assert relerr(feat, z["act/feat"]) < 3e-2
    np.testing.assert_allclose(lp, z["A15/logits"], rtol=0, atol=2e-3)      # logits are O(1e-2) at init (gain 0.01 head)
    np.testing.assert_allclose(val, z["A15/value"], rtol=0, atol=3e-2)
    eng.close()

    g4 = load_npz("g4_impala_lossgrad.npz")
    T, E = 4, 8
    frames = g4["in/frames"][:T].reshape(T * E, 64, 64, 3)
    for xc in (0.0, 0.05):
        eng = Engine("impala", T, E, 15, T * E, precision="bf16")
        eng.set_params(flat)
        for t in range(T + 1):
            eng.put_obs(t, g4["in/frames"][t])
        for t in range(T):
            eng.put_step(t, g4["in/rew"][t], g4["in/done"][t])
        eng.write_field(M.F_ACT, g4["in/act"].astype(np.float32)); eng.write_field(M.F_LOGP, g4["in/logp"]); eng.write_field(M.F_VALUE, g4["in/val"])
        eng.compute_estimates(0.999, 0.95, True, True)
        assert np.array_equal(eng.read_field(M.F_RET), g4["ret"])               # GAE path is fp32 in both modes: bit-exact
        idx = np.random.default_rng(0).permutation(T * E)
        eng.minibatch(idx, T * E, eng.hparams(x_entropy_coef=xc))
        rec = eng.loss_log(reset=False)[0]
        if xc == 0.0:
            ref = npz_json(g4, "raw/summary")
            assert abs(-rec[0] - ref["Loss/pi"]) < 2e-3
            assert abs(-rec[1] - ref["Loss/v"]) < 2e-2 * abs(ref["Loss/v"])
            assert abs(rec[2] - ref["Loss/entropy"]) < 1e-3
        scal = (g4["in/act"][:T], g4["in/logp"][:T], g4["in/val"][:T], g4["ret"], g4["adv"])
        check_bf16_minibatch_against_oracle(eng, shapes, params, frames, idx, scal, dict(x_entropy_coef=xc))
        eng.optimizer_step(5e-4, 0.5, 1)
        assert np.isfinite(eng.get_params()).all()
        eng.close()


def test_feature_sparsity_gradient_bf16_against_bf16_oracle():
    """fs_coef = 0.5 in bf16 mode on the dark-frame fixture inputs: forward tensors, teacher-forced backward and noise-floor bounded
    end-to-end gradients against oracle/ppo_oracle_bf16.py (which is pinned to the reference's fs numbers with its rounding off)."""
    from mi355 import engine as M, layout
    from mi355.engine import Engine
    z = load_npz("g4_impala_feature_sparsity.npz")
    T, E = 4, 8
    shapes = layout.impala_param_shapes(15)
    params = npz_params(load_npz("g3_impala_forward.npz"))
    eng = Engine("impala", T, E, 15, T * E, precision="bf16")
    eng.set_params(layout.flatten(shapes, params))
    for t in range(T + 1):
        eng.put_obs(t, z["in/frames"][t])
    for t in range(T):
        eng.put_step(t, z["in/rew"][t], z["in/done"][t])
    eng.write_field(M.F_ACT, z["in/act"].astype(np.float32)); eng.write_field(M.F_LOGP, z["in/logp"]); eng.write_field(M.F_VALUE, z["in/val"])
    eng.compute_estimates(0.999, 0.95, True, True)
    idx = np.random.default_rng(0).permutation(T * E)
    eng.minibatch(idx, T * E, eng.hparams(fs_coef=0.5))
    scal = (z["in/act"][:T], z["in/logp"][:T], z["in/val"][:T], z["ret"], z["adv"])
    check_bf16_minibatch_against_oracle(eng, shapes, params, z["in/frames"][:T].reshape(T * E, 64, 64, 3), idx, scal, dict(fs_coef=0.5))
    eng.close()


@pytest.mark.parametrize("n", [1, 7, 64, 256])
def test_fused_rollout_tail_equals_the_four_launches(n):
    """Rollout-sized inference passes (n <= 256, bf16) run IMPALA blocks 2 + 3 -- conv+pool, res pair, conv+pool, res pair -- as one
    launch with one workgroup per image (rollout_bf16.hip).  Same banks, same K order per output pixel, same rounding points, same
    first-maximum pooling: features, log-probs and values are BIT-identical to the four-launch path (mi_debug_flags bit 0), which
    in turn is what the update's training-mode forward runs.  Frames include flat patches (pooling ties) and saturated rows."""
    from mi355 import layout
    from mi355.engine import Engine
    z = load_npz("g3_impala_forward.npz")
    flat = layout.flatten(layout.impala_param_shapes(15), npz_params(z))
    rng = np.random.default_rng(n)
    frames = rng.integers(0, 256, size=(n, 64, 64, 3), dtype=np.uint8)
    frames[0, 8:40, 4:60] = 128
    if n > 2:
        frames[2] = 255; frames[1, :, :32] = 0
    eng = Engine("impala", 2, 8, 15, 256, precision="bf16")
    eng.set_params(flat * np.float32(1.7))                 # larger activations: more pooling / ReLU decisions in play
    fused = eng.forward(frames, want_feat=True)
    eng.debug_flags(1)
    plain = eng.forward(frames, want_feat=True)
    eng.debug_flags(0)
    again = eng.forward(frames, want_feat=True)
    eng.close()
    assert np.abs(plain[2]).max() > 1e-2
    for a, b, c in zip(fused, plain, again):
        assert np.array_equal(a, b) and np.array_equal(a, c)


@pytest.mark.parametrize("T,B", [(16, 1024), (18, 1152), (17, 1088)])
def test_fc_bf16_matrix_core_path_matches_small_batch_path(T, B):
    """Minibatches of >= 1024 samples route embedder.fc through the bf16-MFMA kernels of fc_bf16.hip (dedicated forward for n % 64 == 0,
    dedicated data gradient for n % 128 == 0, the tiled NT kernel otherwise, TN weight gradient); smaller ones through the fp32-MFMA
    GEMM on the same bf16-stored activations.  One B-sample minibatch must equal the same samples fed as two accumulated halves
    (n_global = B both times) up to the bf16 rounding of d(feat).  1152 = 18 / 9 row blocks: the XCD block map's partial last group;
    1088: dedicated forward + NT data gradient."""
    from mi355 import engine as M, layout
    from mi355.engine import Engine
    E, A = 64, 15
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, size=(T + 1, E, 64, 64, 3), dtype=np.uint8)
    shapes = layout.impala_param_shapes(A)
    flat = layout.flatten(shapes, npz_params(load_npz("g3_impala_forward.npz")))
    grads, recs = [], []
    for split in (False, True):
        eng = Engine("impala", T, E, A, B, precision="bf16")
        eng.set_params(flat)
        for t in range(T + 1):
            eng.put_obs(t, frames[t]); eng.sync()
        r2 = np.random.default_rng(4)
        eng.write_field(M.F_ACT, r2.integers(0, A, (T, E)).astype(np.float32))
        eng.write_field(M.F_LOGP, (np.log(1 / A) + 0.3 * r2.standard_normal((T, E))).astype(np.float32))
        eng.write_field(M.F_VALUE, (0.5 * r2.standard_normal((T + 1, E))).astype(np.float32))
        eng.write_field(M.F_REW, r2.standard_normal((T, E)).astype(np.float32))
        eng.write_field(M.F_DONE, (r2.random((T, E)) < 0.05).astype(np.float32))
        eng.compute_estimates(0.999, 0.95, True, True)
        idx = np.random.default_rng(5).permutation(T * E)
        if split:
            eng.minibatch(idx[:B // 2], B, eng.hparams()); eng.minibatch(idx[B // 2:], B, eng.hparams())
            log = eng.loss_log()
            recs.append(log[0] + log[1])                      # each record is that half's share of the B-sample means
        else:
            eng.minibatch(idx, B, eng.hparams())
            recs.append(eng.loss_log()[0])
        grads.append(layout.unflatten(shapes, eng.get_grads()))
        eng.close()
    for j in (0, 1, 2):
        assert abs(recs[0][j] - recs[1][j]) < 2e-4 * max(1.0, abs(recs[1][j])), (j, recs[0][j], recs[1][j])
    for k in shapes:
        a, b = grads[0][k].astype(np.float64).ravel(), grads[1][k].astype(np.float64).ravel()
        rel = np.linalg.norm(a - b) / np.linalg.norm(b)
        assert rel < 1e-1, (k, rel)          # deepest layer measured 3.5e-2: bf16 rounding of d(feat) + of every dgrad output below it


@pytest.mark.parametrize("deferred", [False, True])
def test_side_stream_minibatch_equals_the_single_stream_pass(deferred):
    """Update-sized bf16 minibatches without batch-level loss terms fork a side stream behind heads_bwd: the feature-sparsity metric,
    the loss records and embedder.fc's weight / bias gradients run beside the rest of the backward pass and are joined in front of the
    slab sums (engine.hip net_backward).  Same kernels on the same data in the same per-buffer order: loss records, gradients and the
    parameters after three optimizer steps (two of them with accumulated half-minibatches: two forks before one optimizer step) are
    BIT-identical to the single-stream pass (mi_debug_flags bit 4).  deferred: the multi-rank schedule without batch-level loss terms
    (mi_set_multirank mode 2: partial sums into the statistics ring, records derived once per optimize() by mi_loss_log_finalize) --
    what every rank of an 8-GPU run executes; here on one rank, where the cross-rank sum is the identity."""
    from mi355 import engine as M, layout
    from mi355.engine import Engine
    T, E, A, B = 32, 64, 15, 2048
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, size=(T + 1, E, 64, 64, 3), dtype=np.uint8)
    flat = layout.flatten(layout.impala_param_shapes(A), npz_params(load_npz("g3_impala_forward.npz")))
    out = []
    for flags in (0, 16):
        eng = Engine("impala", T, E, A, B, precision="bf16")
        eng.set_params(flat)
        eng.debug_flags(flags)
        if deferred:
            eng.set_multirank(2)
        for t in range(T + 1):
            eng.put_obs(t, frames[t]); eng.sync()
        r2 = np.random.default_rng(4)
        eng.write_field(M.F_ACT, r2.integers(0, A, (T, E)).astype(np.float32))
        eng.write_field(M.F_LOGP, (np.log(1 / A) + 0.3 * r2.standard_normal((T, E))).astype(np.float32))
        eng.write_field(M.F_VALUE, (0.5 * r2.standard_normal((T + 1, E))).astype(np.float32))
        eng.write_field(M.F_REW, r2.standard_normal((T, E)).astype(np.float32))
        eng.write_field(M.F_DONE, (r2.random((T, E)) < 0.05).astype(np.float32))
        eng.compute_estimates(0.999, 0.95, True, True)
        idx = np.random.default_rng(5).permutation(T * E)
        eng.minibatch(idx, B, eng.hparams())
        g1 = eng.get_grads().copy()
        eng.optimizer_step(5e-4, 0.5, 1)
        for k in range(2):
            eng.minibatch(idx[:B // 2], B, eng.hparams()); eng.minibatch(idx[B // 2:], B, eng.hparams())
            eng.optimizer_step(5e-4, 0.5, 1)
        if deferred:
            eng.loss_log_finalize()
        out.append((g1, np.array(eng.loss_log()), eng.get_params().copy()))
        eng.close()
    assert np.abs(out[0][0]).max() > 0 and len(out[0][1]) == 5
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


def test_fc_small_batch_kernel_matches_tiled_kernel():
    """embedder.fc has two bf16 kernels: the 128x64-tiled one for update-sized batches and the latency-oriented one for
    rollout-sized batches (n < 1024: one 16x16 output tile per workgroup, K split over the 4 waves).  Same packed
    weights, same bf16 products, fp32 accumulation in a different order -> equal to accumulation-order noise."""
    from mi355 import layout
    from mi355.engine import Engine
    z = load_npz("g3_impala_forward.npz")
    flat = layout.flatten(layout.impala_param_shapes(15), npz_params(z))
    eng = Engine("impala", 2, 8, 15, 1024, precision="bf16")
    eng.set_params(flat)
    frames = np.random.default_rng(3).integers(0, 256, size=(1024, 64, 64, 3), dtype=np.uint8)
    lp_big, v_big, f_big = eng.forward(frames, want_feat=True)
    lp_small, v_small, f_small = eng.forward(frames[:40], want_feat=True)          # 2 full groups of 16 envs + a ragged one
    eng.close()
    assert np.abs(f_big).max() > 1e-2
    np.testing.assert_allclose(f_small, f_big[:40], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(lp_small, lp_big[:40], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(v_small, v_big[:40], rtol=1e-4, atol=1e-5)
