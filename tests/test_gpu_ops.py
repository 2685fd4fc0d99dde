"""Kernel-level parity (MI355X): each HIP op through the C ABI against plain PyTorch-CPU fp32.
Tolerances: the MFMA path is an exact-fp32 fmaf chain; only the summation order differs from
torch's CPU kernels, so 1e-4 relative to the tensor's scale is generous."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(3, 16, 64), (16, 16, 32), (16, 32, 32), (32, 32, 16), (32, 32, 8)]


@pytest.fixture(scope="module")
def eng():
    from mi355.engine import Engine
    e = Engine("impala", n_steps=4, n_envs=4, n_actions=15, max_batch=16)
    yield e
    e.close()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def nchw(a):
    return torch.from_numpy(np.ascontiguousarray(a)).permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def test_mfma_operand_maps(eng):
    assert eng.selftest_mfma() == 0.0


@pytest.mark.parametrize("M,N,K,ta,tb", [(70, 50, 33, False, False), (128, 256, 2048, False, True),
                                         (16, 256, 4096, True, False), (256, 9, 300, True, False), (5, 3, 2, False, True)])
def test_gemm(eng, M, N, K, ta, tb):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    B = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    ref = (A.T if ta else A).astype(np.float64) @ (B.T if tb else B).astype(np.float64)
    out = eng.op_gemm(A, B, ta, tb)
    assert relerr(out, ref) < 1e-5


def _conv_inputs(cin, cout, hw, n, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    if cin == 3:
        x_u8 = torch.randint(0, 256, (n, hw, hw, 3), generator=g, dtype=torch.uint8).numpy()
        x = torch.from_numpy((x_u8.transpose(0, 3, 1, 2) / 255.0).astype(np.float32))
        return w, b, x_u8, x
    x = torch.randn(n, cin, hw, hw, generator=g)
    return w, b, nhwc(x), x


@pytest.mark.parametrize("cin,cout,hw", SHAPES)
@pytest.mark.parametrize("n", [1, 5])
def test_conv_forward(eng, cin, cout, hw, n):
    w, b, x_dev, x = _conv_inputs(cin, cout, hw, n, 1)
    relu = cin != 3
    res = torch.randn(n, cout, hw, hw, generator=torch.Generator().manual_seed(2))
    ref = F.conv2d(F.relu(x) if relu else x, w, b, padding=1) + res
    out = eng.op_conv3x3(0, cin, cout, hw, w.numpy(), inp=x_dev, relu_in=relu, bias=b.numpy(), res=nhwc(res))
    assert relerr(out, nhwc(ref)) < 1e-5
    out2 = eng.op_conv3x3(0, cin, cout, hw, w.numpy(), inp=x_dev, relu_in=False, bias=None)
    assert relerr(out2, nhwc(F.conv2d(x, w, None, padding=1))) < 1e-5


@pytest.mark.parametrize("cin,cout,hw", SHAPES[1:])
@pytest.mark.parametrize("n", [1, 5])
def test_conv_dgrad(eng, cin, cout, hw, n):
    w, _, _, x = _conv_inputs(cin, cout, hw, n, 3)
    g = torch.Generator().manual_seed(4)
    dout = torch.randn(n, cout, hw, hw, generator=g)
    skip = torch.randn(n, cin, hw, hw, generator=g)
    din = torch.nn.grad.conv2d_input(x.shape, w, dout, padding=1)
    ref = din * (x > 0) + skip
    out = eng.op_conv3x3(1, cin, cout, hw, w.numpy(), dout=nhwc(dout), mask=nhwc(x), res=nhwc(skip))
    assert relerr(out, nhwc(ref)) < 1e-5
    out2 = eng.op_conv3x3(1, cin, cout, hw, w.numpy(), dout=nhwc(dout))
    assert relerr(out2, nhwc(din)) < 1e-5


@pytest.mark.parametrize("cin,cout,hw", SHAPES)
@pytest.mark.parametrize("n", [1, 5, 37])
def test_conv_wgrad(eng, cin, cout, hw, n):
    w, _, x_dev, x = _conv_inputs(cin, cout, hw, n, 5)
    relu = cin != 3
    dout = torch.randn(n, cout, hw, hw, generator=torch.Generator().manual_seed(6))
    xin = F.relu(x) if relu else x
    ref_w = torch.nn.grad.conv2d_weight(xin, w.shape, dout, padding=1)
    ref_b = dout.sum(dim=(0, 2, 3))
    gw, gb = eng.op_conv3x3(2, cin, cout, hw, w.numpy(), inp=x_dev, relu_in=relu, dout=nhwc(dout))
    assert relerr(gw, ref_w.numpy()) < 2e-5
    assert relerr(gb, ref_b.numpy()) < 2e-5


@pytest.mark.parametrize("hw,c", [(64, 16), (32, 32), (16, 32)])
def test_maxpool_with_ties(eng, hw, c):
    g = torch.Generator().manual_seed(hw)
    x = torch.randn(3, c, hw, hw, generator=g)
    x[1] = torch.round(x[1])                 # many exact ties
    x[2, :, : hw // 2] = 0.25                # flat region (Procgen frames have these)
    x.requires_grad_(True)
    y = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    out = eng.op_maxpool(0, nhwc(x.detach()))
    assert np.array_equal(out, nhwc(y.detach()))
    dx = eng.op_maxpool(1, nhwc(x.detach()), dout=nhwc(dy))
    np.testing.assert_allclose(dx, nhwc(x.grad), rtol=0, atol=1e-6)
