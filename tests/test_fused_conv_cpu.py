"""CPU: K7's oracle pinned against torch's own CPU weight gradient (the op the reference's CPU path runs,
URSABench/inference/sghmc.py:80), `fused_conv.Conv2d` on host tensors is the stock module, and the host-side plan of
the C ABI (which shapes K7 takes, how much scratch) - no launch without a GPU."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle_lib
from ursabench_amd import _native, fused_conv, models

SHAPES = [(3, 16, 16, 32, 1), (2, 32, 32, 16, 1), (2, 64, 64, 8, 1), (2, 16, 32, 32, 2), (1, 3, 16, 32, 1)]


@pytest.mark.parametrize('n,cin,cout,hw,stride', SHAPES)
def test_oracle_is_torchs_cpu_weight_gradient(n, cin, cout, hw, stride):
    rng = np.random.default_rng(n * 100 + cin)
    x = rng.standard_normal((n, cin, hw, hw), dtype=np.float32)
    dy = rng.standard_normal((n, cout, hw // stride, hw // stride), dtype=np.float32)
    dw = torch.from_numpy(oracle_lib.conv3x3_wgrad(x, dy, stride))
    shape = (cout, cin, 3, 3)
    ref64 = torch.nn.grad.conv2d_weight(torch.from_numpy(x).double(), shape, torch.from_numpy(dy).double(), stride, 1)
    assert torch.equal(dw, ref64.float())                 # the sum in double, rounded once: bit for bit torch's float64 op
    ref32 = torch.nn.grad.conv2d_weight(torch.from_numpy(x), shape, torch.from_numpy(dy), stride, 1)
    bound = 4e-6 * float(ref64.abs().max())               # any fp32 summation order over <= 6,144 products per element
    assert float((dw - ref32).abs().max()) <= bound


def test_conv2d_on_host_tensors_is_the_stock_module():
    torch.manual_seed(0)
    a = fused_conv.Conv2d(16, 16, 3, 1, 1, bias=False)
    b = nn.Conv2d(16, 16, 3, 1, 1, bias=False)
    b.load_state_dict(a.state_dict())                      # same keys
    assert isinstance(a, nn.Conv2d) and list(a.state_dict()) == ['weight']
    x = torch.randn(2, 16, 32, 32, requires_grad=True)
    xa, xb = x.detach().clone().requires_grad_(), x.detach().clone().requires_grad_()
    ya, yb = a(xa), b(xb)
    assert torch.equal(ya, yb)
    ya.square().sum().backward()
    yb.square().sum().backward()
    assert torch.equal(a.weight.grad, b.weight.grad) and torch.equal(xa.grad, xb.grad)
    c = copy.deepcopy(a)
    assert type(c) is fused_conv.Conv2d and torch.equal(c.weight, a.weight)


def test_models_keep_their_keys_and_parameter_count():
    m = models.PreResNet(10, 20)
    assert sum(p.numel() for p in m.parameters()) == 272282
    convs = [k for k, v in m.named_modules() if isinstance(v, nn.Conv2d)]
    assert len(convs) == 21 and all(isinstance(dict(m.named_modules())[k], fused_conv.Conv2d) for k in convs)


def test_enabled_switch():
    old = fused_conv.enabled()
    try:
        assert fused_conv.enabled(False) == old and fused_conv.enabled() is False
        assert fused_conv.enabled(True) is False and fused_conv.enabled() is True
    finally:
        fused_conv.enabled(old)


def test_plan_covers_the_three_resnet_stages_only():
    k = _native.default_kernels()
    for n in (1, 80, 128, 1000):
        for shape, cout, stride in (((n, 16, 32, 32), 16, 1), ((n, 32, 16, 16), 32, 1), ((n, 64, 8, 8), 64, 1)):
            f = k.conv3x3_wgrad_ws_floats(shape, cout, stride)
            assert f > 0 and f % (cout * shape[1] * 9) == 0          # whole partial copies of dW
    for shape, cout, stride in (((128, 3, 32, 32), 16, 1), ((128, 16, 32, 32), 32, 2), ((128, 16, 16, 16), 16, 1),
                                ((128, 160, 32, 32), 160, 1), ((0, 16, 32, 32), 16, 1), ((128, 16, 32, 16), 16, 1)):
        assert k.conv3x3_wgrad_ws_floats(shape, cout, stride) == 0


def test_argument_errors_do_not_need_a_gpu():
    import ctypes
    lib = _native.load_library()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    p -= p % 16
    f = lib.ursa_conv3x3_wgrad_f32
    assert f(None, p, p, p, 1 << 30, 128, 16, 16, 32, 32, 1, None) == -1          # ENULL
    assert f(p, p, p, p, 1 << 30, 0, 16, 16, 32, 32, 1, None) == -2               # ESIZE
    assert f(p + 4, p, p, p, 1 << 30, 128, 16, 16, 32, 32, 1, None) == -3         # EALIGN: x / dy / ws 16 bytes
    assert f(p, p, p + 2, p, 1 << 30, 128, 16, 16, 32, 32, 1, None) == -3         # dw 4 bytes
    assert f(p, p, p, p, 1 << 30, 128, 3, 16, 32, 32, 1, None) == -5              # EVALUE: shape not covered
    assert f(p, p, p, p, 1 << 30, 128, 16, 16, 32, 32, 2, None) == -5
    assert f(p, p, p, p, 100, 128, 16, 16, 32, 32, 1, None) == -2                 # scratch too small
