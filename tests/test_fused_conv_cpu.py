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

SHAPES = [(3, 16, 16, 32, 3, 1), (2, 32, 32, 16, 3, 1), (2, 64, 64, 8, 3, 1), (2, 16, 32, 32, 3, 2), (1, 3, 16, 32, 3, 1),
          (2, 16, 32, 32, 1, 2), (2, 32, 64, 16, 1, 2), (2, 32, 64, 16, 3, 2)]


@pytest.mark.parametrize('n,cin,cout,hw,ksize,stride', SHAPES)
def test_oracle_is_torchs_cpu_weight_gradient(n, cin, cout, hw, ksize, stride):
    rng = np.random.default_rng(n * 100 + cin)
    x = rng.standard_normal((n, cin, hw, hw), dtype=np.float32)
    dy = rng.standard_normal((n, cout, hw // stride, hw // stride), dtype=np.float32)
    dw = torch.from_numpy(oracle_lib.conv_wgrad(x, dy, ksize, stride))
    shape = (cout, cin, ksize, ksize)
    ref64 = torch.nn.grad.conv2d_weight(torch.from_numpy(x).double(), shape, torch.from_numpy(dy).double(), stride, ksize // 2)
    assert torch.equal(dw, ref64.float())                 # the sum in double, rounded once: bit for bit torch's float64 op
    ref32 = torch.nn.grad.conv2d_weight(torch.from_numpy(x), shape, torch.from_numpy(dy), stride, ksize // 2)
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


COVERED = [(3, 16, 32, 3, 1), (16, 16, 32, 3, 1), (32, 32, 16, 3, 1), (64, 64, 8, 3, 1), (16, 32, 32, 3, 2), (32, 64, 16, 3, 2),
           (16, 32, 32, 1, 2), (32, 64, 16, 1, 2)]          # (Cin, Cout, H = W, ksize, stride): every convolution of PreResNet-20


def test_plan_covers_the_basic_block_resnets_convolutions_only():
    k = _native.default_kernels()
    for n in (1, 80, 128, 1000):
        for cin, cout, hw, ks, st in COVERED:
            f = k.conv_wgrad_ws_floats((n, cin, hw, hw), cout, ks, st)
            assert f > 0 and f % ((cout // 16) * ((cin + 15) // 16) * ks * ks * 256) == 0      # whole partial copies, tile order
    for shape, cout, ks, st in (((128, 16, 16, 16), 16, 3, 1), ((128, 160, 32, 32), 160, 3, 1), ((0, 16, 32, 32), 16, 3, 1),
                                ((128, 16, 32, 16), 16, 3, 1), ((128, 16, 32, 32), 16, 1, 1), ((128, 16, 32, 32), 16, 5, 1),
                                ((128, 64, 8, 8), 64, 3, 2), ((128, 64, 16, 16), 16, 1, 1), ((128, 48, 32, 32), 64, 1, 1)):
        assert k.conv_wgrad_ws_floats(shape, cout, ks, st) == 0
    # K12 (round 6): the 1x1 / stride 1 layers of the Bottleneck networks, K slices bounded by the partial sums' size
    for n in (1, 128, 1024):
        for cin, cout, hw in ((64, 16, 32), (16, 64, 32), (128, 32, 16), (32, 128, 16), (256, 64, 8), (64, 256, 8), (64, 32, 32), (128, 64, 16)):
            f = k.conv_wgrad_ws_floats((n, cin, hw, hw), cout, 1, 1)
            assert f > 0 and f % ((cout // 16) * (cin // 16) * 256) == 0 and f * 4 <= max(16 << 20, 256 * (cout // 16) * (cin // 16) * 1024)
            assert k.conv1x1_supported((n, cin, hw, hw), cout) and k.conv1x1_supported((n, cout, hw, hw), cin, flip=True)
    assert not k.conv1x1_supported((4, 48, 32, 32), 64) and not k.conv1x1_supported((4, 64, 16, 16), 16)
    m = models.PreResNet(10, 20)
    x = torch.zeros(2, 3, 32, 32)
    shapes = []
    hooks = [mod.register_forward_hook(lambda mod, i, o: shapes.append((mod.in_channels, mod.out_channels, i[0].shape[2],
                                                                      mod.kernel_size[0], mod.stride[0])))
             for mod in m.modules() if isinstance(mod, nn.Conv2d)]
    m(x)
    for h in hooks:
        h.remove()
    assert len(shapes) == 21 and set(shapes) == set(COVERED)


def test_argument_errors_do_not_need_a_gpu():
    import ctypes
    lib = _native.load_library()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    p -= p % 16
    f, part, red = lib.ursa_conv_wgrad_f32, lib.ursa_conv_wgrad_partial_f32, lib.ursa_conv_wgrad_reduce_f32
    assert f(None, p, p, p, 1 << 30, 128, 16, 16, 32, 32, 3, 1, None) == -1          # ENULL
    assert f(p, p, None, p, 1 << 30, 128, 16, 16, 32, 32, 3, 1, None) == -1
    assert f(p, p, p, p, 1 << 30, 0, 16, 16, 32, 32, 3, 1, None) == -2               # ESIZE
    assert f(p + 4, p, p, p, 1 << 30, 128, 16, 16, 32, 32, 3, 1, None) == -3         # EALIGN: x / dy / ws 16 bytes
    assert f(p, p, p + 2, p, 1 << 30, 128, 16, 16, 32, 32, 3, 1, None) == -3         # dw 4 bytes
    assert f(p, p, p, p, 1 << 30, 128, 5, 16, 32, 32, 3, 1, None) == -5              # EVALUE: shape not covered
    assert f(p, p, p, p, 1 << 30, 128, 16, 16, 32, 32, 3, 2, None) == -5
    assert f(p, p, p, p, 100, 128, 16, 16, 32, 32, 3, 1, None) == -2                 # scratch too small
    assert part(p, p, None, 1 << 30, 128, 16, 16, 32, 32, 3, 1, None) == -1
    assert part(p, p, p, 100, 128, 16, 16, 32, 32, 3, 1, None) == -2
    assert red(None, 0, None) == 0 and red(None, 2, None) == -1 and red(None, -1, None) == -2
    items = (_native.ConvPending * 2)()
    for it in items:
        it.ws, it.dw, it.N, it.Cin, it.Cout, it.H, it.W, it.ksize, it.stride = p, p, 128, 16, 16, 32, 32, 3, 1
    items[1].Cin = 5                                                                  # the second item is not covered:
    assert red(ctypes.cast(items, ctypes.c_void_p), 2, None) == -5                    # nothing is launched for the first either
    items[1].Cin, items[1].dw = 16, None
    assert red(ctypes.cast(items, ctypes.c_void_p), 2, None) == -1


@pytest.mark.parametrize('n,cin,cout,hw,stride', [(2, 16, 16, 32, 1), (2, 3, 16, 32, 1), (1, 32, 32, 16, 1), (2, 64, 64, 8, 1), (1, 5, 7, 6, 1),
                                                 (2, 16, 32, 32, 2), (2, 32, 64, 16, 2), (1, 5, 7, 6, 2)])
def test_k8_oracle_is_torchs_cpu_convolution(n, cin, cout, hw, stride):
    rng = np.random.default_rng(n + cin)
    x = rng.standard_normal((n, cin, hw, hw), dtype=np.float32)
    w = rng.standard_normal((cout, cin, 3, 3), dtype=np.float32)
    dy = rng.standard_normal((n, cout, hw // stride, hw // stride), dtype=np.float32)
    tx, tw, tdy = torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(dy)
    y = torch.from_numpy(oracle_lib.conv3x3(x, w, stride=stride))
    assert torch.equal(y, torch.nn.functional.conv2d(tx.double(), tw.double(), None, stride, 1).float())
    y32 = torch.nn.functional.conv2d(tx, tw, None, stride, 1)
    assert float((y - y32).abs().max()) <= 4e-6 * float(y.abs().max())
    dx = torch.from_numpy(oracle_lib.conv3x3(dy, w, flip=True, stride=stride))
    assert torch.equal(dx, torch.nn.grad.conv2d_input(x.shape, tw.double(), tdy.double(), stride, 1).float())


def test_k8_plan_and_argument_errors():
    import ctypes
    k = _native.default_kernels()
    for shape, cout in (((128, 16, 32, 32), 16), ((1, 3, 32, 32), 16), ((4096, 32, 16, 16), 32), ((7, 64, 8, 8), 64)):
        assert k.conv3x3_supported(shape, cout)
    for shape, cout in (((128, 16, 32, 32), 3), ((128, 16, 16, 16), 16), ((128, 16, 32, 32), 32), ((0, 16, 32, 32), 16),
                        ((128, 160, 32, 32), 160), ((2, 16, 32, 16), 16)):
        assert not k.conv3x3_supported(shape, cout)
    assert k.conv3x3_supported((128, 16, 32, 32), 32, stride=2) and k.conv3x3_supported((5, 32, 16, 16), 64, stride=2)
    assert k.conv3x3_supported((128, 32, 16, 16), 16, flip=True, stride=2) and k.conv3x3_supported((5, 64, 8, 8), 32, flip=True, stride=2)
    assert k.conv3x3_supported((128, 16, 32, 32), 16, flip=True) and not k.conv3x3_supported((128, 16, 32, 32), 3, flip=True)
    assert not k.conv3x3_supported((128, 16, 32, 32), 16, stride=2) and not k.conv3x3_supported((128, 16, 32, 32), 32, flip=True, stride=2)
    lib = _native.load_library()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    p -= p % 16
    f = lib.ursa_conv3x3_f32
    assert f(p, p, p, 128, 16, 16, 32, 32, 0x4, None) == -4            # EFLAGS
    assert f(None, p, p, 128, 16, 16, 32, 32, 0, None) == -1           # ENULL
    assert f(p, p, p, 0, 16, 16, 32, 32, 0, None) == -2                # ESIZE
    assert f(p + 4, p, p, 128, 16, 16, 32, 32, 0, None) == -3          # EALIGN
    assert f(p, p + 2, p, 128, 16, 16, 32, 32, 0, None) == -3
    assert f(p, p, p, 128, 16, 3, 32, 32, 1, None) == -5               # EVALUE: the stem's input gradient is not covered
    assert f(p, p, p, 128, 16, 32, 32, 32, 0, None) == -5
    assert f(p, p, p, 128, 3, 16, 32, 32, 1, None) == -5               # flipped form: equal-width layers only


@pytest.mark.parametrize('n,cin,cout,hw', [(2, 16, 32, 32), (2, 32, 64, 16), (1, 5, 7, 6)])
def test_k9_oracle_is_torchs_cpu_convolution(n, cin, cout, hw):
    rng = np.random.default_rng(n + cin)
    x = rng.standard_normal((n, cin, hw, hw), dtype=np.float32)
    w = rng.standard_normal((cout, cin, 1, 1), dtype=np.float32)
    dy = rng.standard_normal((n, cout, hw // 2, hw // 2), dtype=np.float32)
    tx, tw, tdy = torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(dy)
    assert torch.equal(torch.from_numpy(oracle_lib.conv1x1s2(x, w)), torch.nn.functional.conv2d(tx.double(), tw.double(), None, 2, 0).float())
    assert torch.equal(torch.from_numpy(oracle_lib.conv1x1s2(dy, w, flip=True)),
                       torch.nn.grad.conv2d_input(x.shape, tw.double(), tdy.double(), 2, 0).float())


def test_k9_plan_and_argument_errors():
    import ctypes
    k = _native.default_kernels()
    assert k.conv1x1s2_supported((128, 16, 32, 32), 32) and k.conv1x1s2_supported((3, 32, 16, 16), 64)
    assert k.conv1x1s2_supported((128, 32, 16, 16), 16, flip=True) and k.conv1x1s2_supported((3, 64, 8, 8), 32, flip=True)
    assert not k.conv1x1s2_supported((128, 16, 32, 32), 16) and not k.conv1x1s2_supported((128, 32, 16, 16), 16)
    assert not k.conv1x1s2_supported((128, 16, 32, 32), 32, flip=True) and not k.conv1x1s2_supported((0, 16, 32, 32), 32)
    lib = _native.load_library()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    p -= p % 16
    f = lib.ursa_conv1x1s2_f32
    assert f(p, p, p, 128, 16, 32, 32, 32, 0x2, None) == -4
    assert f(None, p, p, 128, 16, 32, 32, 32, 0, None) == -1
    assert f(p, p, p, 0, 16, 32, 32, 32, 0, None) == -2
    assert f(p + 4, p, p, 128, 16, 32, 32, 32, 0, None) == -3
    assert f(p, p, p, 128, 16, 16, 32, 32, 0, None) == -5
    assert f(p, p, p, 128, 16, 32, 32, 32, 1, None) == -5               # the flipped form of that layer is (32, 16, 16)
