"""CPU: the host logic around K13 - who takes the fused `conv1x1(relu(bn(x)))` and who does not - and the library's answers that
need no device. (The launches themselves: tests/test_fused_bottleneck_gpu.py.)"""
import ctypes

import torch
import torch.nn as nn

from ursabench_amd import _native, fused_bottleneck, fused_conv, models

OK, ENULL, ESIZE, EALIGN, EVALUE = 0, -1, -2, -3, -5          # include/ursa_hip.h


def test_host_tensors_never_take_k13():
    blk = models._PreActBottleneck(64, 16).train()
    x = torch.randn(64, 64, 32, 32, requires_grad=True)             # 16 MB: large enough, but on the host
    assert not fused_bottleneck.eligible(blk.bn1, blk.conv1, x)
    y, sc = blk(x)                                                  # the stock ops, op for op the reference's
    assert y.shape == (64, 64, 32, 32) and sc is x
    (y + sc).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in blk.parameters())


def test_which_layers_the_library_covers_is_a_host_computation():
    K = _native.HipKernels()
    for cin, cout, h in [(64, 16, 32), (16, 64, 32), (128, 32, 16), (32, 128, 16), (256, 64, 8), (64, 256, 8), (64, 32, 32), (128, 64, 16)]:
        assert K.preact_conv1x1_supported((1024, cin, h, h), cout)
        assert K.conv_wgrad_ws_floats((1024, cin, h, h), cout, 1, 1) > 0
    for cin, cout, h in [(48, 16, 32), (64, 16, 16), (16, 64, 8), (64, 24, 32)]:
        assert not K.preact_conv1x1_supported((8, cin, h, h), cout)
    assert not K.preact_conv1x1_supported((0, 64, 32, 32), 16)


def test_argument_errors_come_back_as_codes_without_a_device():
    lib = _native.HipKernels().lib
    one = ctypes.c_void_p(16)                                       # any aligned non-null address: nothing is dereferenced before the checks
    assert lib.ursa_bn_stats_f32(None, None, None, one, one, None, None, one, one, 4, 16, 64, 1e-5, 0.1, None) == ENULL
    assert lib.ursa_bn_stats_f32(one, one, None, one, one, None, None, one, one, 4, 16, 64, 1e-5, 0.1, None) == ENULL   # addend without z_out
    assert lib.ursa_bn_stats_f32(one, None, None, one, one, one, None, one, one, 4, 16, 64, 1e-5, 0.1, None) == ENULL   # one running buffer
    assert lib.ursa_bn_stats_f32(one, None, None, one, one, None, None, one, one, 1, 16, 1, 1e-5, 0.1, None) == EVALUE  # one value per channel
    assert lib.ursa_bn_stats_f32(one, None, None, one, one, None, None, one, one, 0, 16, 64, 1e-5, 0.1, None) == OK     # empty batch
    assert lib.ursa_preact_conv1x1_f32(one, None, one, one, 4, 64, 16, 32, 32, None) == ENULL
    assert lib.ursa_preact_conv1x1_f32(one, one, one, one, 4, 48, 16, 32, 32, None) == EVALUE
    assert lib.ursa_preact_conv1x1_f32(ctypes.c_void_p(20), one, one, one, 4, 64, 16, 32, 32, None) == EALIGN
    assert lib.ursa_preact_wgrad1x1_partial_f32(one, None, one, one, 1 << 30, 4, 64, 16, 32, 32, None) == ENULL
    assert lib.ursa_preact_wgrad1x1_partial_f32(one, one, one, one, 8, 4, 64, 16, 32, 32, None) == ESIZE
    assert lib.ursa_preact_wgrad1x1_partial_f32(one, one, one, one, 1 << 30, 4, 16, 16, 32, 32, None) == EVALUE          # 16 -> 16: not covered


def test_k14_coverage_and_argument_errors_without_a_device():
    K = _native.HipKernels()
    lib = K.lib
    for cd, cx, h in [(16, 64, 32), (32, 128, 16), (64, 256, 8), (32, 64, 32), (64, 128, 16)]:
        assert K.preact_conv1x1_bwd_nl((1024, cd, h, h), cx) >= 512
    assert K.preact_conv1x1_bwd_nl((1024, 16, 32, 32), 64) == 1024 and K.preact_conv1x1_bwd_nl((8, 16, 32, 32), 64) == 8 * 8     # small batches: an image cut into up to 8 pieces (two chunks each)
    for cd, cx, h in [(64, 16, 32), (16, 16, 32), (16, 64, 16), (48, 64, 32)]:               # widening / uncovered: K12 + K6
        assert K.preact_conv1x1_bwd_nl((1024, cd, h, h), cx) == 0
    one = ctypes.c_void_p(16)
    assert lib.ursa_preact_conv1x1_bwd_sums_f32(one, one, one, one, None, 4, 16, 64, 32, 32, None) == ENULL
    assert lib.ursa_preact_conv1x1_bwd_sums_f32(one, one, one, one, one, 4, 64, 16, 32, 32, None) == EVALUE
    assert lib.ursa_preact_conv1x1_bwd_sums_f32(one, one, one, one, ctypes.c_void_p(8), 4, 16, 64, 32, 32, None) == EALIGN
    assert lib.ursa_preact_conv1x1_bwd_dx_f32(one, one, one, one, None, None, one, 4, 16, 64, 32, 32, None) == ENULL
    assert lib.ursa_preact_conv1x1_bwd_dx_f32(one, one, one, one, one, None, one, 0, 16, 64, 32, 32, None) == ESIZE
    assert lib.ursa_bn_bwd_coef_f32(None, 4, one, one, one, one, one, 4096, 64, None) == ENULL
    assert lib.ursa_bn_bwd_coef_f32(one, 0, one, one, one, one, one, 4096, 64, None) == ESIZE
    assert lib.ursa_bn_bwd_coef_f32(ctypes.c_void_p(8), 4, one, one, one, one, one, 4096, 64, None) == EALIGN


def test_switch_and_plain_modules():
    old = fused_bottleneck.enabled(False)
    try:
        assert fused_bottleneck.enabled() is False
    finally:
        fused_bottleneck.enabled(old)
    old = fused_bottleneck.recompute_backward(False)
    assert fused_bottleneck.recompute_backward(old) is False and fused_bottleneck.recompute_backward() is old
    bn, x = nn.BatchNorm2d(64), torch.randn(64, 64, 32, 32, requires_grad=True)
    assert not fused_bottleneck.eligible(bn, nn.Conv2d(64, 16, 1, bias=False), x)          # not a fused_conv.Conv2d
    assert not fused_bottleneck.eligible(bn, fused_conv.Conv2d(64, 16, 1, bias=True), x)
