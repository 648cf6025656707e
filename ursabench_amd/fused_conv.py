"""The weight gradient of the benchmark networks' 3x3 convolutions as gfx950 launches (K7, include/ursa_hip.h).

`Conv2d` is what `ursabench_amd.models` instantiates where the reference's networks write `nn.Conv2d(...)`
(URSABench/models/preresnet.py:25-27,62-64,100). It IS an `nn.Conv2d` (same parameters, state_dict keys, initialisation);
the forward and the input gradient stay MIOpen's. What changes, for the layer shapes K7 covers, on contiguous fp32 NCHW HIP
tensors while gradients are recorded: the weight gradient of `loss.backward()` (URSABench/inference/sghmc.py:80) is computed
by `ursa_conv3x3_wgrad_f32` - two launches straight from the NCHW tensors, exact fp32, fixed summation order - instead of
MIOpen's sequence for these sizes (two layout transposes, a zero fill, an atomics-based implicit GEMM, a transpose back:
92 of the 237 launches of a PreResNet-20 step, profiles/r05_step_timeline.json).

Anything else - other kernel sizes / strides / channel counts, bias, groups, dilation, host tensors, other dtypes or layouts,
double backward - takes `nn.Conv2d.forward`, the stock path, unchanged. On a HIP tensor the K7 path needs
csrc/libursa_hip.so (no silent fallback: a missing library raises). `URSA_FUSED_CONV=0` in the environment, or
`enabled(False)`, selects the stock path everywhere (A/B runs).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import _native

_on = os.environ.get('URSA_FUSED_CONV', '1') != '0'


def enabled(flag=None):
    """Query / set whether covered layers take K7's weight gradient (process-wide)."""
    global _on
    old = _on
    if flag is not None:
        _on = bool(flag)
    return old


class _Conv3x3(torch.autograd.Function):
    """y = conv2d(x, w, stride, padding=1): forward and dx are ATen's (MIOpen), dw is K7's."""

    @staticmethod
    def forward(ctx, x, w, stride, ws_floats):
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.ws_floats = stride, ws_floats
        return F.conv2d(x, w, None, stride, 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        s = ctx.stride
        dy = dy.contiguous()
        if dy.data_ptr() % 16:                 # a contiguous view at an odd offset: K7 wants 16-byte aligned operands
            dy = dy.clone()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            _native.default_kernels().conv3x3_wgrad(x, dy, dw, x.new_empty(ctx.ws_floats), s)
        return dx, dw, None, None


def _covered(m, x):
    """Scratch floats if K7 takes this call's weight gradient, else 0."""
    if not (_on and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and torch.is_grad_enabled()
            and m.weight.requires_grad and m.bias is None and m.kernel_size == (3, 3) and m.padding == (1, 1)
            and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == 'zeros' and m.stride[0] == m.stride[1]
            and m.weight.dtype == torch.float32 and x.is_contiguous() and m.weight.is_contiguous()
            and x.data_ptr() % 16 == 0):
        return 0
    return _native.default_kernels().conv3x3_wgrad_ws_floats(x.shape, m.out_channels, m.stride[0])


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose weight gradient is K7's where K7 covers the layer (module docstring); the stock module otherwise."""

    def forward(self, x):
        n = _covered(self, x)
        if n:
            return _Conv3x3.apply(x, self.weight, self.stride[0], n)
        return super().forward(x)
