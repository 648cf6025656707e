"""The 3x3 convolutions of the benchmark networks with their WEIGHT GRADIENT taken on channels-last operands.

Convolutions stay stock MIOpen (north_star). What this module changes is which tensors the weight-gradient
call is handed. MIOpen's fastest weight-gradient kernel for these layers (`igemm_wrw_gtcx35_nhwc_fp32`, chosen by its own
exhaustive search) works on NHWC tensors; handed the networks' NCHW activations it transposes both operands itself and
transposes the result back: `batched_transpose_*` was 15 % of a PreResNet-20 training step's kernel time and 16 % of a
PreResNet-164 HMC step (profiles/r03_bench_kernel_stats.csv, r03_c5_kernel_stats.csv). Both operands of every such
gradient are outputs of K6 launches - the convolution's input is a `relu(bn(.))`, its output gradient the backward of the
next BatchNorm - and K6 can store them channels-last as well at next to no cost (`fused_bn.bn_relu(..., twins=)`,
csrc/ursa_bn.hip k_bn_fwd_apply_t / k_bn_bwd_dx_t). `conv2d` below is `nn.Conv2d.forward` whose backward

  * takes the input gradient as before: NCHW, MIOpen's Winograd kernel;
  * takes the weight gradient with `aten.convolution_backward` on the two NHWC twins (found as `._ursa_nhwc` on the
    input and on the incoming gradient) - MIOpen then runs the same kernel WITHOUT its transposes - and returns it in the
    parameter's own layout;
  * falls back to the stock backward whenever a twin is missing (host tensors, other dtypes, a gradient that autograd
    had to accumulate, `URSA_NHWC_WGRAD=0`).

`PYTORCH_MIOPEN_SUGGEST_NHWC=1` must be in the environment before the first convolution (PyTorch only hands MIOpen
channels-last descriptors then); importing this module sets it if it is unset. It changes nothing for NCHW tensors
(measured: tools/exp/nhwc_env_probe.py)."""
import os

os.environ.setdefault('PYTORCH_MIOPEN_SUGGEST_NHWC', '1')

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch.autograd.function import once_differentiable  # noqa: E402

stats = dict(nhwc_wgrad=0, stock_backward=0)       # which backward ran (tests assert the fast path is the one taken)


def _twin(t, like=None):
    tw = getattr(t, '_ursa_nhwc', None)
    if tw is None or tw.shape != t.shape or tw.device != t.device or tw.dtype != t.dtype:
        return None
    return tw


class _ConvNHWCWgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, x_nhwc, stride, padding, dilation, groups):
        ctx.save_for_backward(x, weight, x_nhwc)
        ctx.conf = (stride, padding, dilation, groups, None if bias is None else list(bias.shape))
        return F.conv2d(x, weight, bias, stride, padding, dilation, groups)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, x_nhwc = ctx.saved_tensors
        stride, padding, dilation, groups, bias_sizes = ctx.conf
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = bias_sizes is not None and ctx.needs_input_grad[2]
        dy_nhwc = _twin(dy)
        conv_bwd = torch.ops.aten.convolution_backward
        if dy_nhwc is None or not need_w:
            stats['stock_backward'] += 1
            dx, dw, db = conv_bwd(dy.contiguous(), x, weight, bias_sizes, stride, padding, dilation, False, [0, 0], groups,
                                  [need_x, need_w, need_b])
            return dx, dw, db, None, None, None, None, None
        stats['nhwc_wgrad'] += 1
        dx = None
        if need_x:
            dx = conv_bwd(dy, x, weight, None, stride, padding, dilation, False, [0, 0], groups, [True, False, False])[0]
        _, dw, db = conv_bwd(dy_nhwc, x_nhwc, weight, bias_sizes, stride, padding, dilation, False, [0, 0], groups,
                             [False, True, need_b])
        return dx, dw.contiguous(), db, None, None, None, None, None    # (MIOpen returns dw channels-last; the parameter is not)


def conv2d(conv, x):
    """`conv(x)` for an `nn.Conv2d` with zero-padding; with a twin on `x` (and gradients enabled) the weight gradient will
    be taken on channels-last operands."""
    tw = _twin(x) if torch.is_grad_enabled() else None
    if tw is None or conv.padding_mode != 'zeros' or isinstance(conv.padding, str):
        return conv(x)
    return _ConvNHWCWgrad.apply(x, conv.weight, conv.bias, tw, list(conv.stride), list(conv.padding), list(conv.dilation),
                                conv.groups)
