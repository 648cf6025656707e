"""relu(bn(x)) of the benchmark networks' pre-activation blocks as gfx950 launches (K6, include/ursa_hip.h).

`bn_relu(bn, x)` is what `ursabench_amd.models` calls where the reference's networks write
`self.relu(self.bn(x))` (URSABench/models/preresnet.py:40-41,45-46,76-85,146; wideresnet.py:47,49,117). `bn` stays a
plain `nn.BatchNorm2d` (same state_dict keys, same running-statistics / momentum / counter semantics as
torch.nn.modules.batchnorm._BatchNorm.forward); only the launches differ:

  training, fp32 contiguous NCHW on a HIP device:  2 launches forward, 2 backward (autograd.Function below)
  evaluation, no gradient needed:                  1 launch
  `add_bn_relu(bn, (a, b))`: the residual sum z = a + b that ends a block (`out += residual`, preresnet.py:49-52) and
  the relu(bn(z)) that opens the next one in the SAME launches: the add folds into the statistics pass, and the
  accumulation of z's two gradients (autograd's own add launch) into the backward's second pass
  anything else (host tensors, other dtypes / layouts, momentum=None, no affine, gradients through an
  evaluation-mode layer):                          `F.relu(bn(x))`, the stock path

On a HIP tensor the fused path needs csrc/libursa_hip.so (no silent fallback: a missing library raises).
`URSA_FUSED_BN=0` in the environment, or `enabled(False)`, selects the stock path everywhere (A/B runs);
`URSA_BN_TWO_LAUNCH=1` keeps the two-launch kernels where the one-pass form would apply.
"""
import os

import torch
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import _native

_on = os.environ.get('URSA_FUSED_BN', '1') != '0'
_two_launch = os.environ.get('URSA_BN_TWO_LAUNCH', '0') == '1'     # A/B: never take the one-pass form


def enabled(flag=None):
    """Query / set the process-wide switch; returns the previous value."""
    global _on
    old = _on
    if flag is not None:
        _on = bool(flag)
    return old


class _BNReLUTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, relu):
        K = _native.default_kernels()
        C = x.shape[1]
        y = torch.empty_like(x)
        stats = x.new_empty(2, C)                       # save_mean, save_invstd
        ws = x.new_empty(_native.bn_ws_floats(C))
        K.bn_relu_forward(x, y, weight, bias, running_mean, running_var, stats[0], stats[1], ws, eps=eps,
                          momentum=momentum, relu=relu, two_launch=_two_launch)
        ctx.save_for_backward(x, weight, bias, stats)
        ctx.relu = relu
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, bias, stats = ctx.saved_tensors
        K = _native.default_kernels()
        C = x.shape[1]
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dwb = x.new_empty(2, C)
        ws = x.new_empty(_native.bn_ws_floats(C))
        K.bn_relu_backward(x, dy, dx, weight, bias, stats[0], stats[1], dwb[0], dwb[1], ws, relu=ctx.relu,
                           two_launch=_two_launch)
        return dx, dwb[0], dwb[1], None, None, None, None, None


class _AddBNReLUTrain(torch.autograd.Function):
    """(z, y) = (a + b, relu(bn(a + b))). backward: d(a) = d(b) = dz + bn_relu_backward(dy) in the same two launches."""

    @staticmethod
    def forward(ctx, a, b, weight, bias, running_mean, running_var, eps, momentum, relu):
        ctx.set_materialize_grads(False)                 # an unused output's gradient arrives as None, not as zeros
        K = _native.default_kernels()
        C = a.shape[1]
        z, y = torch.empty_like(a), torch.empty_like(a)
        stats = a.new_empty(2, C)
        ws = a.new_empty(_native.bn_ws_floats(C))
        K.bn_relu_forward(a, y, weight, bias, running_mean, running_var, stats[0], stats[1], ws, eps=eps,
                          momentum=momentum, relu=relu, addend=b, z_out=z, two_launch=_two_launch)
        ctx.save_for_backward(z, weight, bias, stats)
        ctx.relu = relu
        return z, y

    @staticmethod
    @once_differentiable
    def backward(ctx, dz, dy):
        z, weight, bias, stats = ctx.saved_tensors
        K = _native.default_kernels()
        C = z.shape[1]
        if dy is None:                                   # y unused: only the sum's own gradient flows
            return dz, dz, None, None, None, None, None, None, None
        dy = dy.contiguous()
        dz = None if dz is None else dz.contiguous()
        dx = torch.empty_like(z)
        dwb = z.new_empty(2, C)
        ws = z.new_empty(_native.bn_ws_floats(C))
        K.bn_relu_backward(z, dy, dx, weight, bias, stats[0], stats[1], dwb[0], dwb[1], ws, relu=ctx.relu, dz=dz,
                           two_launch=_two_launch)
        return dx, dx, dwb[0], dwb[1], None, None, None, None, None


def _fusable(bn, x):
    """Plain BatchNorm1d/2d/3d (SyncBatchNorm's statistics span processes: stock path) with affine parameters on a
    contiguous fp32 [N, C, *] HIP tensor; the launch grid carries the channel index in its y dimension."""
    return (_on and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3 and x.is_contiguous() and x.numel() > 0
            and x.shape[1] <= 65535
            and isinstance(bn, torch.nn.modules.batchnorm._BatchNorm) and not isinstance(bn, torch.nn.SyncBatchNorm)
            and bn.affine and bn.weight.dtype == torch.float32 and bn.weight.device == x.device
            and bn.weight.is_contiguous() and bn.bias.is_contiguous())


def _stock(bn, x, relu):
    """The reference's own ops: the module's forward, then the in-place ReLU of `nn.ReLU(inplace=True)`."""
    y = bn(x)
    return F.relu(y, inplace=True) if relu else y


def bn_relu(bn, x, relu=True):
    """relu(bn(x)) (or bn(x) with relu=False) with the module semantics of nn.BatchNorm2d."""
    if not _fusable(bn, x):
        return _stock(bn, x, relu)
    use_batch_stats = bn.training or bn.running_mean is None
    if not use_batch_stats:
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad)
        if needs_grad:
            return _stock(bn, x, relu)
        y = torch.empty_like(x)
        _native.default_kernels().bn_relu_eval(x, y, bn.weight, bn.bias, bn.running_mean, bn.running_var, eps=bn.eps,
                                               relu=relu)
        return y
    track = bn.training and bn.track_running_stats and bn.running_mean is not None
    if track and bn.momentum is None:                   # cumulative average: the factor depends on a device counter
        return _stock(bn, x, relu)
    if x.numel() // x.shape[1] < 2:
        raise ValueError(f'Expected more than 1 value per channel when training, got input size {tuple(x.shape)}')
    if track and bn.num_batches_tracked is not None:    # None inside util.deferred_bn_counters
        bn.num_batches_tracked.add_(1)
    rm, rv = (bn.running_mean, bn.running_var) if track else (None, None)
    return _BNReLUTrain.apply(x, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum if track else 0.0, relu)


def add_bn_relu(bn, x, relu=True):
    """`x` is a tensor, or a pending residual sum `(a, b)` standing for a + b (what a pre-activation block returns here
    instead of adding). Returns `(z, relu(bn(z)))` with z the summed tensor: the next block needs z for its own
    shortcut. A plain tensor, host tensors and everything `bn_relu` sends to the stock path: z = a + b with torch's add,
    then `bn_relu` - op for op the reference's `out += residual` followed by the next block's bn / relu."""
    if not isinstance(x, tuple):
        return x, bn_relu(bn, x, relu)
    a, b = x
    same = (b.shape == a.shape and b.dtype == a.dtype and b.device == a.device and b.is_contiguous()
            and a.data_ptr() != b.data_ptr())
    if not (same and _fusable(bn, a)):
        z = a + b
        return z, bn_relu(bn, z, relu)
    use_batch_stats = bn.training or bn.running_mean is None
    if not use_batch_stats:
        needs_grad = torch.is_grad_enabled() and (a.requires_grad or b.requires_grad or bn.weight.requires_grad
                                                  or bn.bias.requires_grad)
        if needs_grad:
            z = a + b
            return z, _stock(bn, z, relu)
        z, y = torch.empty_like(a), torch.empty_like(a)
        _native.default_kernels().bn_relu_eval(a, y, bn.weight, bn.bias, bn.running_mean, bn.running_var, eps=bn.eps,
                                               relu=relu, addend=b, z_out=z)
        return z, y
    track = bn.training and bn.track_running_stats and bn.running_mean is not None
    if track and bn.momentum is None:
        z = a + b
        return z, _stock(bn, z, relu)
    if a.numel() // a.shape[1] < 2:
        raise ValueError(f'Expected more than 1 value per channel when training, got input size {tuple(a.shape)}')
    if track and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    rm, rv = (bn.running_mean, bn.running_var) if track else (None, None)
    return _AddBNReLUTrain.apply(a, b, bn.weight, bn.bias, rm, rv, bn.eps, bn.momentum if track else 0.0, relu)
