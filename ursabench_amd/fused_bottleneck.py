"""K13: `conv1x1(relu(bn(x)))` of the Bottleneck pre-activation unit without storing the normalised activation.

Reference ops: URSABench/models/preresnet.py:70-87 - `out = self.bn1(x); out = self.relu(out); out = self.conv1(out)` and the same
around `bn3` / `conv3` - and their backward inside hamiltorch's potential gradient (URSABench/inference/hmc.py:71-75). At the HMC
configuration's 1,024-row batches these activations are 67-268 MB and K6's second forward launch (read x, write relu(bn(x))) is
pure traffic: here K6's FIRST launch (statistics; with the previous block's `out += residual` folded in) is followed by a
64-thread-per-channel merge (`ursa_bn_stats_f32`), and K12's GEMM applies `relu(fma(x, scale, shift))` - K6's own expression - to
the rows of x as it stages them (`ursa_preact_conv1x1_f32`); the weight gradient rebuilds the same rows the same way
(`ursa_preact_wgrad1x1_partial_f32`), the input gradient is K12's flipped launch and the BatchNorm backward K6's two launches,
which need x and the saved block only. Every value has the bits of the K6 (two-launch form) + K12 path it replaces
(tests/test_fused_bottleneck_gpu.py).

K14: where the layer narrows (a block's conv1: 64 -> 16 ...), the convolution's input gradient dh is the widest tensor of the
unit and K6's backward would read it twice after K12 wrote it; `ursa_preact_conv1x1_bwd_*` run the memory-bound flipped GEMM twice
instead - once for the BatchNorm backward's two sums, once more applying K6's dx expression - and dh never exists in memory
(`URSA_K14=0`: K12's launch + K6's two).

Taken by `models._PreActBottleneck` when `eligible`; `URSA_K13=0` keeps the K6 + K12 launches (A/B).
"""
import os

import torch
from torch.autograd.function import once_differentiable

from . import _native, fused_bn, fused_conv

_on = os.environ.get('URSA_K13', '1') != '0'
_k14 = os.environ.get('URSA_K14', '1') != '0'
#: below this many bytes of activation K6 may take its one-pass form (one workgroup per channel, a different summation tree)
#: and the fold saves little: the K6 + K12 launches stay
MIN_BYTES = 8 << 20


def enabled(flag=None):
    """Query / set the process-wide switch; returns the previous value."""
    global _on
    old = _on
    if flag is not None:
        _on = bool(flag)
    return old


def recompute_backward(flag=None):
    """Query / set whether the narrowing layers' backward takes K14 (two GEMM passes, no stored input gradient)."""
    global _k14
    old = _k14
    if flag is not None:
        _k14 = bool(flag)
    return old


def _aligned(t):
    t = t.contiguous()
    return t.clone() if t.data_ptr() % 16 else t


class _BNReLUConv1x1(torch.autograd.Function):
    """(z, y) = (a [+ b], conv1x1(relu(bn(z)), w)); z is returned only in the residual form (b given)."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, running_mean, running_var, eps, momentum, w, ws_floats, k14):
        ctx.set_materialize_grads(False)
        K = _native.default_kernels()
        C = a.shape[1]
        save = a.new_empty(4, C)
        z = torch.empty_like(a) if b is not None else None
        K.bn_stats(a, gamma, beta, running_mean, running_var, save, a.new_empty(_native.bn_ws_floats(C)), eps=eps, momentum=momentum,
                   addend=b, z_out=z)
        x = a if b is None else z
        y = K.preact_conv1x1(x, save, w)
        ctx.save_for_backward(x, gamma, beta, save, w)
        ctx.residual, ctx.ws_floats, ctx.weight, ctx.k14 = b is not None, ws_floats, w, k14
        ctx.sink = getattr(fused_conv._tls, 'sink', None)
        if b is None:
            return y
        return z, y

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        x, gamma, beta, save, w = ctx.saved_tensors
        dz, dy = grads if ctx.residual else (None, grads[0])
        K = _native.default_kernels()
        C = x.shape[1]
        if dy is None:                                          # y unused: only the sum's own gradient flows
            return dz, (dz if ctx.residual else None), None, None, None, None, None, None, None, None, None
        dy = _aligned(dy)
        dw = None
        if ctx.needs_input_grad[8]:
            def first():
                return K.preact_wgrad1x1_partial(x, save, dy, w.shape, x.new_empty(ctx.ws_floats))
            if ctx.sink is not None:
                ctx.sink.append((ctx.sink.launch(first, x, dy, save), ctx.weight))
            else:
                dw = torch.empty_like(w)
                K.conv_wgrad_reduce([(first(), dw)])
        dx = torch.empty_like(x)
        dgb = x.new_empty(2, C)
        dz = None if dz is None else _aligned(dz)
        if ctx.k14:                                             # K14: the flipped GEMM twice, its result never stored
            K.preact_conv1x1_bwd(dy, w, x, save, gamma, dx, dgb[0], dgb[1], dz=dz)
        else:
            dh = K.conv1x1(dy, w, flip=True)
            K.bn_relu_backward(x, dh, dx, gamma, beta, save[0], save[1], dgb[0], dgb[1], x.new_empty(_native.bn_ws_floats(C)), relu=True,
                               dz=dz, two_launch=True, gate=save[2:])
        return dx, (dx if ctx.residual else None), dgb[0], dgb[1], None, None, None, None, dw, None, None


def eligible(bn, conv, a, b=None):
    """Whether `conv(relu(bn(a [+ b])))` takes K13: a training-mode BatchNorm2d with affine parameters and a 1x1 / stride 1
    `fused_conv.Conv2d` the library covers in all three directions, gradients recorded, contiguous fp32 NCHW on a HIP device, a
    large activation, no parity instrument installed (it observes relu(bn(x)), which is not stored here)."""
    if not (_on and fused_bn._on and fused_conv._on and fused_conv._k8 and fused_conv._k12 and fused_bn._probe is None):
        return False
    if fused_bn.held_allowed():                                 # whoever opted into K6's held form keeps it (its own one-launch forms)
        return False
    if not (isinstance(conv, fused_conv.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None):
        return False
    w = conv.weight
    if not (fused_bn._fusable(bn, a) and isinstance(bn, torch.nn.BatchNorm2d) and a.dim() == 4 and a.numel() * 4 >= MIN_BYTES
            and a.data_ptr() % 16 == 0 and w.dtype == torch.float32 and w.is_contiguous() and w.data_ptr() % 16 == 0
            and w.device == a.device and torch.is_grad_enabled() and (a.requires_grad or w.requires_grad)):
        return False
    if b is not None and not (b.shape == a.shape and b.dtype == a.dtype and b.device == a.device and b.is_contiguous()
                              and b.data_ptr() % 16 == 0 and a.data_ptr() != b.data_ptr()):
        return False
    if not bn.training or bn.running_mean is None or not bn.track_running_stats or bn.momentum is None:
        return False                                            # evaluation / cumulative averages: fused_bn's own handling
    if a.numel() // a.shape[1] < 2:
        return False
    key = (tuple(a.shape), w.requires_grad)
    plan = conv.__dict__.get('_ursa_k13')
    if plan is None or plan[0] != key:
        K = _native.default_kernels()
        N, cin, H, W = a.shape
        ok = (K.preact_conv1x1_supported(a.shape, conv.out_channels) and K.conv1x1_supported((N, conv.out_channels, H, W), cin, flip=True))
        ws = K.conv_wgrad_ws_floats(a.shape, conv.out_channels, 1, 1) if w.requires_grad else 0
        # K14 where it pays: at 8 x 8 (256 <- 64 channels: the kernel holds 64 + 128 weight / accumulator registers per lane and
        # runs one wave per SIMD) the two passes take 98 us against 93 for K12's launch + K6's two (profiles/r06_k12_bench.json)
        k14 = K.preact_conv1x1_bwd_nl((N, conv.out_channels, H, W), cin) > 0 and H * W >= 256
        plan = conv.__dict__['_ursa_k13'] = (key, bool(ok) and (ws > 0 or not w.requires_grad), ws, k14)
    return plan[1]


def bn_relu_conv1x1(bn, conv, x):
    """`x`: a tensor or a pending residual sum (a, b). Returns (z, conv(relu(bn(z)))) with z the summed tensor (x itself for a plain
    tensor). The caller has checked `eligible`."""
    a, b = x if isinstance(x, tuple) else (x, None)
    if bn.num_batches_tracked is not None:                      # None inside util.deferred_bn_counters
        bn.num_batches_tracked.add_(1)
    _, _, ws, k14 = conv.__dict__['_ursa_k13']
    out = _BNReLUConv1x1.apply(a, b, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, conv.weight, ws, k14 and _k14)
    return (a, out) if b is None else out
