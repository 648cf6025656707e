"""The convolutions of the benchmark networks' training step as gfx950 launches (K7, K8, K9: include/ursa_hip.h).

`Conv2d` is what `ursabench_amd.models` instantiates where the reference's networks write `nn.Conv2d(...)`
(URSABench/models/preresnet.py:25-27,62-64,100,130-136). It IS an `nn.Conv2d` (same parameters, state_dict keys,
initialisation). For the layer shapes the library covers - every convolution of the BasicBlock pre-activation ResNets - on
contiguous fp32 NCHW HIP tensors, WHILE GRADIENTS ARE RECORDED (a training / HMC step):

  forward and input gradient   K8 `ursa_conv3x3_f32` (3x3, stride 1 and 2) / K9 `ursa_conv1x1s2_f32` (1x1 stride 2 shortcuts) / K12
                               `ursa_conv1x1_f32` (the 1x1 stride 1 layers of the Bottleneck networks: a GEMM on the NCHW planes): one
                               launch each, direct convolution in exact fp32 on the matrix pipe, instead of MIOpen's Winograd /
                               implicit-GEMM launches (18.3 us -> 10.3 us per 3x3 layer and direction at the workload's sizes)
  weight gradient              K7 `ursa_conv_wgrad_f32`: two launches straight from the NCHW tensors, fixed summation order, instead
                               of MIOpen's sequence for these sizes (two layout transposes, a zero fill, an atomics-based implicit
                               GEMM, a transpose back: 92 of the 237 launches of a PreResNet-20 step in round 4)

Evaluation-mode forwards (no gradient recorded: the BMA predictive at thousands of rows per batch) stay MIOpen's - its
Winograd launch is the faster one there (bench.py `bma` leg: 23.7 k vs 21.9 k predictions/s).

`deferred()`: whoever owns the destination of the gradients (the chain engine: its flat arena) can take K7's second launch -
the fixed-order sum over the K slices - for ALL layers of a backward pass at once: the backward of a covered layer applied
inside the context runs the first launch only and records (partial sums, weight); `flush()` then writes every layer's dW
where the caller says, in one launch. Outside the context each layer's backward returns its weight gradient as autograd expects.

Anything else - other kernel sizes / strides / channel counts, bias, groups, dilation, host tensors, other dtypes or layouts,
double backward - takes `nn.Conv2d.forward`, the stock path, unchanged. On a HIP tensor the covered paths need
csrc/libursa_hip.so (no silent fallback: a missing library raises). `URSA_FUSED_CONV=0` in the environment, or
`enabled(False)`, selects the stock path everywhere; `URSA_FUSED_CONV_FWD=0` / `forward_enabled(False)` keeps K7 and hands
forward / input gradient back to MIOpen (A/B runs).
"""
import os
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import _native

_on = os.environ.get('URSA_FUSED_CONV', '1') != '0'
_k8 = os.environ.get('URSA_FUSED_CONV_FWD', '1') != '0'      # K8 (forward / input gradient) separately, for A/B runs
_k12 = os.environ.get('URSA_K12', '1') != '0'                # K12 (the 1x1 stride-1 layers of the Bottleneck networks), for A/B runs


def enabled(flag=None):
    """Query / set whether covered layers take this module's launches at all (process-wide)."""
    global _on
    old = _on
    if flag is not None:
        _on = bool(flag)
    return old


def forward_enabled(flag=None):
    """Query / set whether covered layers take K8 (forward / input gradient); K7 (weight gradient) is not affected."""
    global _k8
    old = _k8
    if flag is not None:
        _k8 = bool(flag)
    return old


_tls = threading.local()          # deferred(): the sink of the thread that runs the forward


class Sink(list):
    """What `deferred()` hands out: the list of (record of the first K7 launch, weight parameter), plus - when the caller gave a
    side stream - what it takes to run those launches BESIDE the rest of the backward pass: nothing in a backward pass depends
    on a weight gradient, so the K7 launches form a branch of their own (inside a captured step: a parallel branch of the
    hipGraph) that only the flush joins. `keep` holds the operands the side launches read until the join: autograd drops its
    reference to a gradient tensor as soon as the node returns, and the block could otherwise be handed to a later allocation
    of the main stream while the side launch still reads it."""

    def __init__(self, side=None):
        super().__init__()
        self.side, self.keep, self.forked = side, [], False

    def launch(self, fn, *operands):
        """fn() enqueues one first-K7 launch and returns its record: on the side stream (after everything enqueued so far on
        the current one) if there is one, else in place. `operands`: the tensors it reads."""
        if self.side is None:
            return fn()
        cur = torch.cuda.current_stream(self.side.device)
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            rec = fn()
        self.keep.append(operands)
        self.forked = True
        return rec

    def join(self):
        """The current stream waits for the side launches (before `flush`); the held operands are released."""
        if self.forked:
            torch.cuda.current_stream(self.side.device).wait_stream(self.side)
            self.forked = False
        self.keep.clear()


class deferred:
    """Context around a FORWARD pass: the backward of every covered layer applied inside leaves (record of the first K7
    launch, weight parameter) in the returned list instead of returning a weight gradient - the weight's `.grad` stays None -
    and the caller hands the list to `flush()` once `backward()` has returned. The decision is taken in the forward (this
    thread) and travels in the autograd context: the backward runs on autograd's device thread. `side`: a stream the first
    launches may run on beside the backward pass (`Sink`); the caller then calls `join()` on the list before `flush()`."""

    def __init__(self, side=None):
        self.side = side

    def __enter__(self):
        self.old = getattr(_tls, 'sink', None)
        _tls.sink = sink = Sink(self.side)
        return sink

    def __exit__(self, *exc):
        _tls.sink = self.old
        return False


def flush(pending, dest):
    """The second K7 launch for every layer recorded in `pending` (from `deferred()`), at once. dest(weight) -> the contiguous
    fp32 tensor of the weight's shape that receives its gradient (overwritten). A weight that occurs more than once (a layer
    applied twice in one forward) gets the sum of its gradients."""
    if not pending:
        return
    items, extra, seen = [], [], {}
    for rec, w in pending:
        if id(w) in seen:
            tmp = torch.empty_like(seen[id(w)])
            extra.append((seen[id(w)], tmp))
            items.append((rec, tmp))
        else:
            seen[id(w)] = d = dest(w)
            items.append((rec, d))
    _native.default_kernels().conv_wgrad_reduce(items)
    for d, tmp in extra:
        d.add_(tmp)


class _Conv(torch.autograd.Function):
    """y = conv2d(x, w, stride, padding=k // 2). Per call, decided in Conv2d.forward: the forward / dx by K8 or ATen (MIOpen),
    dw by K7 or ATen."""

    @staticmethod
    def forward(ctx, x, w, stride, ws_floats, k8_fwd, k8_bwd):
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.ws_floats, ctx.k8_bwd, ctx.weight, ctx.sink = stride, ws_floats, k8_bwd, w, getattr(_tls, 'sink', None)
        if k8_fwd:
            k = _native.default_kernels()
            if w.shape[2] == 3:
                return k.conv3x3(x, w, stride=stride)
            return k.conv1x1s2(x, w) if stride == 2 else k.conv1x1(x, w)
        return F.conv2d(x, w, None, stride, w.shape[2] // 2)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        s, pad = ctx.stride, w.shape[2] // 2
        k = _native.default_kernels()
        dy = dy.contiguous()
        if dy.data_ptr() % 16:                 # a contiguous view at an odd offset: the kernels want 16-byte aligned operands
            dy = dy.clone()
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = dw = None
        if need_dx and ctx.k8_bwd:
            if w.shape[2] == 3:
                dx = k.conv3x3(dy, w, flip=True, stride=s)
            else:
                dx = k.conv1x1s2(dy, w, flip=True) if s == 2 else k.conv1x1(dy, w, flip=True)
            need_dx = False
        if need_dw and ctx.ws_floats:
            if ctx.sink is not None:
                ctx.sink.append((ctx.sink.launch(lambda: k.conv_wgrad_partial(x, dy, w.shape, x.new_empty(ctx.ws_floats), s), x, dy), ctx.weight))
            else:
                ws = x.new_empty(ctx.ws_floats)
                dw = torch.empty_like(w)
                k.conv_wgrad(x, dy, dw, ws, s)
            need_dw = False
        if need_dx or need_dw:
            rx, rw, _ = torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1,
                                                            [need_dx, need_dw, False])
            dx, dw = (rx if need_dx else dx), (rw if need_dw else dw)
        return dx, dw, None, None, None, None


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose launches are K8's / K7's where they cover the layer (module docstring); the stock module otherwise."""

    def forward(self, x):
        ks, st, w = self.kernel_size[0], self.stride[0], self.weight
        if not (_on and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and self.bias is None
                and self.kernel_size in ((3, 3), (1, 1)) and self.padding == (ks // 2, ks // 2) and self.dilation == (1, 1)
                and self.groups == 1 and self.padding_mode == 'zeros' and st == self.stride[1] and w.dtype == torch.float32
                and x.is_contiguous() and w.is_contiguous() and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0):
            return super().forward(x)
        if not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)):
            # no gradient recorded (evaluation, the BMA predictive at 4,096 rows): MIOpen's Winograd launch is the faster one at
            # large batches (bench.py bma leg: 23.7 k vs 21.9 k predictions/s) - K8 is for the training step's sizes
            return super().forward(x)
        key = (tuple(x.shape), x.requires_grad, w.requires_grad, _k8, _k12)
        plan = self.__dict__.get('_ursa_plan')
        if plan is None or plan[0] != key:                     # what the library covers for this call: asked once per shape
            k = _native.default_kernels()
            N, cin, H, W = x.shape
            k8, k9, k12 = _k8 and ks == 3 and st in (1, 2), _k8 and ks == 1 and st == 2, _k8 and _k12 and ks == 1 and st == 1
            fwd = ((k8 and k.conv3x3_supported(x.shape, self.out_channels, stride=st)) or (k9 and k.conv1x1s2_supported(x.shape, self.out_channels))
                   or (k12 and k.conv1x1_supported(x.shape, self.out_channels)))
            bwd = x.requires_grad and ((k8 and k.conv3x3_supported((N, self.out_channels, H // st, W // st), cin, flip=True, stride=st))
                                       or (k9 and k.conv1x1s2_supported((N, self.out_channels, H // 2, W // 2), cin, flip=True))
                                       or (k12 and k.conv1x1_supported((N, self.out_channels, H, W), cin, flip=True)))
            ws = k.conv_wgrad_ws_floats(x.shape, self.out_channels, ks, st) if w.requires_grad and (_k12 or ks != 1 or st != 1) else 0
            plan = self.__dict__['_ursa_plan'] = (key, ws, bool(fwd), bool(bwd))
        _, ws, fwd, bwd = plan
        if not (fwd or bwd or ws):
            return super().forward(x)
        return _Conv.apply(x, w, st, ws, fwd, bwd)
