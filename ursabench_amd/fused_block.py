"""The pre-activation unit `conv(relu(bn(x)))` (+ `out += residual`) of the BasicBlock ResNets as one gfx950 launch each way
(K10, include/ursa_hip.h; URSABench/models/preresnet.py:33-52).

K6 (`fused_bn`) and K8 / K7 (`fused_conv`) run a block as `statistics -> normalise -> convolution` three launches at a time,
each re-reading what the previous one wrote. Here the chain of a training step is, per unit,

    forward    ONE launch   K8 with the BatchNorm + ReLU applied while the input tile is staged (statistics merged from the
                            partial sums the PRODUCER of that tensor left) and the (sum, sum of squares) of its own output -
                            after `+= residual` where the block ends - taken from the accumulators: relu(bn(x)) is never stored
    backward   THREE        K7's first launch with the same staging transform (weight gradient), K8's input-gradient launch
                            that also gates its result with the ReLU mask and leaves the two sums of the BatchNorm backward,
                            and K6's `dx` launch (which also adds the gradient arriving over the shortcut)

instead of 3 + 4 (+ autograd's add). A PreResNet-20 step: 146 -> ~100 launches. Same arithmetic as the unfused launches, op for
op: the convolutions are K8's / K7's, the statistics are double sums rounded once (so the same floats whatever the summation
tree), normalise / gate / dx are K6's expressions - tests/test_fused_block_gpu.py pins fused == unfused bit for bit.

`trunk(model, x)` runs conv1 .. layer3 .. bn .. relu of a `models.PreResNet` built from BasicBlocks this way and returns the
rectified activation the pooling takes; `models.PreResNet.forward` calls it when `eligible()` says so: training-mode batch
statistics, gradients recorded, contiguous fp32 NCHW on a HIP device, every layer shape covered, no parity instrument
(`fused_bn.GateProbe`) installed - the instrument observes relu(bn(x)), which does not exist here; with one installed the
network takes the K6 / K8 launches, which these are pinned equal to. `URSA_FUSED_BLOCK=0` / `enabled(False)`: never.
On a HIP tensor the path needs csrc/libursa_hip.so (no silent fallback: a missing library raises).
"""
import os
import threading
import weakref

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from . import _native, fused_bn, fused_conv

_on = os.environ.get('URSA_FUSED_BLOCK', '1') != '0'
_pair = os.environ.get('URSA_BWD_PAIR', '1') != '0'      # a unit's input-gradient and weight-gradient launches as one (A/B: 0)
_eval = os.environ.get('URSA_FUSED_EVAL', '1') != '0'    # evaluation forwards: one launch per bn -> relu -> conv unit (A/B: 0)


def eval_fused(flag=None):
    """Query / set whether evaluation-mode forwards of the BasicBlock networks take the fused units (`eval_trunk`)."""
    global _eval
    old = _eval
    if flag is not None:
        _eval = bool(flag)
    return old


def paired(flag=None):
    """Query / set whether a unit's two backward convolutions go out as ONE launch (ursa_preact_bwd_pair_f32)."""
    global _pair
    old = _pair
    if flag is not None:
        _pair = bool(flag)
    return old


_tls = threading.local()
_GROUP_SEPARATE = os.environ.get('URSA_GROUP_PAIR', '0') != '1'    # ChainGroup's chains keep the two backward launches apart (A/B: 1)


class separate_launches:
    """Context around a FORWARD pass: the units applied inside keep their input-gradient and weight-gradient launches apart,
    whatever `paired()` says (the decision is taken in the forward and travels to the backward in its autograd context).
    ChainGroup's chains run inside it. While the paired kernel held a CU alone that was the faster form there (8 chains per GPU:
    4.95 vs 4.79 samples/s paired); with the kernels capped at two waves per SIMD the pair is 1.5 % ahead in a group too (5.48 vs
    5.40, `profiles/r06_group_pair_ab.json`; `URSA_GROUP_PAIR=1` takes it) - the group keeps separate launches so that the paired
    kernel, the one `bench.py`'s roofline is quoted on, runs only alone on the device in the default command and its rocprofv3
    average (`profiles/r06_bench_kernel_stats.csv`) is not a mix of lone and eight-way concurrent executions."""

    def __enter__(self):
        self.old = getattr(_tls, 'separate', False)
        _tls.separate = True
        return self

    def __exit__(self, *exc):
        _tls.separate = self.old
        return False


class group_launches:
    """What ChainGroup wraps each chain's forward in: `separate_launches`, or nothing under URSA_GROUP_PAIR=1."""

    def __enter__(self):
        self.ctx = separate_launches() if _GROUP_SEPARATE else None
        if self.ctx is not None:
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def enabled(flag=None):
    """Query / set the process-wide switch; returns the previous value."""
    global _on
    old = _on
    if flag is not None:
        _on = bool(flag)
    return old


# module -> {key: zeroed uint8 scratch}. One PRIVATE persistent buffer per (layer, direction): the slots and counters of a K10
# launch must be zero when it starts and are zero again when it has drained; a layer's forward and backward never overlap
# (data dependence) and two layers never share. Weak keys: the scratch goes with its module.
_scratch = weakref.WeakKeyDictionary()


def _scratch_for(owner, key, geo, device):
    """geo: _native.preact_geometry()'s tuple (nl, scratch bytes, workgroups per channel, error word offset)."""
    d = _scratch.get(owner)
    if d is None:
        d = _scratch[owner] = {}
    ent = d.get(key)
    if ent is None or ent[0].device != device or ent[0].numel() < geo[1] or ent[1] != geo[3]:
        # (inside a capture with no eager warm-up the buffer belongs to the graph's pool and its zero fill is replayed: fine)
        ent = (torch.zeros(geo[1], dtype=torch.uint8, device=device), geo[3])
        if not torch.cuda.is_current_stream_capturing():
            d[key] = ent
    return ent[0]


def check(device=None):
    """Raises if a K10 launch's bounded poll ran out since the last check (its sums were NaN on the device already) or left
    its counters / slots non-zero. Never in a correct run. Reads every registered scratch (a device sync): for tests and
    debugging - the samplers do not call it."""
    bad = []
    for owner, d in list(_scratch.items()):
        for key, (t, err_off) in d.items():
            if device is not None and t.device != torch.device(device):
                continue
            if bool(t.any()):
                err = int(t[err_off:err_off + 4].view(torch.int32))
                bad.append((type(owner).__name__, key, 'poll ran out' if err else 'scratch not zero after the launch'))
                t.zero_()
    if bad:
        raise RuntimeError(f'K10 scratch: {bad}')


def _aligned(t):
    t = t.contiguous()
    return t.clone() if t.data_ptr() % 16 else t


def _wgrad(ctx_sink, K, x, bn_save, dy, w, ws_floats, stride):
    """Weight gradient by K7 (x operand rebuilt from bn_save while staged when given): into the engine's sink (deferred second
    launch) or returned."""
    def first():
        ws = x.new_empty(ws_floats)
        return (K.preact_wgrad_partial(x, bn_save, dy, w.shape, ws, stride) if bn_save is not None
                else K.conv_wgrad_partial(x, dy, w.shape, ws, stride))
    if ctx_sink is not None:
        ctx_sink.append((ctx_sink.launch(first, x, dy, bn_save), w))     # (beside the backward pass if the sink has a side stream)
        return None
    dw = torch.empty_like(w)
    K.conv_wgrad_reduce([(first(), dw)])
    return dw


class _Stem(torch.autograd.Function):
    """(y, partial) = conv1(x) with the statistics of y (URSABench/models/preresnet.py:100,138). x carries no gradient."""

    @staticmethod
    def forward(ctx, x, w, owner, ws_floats):
        ctx.set_materialize_grads(False)                        # (or autograd fills a float64 zero "gradient" for the partial sums)
        K = _native.default_kernels()
        geo = K.preact_geometry(x.shape, w.shape[0])
        y = x.new_empty(x.shape[0], w.shape[0], x.shape[2], x.shape[3])
        part = torch.empty(w.shape[0], geo[0], 2, dtype=torch.float64, device=x.device)
        K.preact_conv3x3(x, w, y, part, _scratch_for(owner, 'f', geo, x.device))
        ctx.save_for_backward(x, w)
        ctx.ws_floats, ctx.sink, ctx.weight = ws_floats, getattr(fused_conv._tls, 'sink', None), w
        ctx.mark_non_differentiable(part)
        return y, part

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _):
        x, w = ctx.saved_tensors
        if dy is None or not ctx.needs_input_grad[1]:
            return None, None, None, None
        dw = _wgrad(ctx.sink, _native.default_kernels(), x, None, _aligned(dy), ctx.weight, ctx.ws_floats, 1)
        return None, dw, None, None


class _Unit(torch.autograd.Function):
    """(y, partial_y, x) = (conv(relu(bn(x))) [+ shortcut], the statistics of y, x again). The third output is x itself, handed
    back so that whoever uses x on its other path (the block's shortcut) hangs off THIS node: the gradient of that path then
    arrives here and is added inside the dx launch instead of by an add launch of autograd's (fused_bn._AddBNReLUTrain's trick)."""

    @staticmethod
    def forward(ctx, x, px, gamma, beta, w, shortcut, bn, conv, stride, ws_floats):
        ctx.set_materialize_grads(False)
        K = _native.default_kernels()
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        dev = x.device
        geo = K.preact_geometry(x.shape, Cout, stride=stride, bn=True, add=shortcut is not None)
        y = x.new_empty(N, Cout, H // stride, W // stride)
        part = torch.empty(Cout, geo[0], 2, dtype=torch.float64, device=dev)
        save = x.new_empty(4, Cin)
        track = bn.training and bn.track_running_stats and bn.running_mean is not None
        if track and bn.num_batches_tracked is not None:        # None inside util.deferred_bn_counters
            bn.num_batches_tracked.add_(1)
        rm, rv = (bn.running_mean, bn.running_var) if track else (None, None)
        K.preact_conv3x3(x, w, y, part, _scratch_for(conv, 'f', geo, dev), stride=stride,
                         bn=(px, gamma, beta, rm, rv, save, bn.eps, bn.momentum if track else 0.0), add=shortcut)
        ctx.save_for_backward(x, gamma, w, save)
        ctx.stride, ctx.ws_floats, ctx.conv, ctx.weight = stride, ws_floats, conv, w
        ctx.sink = getattr(fused_conv._tls, 'sink', None)
        ctx.pair = _pair and not getattr(_tls, 'separate', False)
        ctx.has_shortcut = shortcut is not None
        ctx.mark_non_differentiable(part)
        return y, part, x

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, _, dxa):
        x, gamma, w, save = ctx.saved_tensors
        K = _native.default_kernels()
        s = ctx.stride
        dev = x.device
        if dy is None:                                          # the unit's result unused: only the shortcut path's gradient flows
            return dxa, None, None, None, None, None, None, None, None, None
        dy = _aligned(dy)
        Cin = x.shape[1]
        geo = K.preact_geometry(dy.shape, Cin, flip=True, stride=s)
        g = torch.empty_like(x)
        pb = torch.empty(Cin, geo[0], 2, dtype=torch.float64, device=dev)
        dw = None
        if ctx.needs_input_grad[4] and ctx.pair and (ctx.sink is None or ctx.sink.side is None):
            # the unit's two backward convolutions in ONE launch (their workgroups interleaved)
            rec = K.preact_bwd_pair(dy, w, g, x, save, pb, x.new_empty(ctx.ws_floats), s)
            if ctx.sink is not None:
                ctx.sink.append((rec, ctx.weight))
            else:
                dw = torch.empty_like(w)
                K.conv_wgrad_reduce([(rec, dw)])
        else:
            if ctx.needs_input_grad[4]:
                dw = _wgrad(ctx.sink, K, x, save, dy, ctx.weight, ctx.ws_floats, s)
            K.preact_conv3x3(dy, w, g, pb, None, stride=s, flip=True, bwd=(x, save))
        dx = torch.empty_like(x)
        dgb = x.new_empty(2, Cin)
        K.bn_bwd_dx(x, g, dx, gamma, save, pb, dgb[0], dgb[1], dz=None if dxa is None else _aligned(dxa))
        return dx, None, dgb[0], dgb[1], dw, (dy if ctx.has_shortcut else None), None, None, None, None


class _FinalBN(torch.autograd.Function):
    """relu(bn(z)) of the network's last BatchNorm (preresnet.py:146) from the partial sums the last unit left: K6's normalise
    launch alone forward, K6's two launches backward."""

    @staticmethod
    def forward(ctx, z, pz, gamma, beta, bn):
        K = _native.default_kernels()
        C = z.shape[1]
        y = torch.empty_like(z)
        save = z.new_empty(4, C)
        track = bn.training and bn.track_running_stats and bn.running_mean is not None
        if track and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        rm, rv = (bn.running_mean, bn.running_var) if track else (None, None)
        K.bn_apply(z, y, pz, gamma, beta, rm, rv, save, eps=bn.eps, momentum=bn.momentum if track else 0.0, relu=True)
        ctx.save_for_backward(z, gamma, beta, save)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        z, gamma, beta, save = ctx.saved_tensors
        K = _native.default_kernels()
        C = z.shape[1]
        dx = torch.empty_like(z)
        dgb = z.new_empty(2, C)
        K.bn_relu_backward(z, dy.contiguous(), dx, gamma, beta, save[0], save[1], dgb[0], dgb[1], z.new_empty(_native.bn_ws_floats(C)),
                           relu=True, gate=save[2:])
        return dx, None, dgb[0], dgb[1], None


_ones = {}


def one(device):
    """The persistent 1.0 the engine hands to `loss.backward()` for a loss that came from `trunk_loss`: `_Head.backward`
    recognises it by address and skips the multiplication by grad_output (and autograd's ones_like fill launch is gone too)."""
    device = torch.device(device)
    t = _ones.get(device)
    if t is None:
        t = _ones[device] = torch.ones((), device=device)
    return t


class _Head(torch.autograd.Function):
    """loss = cross_entropy(fc(avgpool(relu(bn(z)))), target) (URSABench/models/preresnet.py:146-150 + the samplers' CrossEntropyLoss)
    as K11's launches: BatchNorm + ReLU + pooling in one (the activation is not stored), the classifier, the loss AND their gradients
    in one workgroup, and in the backward one launch that turns the pooled gradient into dz / dgamma / dbeta."""

    @staticmethod
    def forward(ctx, z, pz, gamma, beta, W, b, target, bn, ignore_index):
        K = _native.default_kernels()
        N, C = z.shape[0], z.shape[1]
        save, pooled = z.new_empty(4, C), z.new_empty(N, C)
        track = bn.training and bn.track_running_stats and bn.running_mean is not None
        if track and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        rm, rv = (bn.running_mean, bn.running_var) if track else (None, None)
        K.bn_relu_pool(z, pz, gamma, beta, rm, rv, save, pooled, eps=bn.eps, momentum=bn.momentum if track else 0.0)
        loss = z.new_empty(1)
        dW, db, dp = torch.empty_like(W), (None if b is None else torch.empty_like(b)), torch.empty_like(pooled)
        K.fc_ce(pooled, W, b, target, loss, dW, db, dp, ignore_index=ignore_index)
        ctx.save_for_backward(z, gamma, save, dW, dp, *(() if db is None else (db,)))
        return loss.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, gl):
        z, gamma, save, dW, dp, *rest = ctx.saved_tensors
        db = rest[0] if rest else None
        K = _native.default_kernels()
        C = z.shape[1]
        if gl.data_ptr() != one(z.device).data_ptr():            # a caller's own grad_output: scale (three small launches)
            dW, dp = dW * gl, dp * gl
            db = None if db is None else db * gl
        dz, dgb = torch.empty_like(z), z.new_empty(2, C)
        K.bn_relu_pool_bwd(z, dp, gamma, save, dz, dgb[0], dgb[1])
        return dz, None, dgb[0], dgb[1], dW, db, None, None, None


def _bn_ok(bn, dev):
    return (type(bn) is nn.BatchNorm2d and bn.training and bn.affine and bn.momentum is not None and bn.weight.dtype == torch.float32
            and bn.weight.device == dev and bn.weight.is_contiguous() and bn.bias.is_contiguous())


def _conv_ok(conv, cin, cout, stride):
    return (isinstance(conv, fused_conv.Conv2d) and conv.bias is None and conv.kernel_size == (3, 3) and conv.stride == (stride, stride)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros'
            and conv.in_channels == cin and conv.out_channels == cout and conv.weight.dtype == torch.float32
            and conv.weight.is_contiguous() and conv.weight.data_ptr() % 16 == 0)


def _plan(model, x):
    """Per-shape decision, asked once: the K7 scratch sizes of every 3x3 layer in trunk order, or None when some layer is not
    covered. Cached on the module under the input's shape."""
    key = (tuple(x.shape), x.device)
    cached = model.__dict__.get('_ursa_block_plan')
    if cached is not None and cached[0] == key:
        return cached[1]
    K = _native.default_kernels()
    dev = x.device
    N, _, H, W = x.shape
    plan = None
    ok = (x.shape[1] == 3 and _conv_ok(model.conv1, 3, 16, 1) and K.preact_geometry(x.shape, 16) is not None and _bn_ok(model.bn, dev))
    if ok:
        ws = [K.conv_wgrad_ws_floats(x.shape, 16, 3, 1)]
        shape = (N, 16, H, W)
        for stage in (model.layer1, model.layer2, model.layer3):
            for blk in stage:
                st, planes = blk.conv1.stride[0], blk.conv1.out_channels
                cin = shape[1]
                mid = (N, planes, shape[2] // st, shape[3] // st)
                ok = (ok and _bn_ok(blk.bn1, dev) and _bn_ok(blk.bn2, dev) and _conv_ok(blk.conv1, cin, planes, st)
                      and _conv_ok(blk.conv2, planes, planes, 1) and (blk.downsample is not None) == (st != 1 or cin != planes)
                      and K.preact_geometry(shape, planes, stride=st, bn=True) is not None
                      and K.preact_geometry(mid, cin, flip=True, stride=st) is not None
                      and K.preact_geometry(mid, planes, bn=True, add=True) is not None
                      and K.preact_geometry(mid, planes, flip=True) is not None)
                if not ok:
                    break
                ws.append(K.conv_wgrad_ws_floats(shape, planes, 3, st))
                ws.append(K.conv_wgrad_ws_floats(mid, planes, 3, 1))
                shape = mid
            if not ok:
                break
        if ok and all(ws) and shape[1] == model.bn.num_features:
            plan = (ws, [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)])
    model.__dict__['_ursa_block_plan'] = (key, plan)
    return plan


def eligible(model, x):
    """Whether `trunk(model, x)` applies to this call (module docstring)."""
    if not (_on and fused_conv._on and fused_conv._k8 and fused_bn._on and fused_bn._probe is None and not fused_bn._two_launch):
        return False
    if not (torch.is_grad_enabled() and model.training and isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32
            and x.dim() == 4 and x.is_contiguous() and x.data_ptr() % 16 == 0 and not torch.is_autocast_enabled()):
        return False
    if not any(p.requires_grad for p in (model.conv1.weight, model.bn.weight)):
        return False
    plan = _plan(model, x)
    return plan is not None and all(b.training for b in plan[1])


def head_eligible(model, x, target, crit):
    """Whether `trunk_loss` applies: `trunk` does, the criterion is a plain mean nn.CrossEntropyLoss and K11 covers the head."""
    if not (type(crit) is nn.CrossEntropyLoss and crit.weight is None and crit.reduction == 'mean' and crit.label_smoothing == 0.0):
        return False
    fc = model.fc
    if not (type(fc) is nn.Linear and fc.weight.dtype == torch.float32 and fc.weight.is_contiguous()
            and (fc.bias is None or fc.bias.is_contiguous())):
        return False
    if not (isinstance(target, torch.Tensor) and target.is_cuda and target.dtype == torch.int64 and target.dim() == 1
            and target.is_contiguous() and target.shape[0] == x.shape[0]):
        return False
    if not eligible(model, x):
        return False
    return _native.default_kernels().head_supported((x.shape[0], model.bn.num_features, x.shape[2] // 4, x.shape[3] // 4), fc.out_features)


def _units(model, x):
    """conv1 -> layer1..3 as K10 launches: the last residual sum and the partial sums of its statistics."""
    ws = iter(_plan(model, x)[0])
    y, p = _Stem.apply(x, model.conv1.weight, model.conv1, next(ws))
    for stage in (model.layer1, model.layer2, model.layer3):
        for blk in stage:
            st = blk.conv1.stride[0]
            y1, p1, za = _Unit.apply(y, p, blk.bn1.weight, blk.bn1.bias, blk.conv1.weight, None, blk.bn1, blk.conv1, st, next(ws))
            sc = za if blk.downsample is None else blk.downsample(za)
            y, p, _ = _Unit.apply(y1, p1, blk.bn2.weight, blk.bn2.bias, blk.conv2.weight, sc, blk.bn2, blk.conv2, 1, next(ws))
    return y, p


def trunk_loss(model, x, target, crit):
    """The whole training forward as K10 + K11 launches: the scalar loss crit(model(x), target). Call only when
    `head_eligible(model, x, target, crit)`; hand `one(x.device)` to its `.backward()`."""
    y, p = _units(model, x)
    return _Head.apply(y, p, model.bn.weight, model.bn.bias, model.fc.weight, model.fc.bias, target, model.bn, crit.ignore_index)


def _eval_plan(model, x):
    key = ('eval', tuple(x.shape), x.device)
    cached = model.__dict__.get('_ursa_block_eval_plan')
    if cached is not None and cached[0] == key:
        return cached[1]
    K = _native.default_kernels()
    N, _, H, W = x.shape
    ok = x.shape[1] == 3 and _conv_ok(model.conv1, 3, 16, 1) and K.conv3x3_supported(x.shape, 16)
    shape = (N, 16, H, W)
    if ok:
        for stage in (model.layer1, model.layer2, model.layer3):
            for blk in stage:
                st, planes = blk.conv1.stride[0], blk.conv1.out_channels
                cin = shape[1]
                mid = (N, planes, shape[2] // st, shape[3] // st)
                ok = (ok and _conv_ok(blk.conv1, cin, planes, st) and _conv_ok(blk.conv2, planes, planes, 1)
                      and (blk.downsample is not None) == (st != 1 or cin != planes)
                      and (blk.downsample is None or K.conv1x1s2_supported(shape, planes))
                      and K.preact_eval_supported(shape, planes, stride=st) and K.preact_eval_supported(mid, planes, add=True))
                shape = mid
            if not ok:
                break
    plan = bool(ok and shape[1] == model.bn.num_features)
    model.__dict__['_ursa_block_eval_plan'] = (key, plan)
    return plan


def eval_eligible(model, x):
    """Whether `eval_trunk(model, x)` applies: an evaluation-mode forward (running statistics everywhere) with no gradient recorded,
    contiguous fp32 NCHW on a HIP device, every layer covered. An ensemble member's forward in the BMA predictive."""
    if not (_on and _eval and fused_conv._on and fused_bn._on):
        return False
    if not ((not model.training) and isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous() and x.data_ptr() % 16 == 0 and not torch.is_autocast_enabled()):
        return False
    if torch.is_grad_enabled() and (x.requires_grad or model.conv1.weight.requires_grad):
        return False
    bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    if any(b.training or b.running_mean is None or not b.affine or type(b) is not nn.BatchNorm2d for b in bns):
        return False
    return _eval_plan(model, x)


def eval_trunk(model, x):
    """conv1 -> layer1..3 of a BasicBlock PreResNet in evaluation mode: one launch per bn -> relu -> conv unit (the BatchNorm's
    running-statistics transform applied while the convolution stages its tile, `out += residual` in the epilogue) instead of
    K6's evaluation launch + a convolution launch each; returns the last residual sum (the caller applies the final bn / relu)."""
    K = _native.default_kernels()
    y = K.conv3x3(x, model.conv1.weight)
    for stage in (model.layer1, model.layer2, model.layer3):
        for blk in stage:
            st = blk.conv1.stride[0]
            N, _, H, W = y.shape
            planes = blk.conv1.out_channels
            b1, b2 = blk.bn1, blk.bn2
            y1 = K.preact_eval(y, blk.conv1.weight, y.new_empty(N, planes, H // st, W // st), b1.weight, b1.bias, b1.running_mean, b1.running_var,
                               eps=b1.eps, stride=st)
            sc = y if blk.downsample is None else K.conv1x1s2(y, blk.downsample[0].weight)
            y = K.preact_eval(y1, blk.conv2.weight, torch.empty_like(y1), b2.weight, b2.bias, b2.running_mean, b2.running_var, eps=b2.eps, add=sc)
    return y


def trunk(model, x):
    """conv1 -> layer1..3 -> bn -> relu of a BasicBlock PreResNet (URSABench/models/preresnet.py:138-146) as K10 launches; returns
    relu(bn(z)) of the last residual sum. Call only when `eligible(model, x)`."""
    y, p = _units(model, x)
    return _FinalBN.apply(y, p, model.bn.weight, model.bn.bias, model.bn)
