"""Host helpers the hot path needs, with the semantics of URSABench/util.py (file:line cited
per function). Everything here is plumbing; the arithmetic lives in the HIP kernels."""
import random
import time

import numpy as np
import torch
from torch.nn import CrossEntropyLoss
from torch.nn.modules.batchnorm import _BatchNorm

from . import fused_bn

_random_seed = None


def set_random_seed(seed=None):
    """util.py:20-29 — seeds python, numpy, torch (and the HIP generator). The samplers read
    torch.initial_seed() at construction as the Philox key of their chain."""
    global _random_seed
    if seed is None:
        seed = int((time.time() * 1e6) % 1e8)
    _random_seed = seed
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def silent(fn):
    """util.py:40-50 — run fn with stdout discarded (the timing harness silences the samplers' prints)."""
    import contextlib
    import io

    def silent_fn(*args, **kwargs):
        with contextlib.redirect_stdout(io.StringIO()):
            return fn(*args, **kwargs)
    return silent_fn


def get_loss_criterion(loss='multi_class_linear_output', **kwargs):
    """util.py:80-89 — mean-reduced cross entropy is the only supported likelihood."""
    if loss != 'multi_class_linear_output':
        raise NotImplementedError
    return CrossEntropyLoss(**kwargs)


def reset_model(model):
    """util.py:92-107 — reset_parameters() on TOP-LEVEL children only (nested blocks are not
    re-initialised: a reference quirk callers rely on for timing, kept). In place, so arena
    views stay valid."""
    if not isinstance(model, torch.nn.Module):
        raise NotImplementedError
    for _, child in model.named_children():
        fn = getattr(child, 'reset_parameters', None)
        if fn is not None:
            fn()
    return model


def flatten(lst):
    """util.py:163-169."""
    return torch.cat([t.contiguous().view(-1, 1) for t in lst]).view(-1)


def set_weights(model, vector, device=None):
    """util.py:172-176."""
    off = 0
    for p in model.parameters():
        n = p.numel()
        p.data.copy_(vector[off:off + n].view(p.size()).to(device))
        off += n


def adjust_learning_rate(optimizer, lr):
    """util.py:179-182."""
    for g in optimizer.param_groups:
        g['lr'] = lr
    return lr


def central_smoothing(proba, gamma=1e-4):
    """util.py:126-134 (host version, used only by the metric surface on [N, C] means)."""
    return (1 - gamma) * proba + gamma * 1 / (proba.shape[1])


def compute_predictive_entropy(proba):
    """util.py:137-144."""
    return -(proba * torch.log(proba)).sum(dim=-1)


class deferred_bn_counters:
    """Train-mode BatchNorm bumps `num_batches_tracked` with its own tiny kernel per layer per forward (19 launches
    of ~4 us in a PreResNet-20 step: 3 % of it, `CUDAFunctorOnSelf_add<long>` in profiles/r02_bench_kernel_stats.csv).
    Inside this context the layers see no counter (nn.BatchNorm skips the increment when the buffer is None); on exit
    the counters are put back and bumped by `forwards` with ONE multi-tensor add: same integers, one launch.
    Layers with momentum=None (cumulative average: the counter feeds the update) are left alone."""

    def __init__(self, model, forwards=1):
        self.mods = [m for m in model.modules() if isinstance(m, _BatchNorm) and m.momentum is not None
                     and m.track_running_stats and m._buffers.get('num_batches_tracked') is not None]
        self.forwards = forwards

    def __enter__(self):
        self.counters = [m._buffers['num_batches_tracked'] for m in self.mods]
        for m in self.mods:
            m._buffers['num_batches_tracked'] = None
        return self

    def __exit__(self, *exc):
        for m, c in zip(self.mods, self.counters):
            m._buffers['num_batches_tracked'] = c
        if self.counters and exc[0] is None and self.forwards:
            torch._foreach_add_(self.counters, self.forwards)
        return False


def check_bn(model):
    """util.py:185-193."""
    return any(isinstance(m, _BatchNorm) for m in model.modules())


def bn_update(loader, model, subset=None, device=None, **kwargs):
    """util.py:212-247 — reset BN statistics and re-estimate them with one train-mode pass using
    the cumulative-average momentum b/(n+b). The reference hard-codes input.cuda() (:236); here
    the batch goes to the model's own device. Statistics are reset IN PLACE (the reference
    re-assigns the buffers, :198-199) so member-bank views stay valid."""
    bn_update_many(loader, [model], subset=subset, device=device, **kwargs)


def bn_update_many(loader, models, subset=None, device=None, streams=None, **kwargs):
    """bn_update for several independent models in ONE pass over the loader: every batch is pushed through each
    model, each on its own stream when `streams` is given (the forwards of different SWAG members share nothing, and
    a 128-row WideResNet-28-10 forward leaves a quarter of the GPU idle). Per model the operations, their order and
    their arguments are exactly bn_update's, so the statistics are the same bits."""
    groups = [[m for m in model.modules() if isinstance(m, _BatchNorm)] for model in models]
    if not any(groups):
        return
    if device is None:
        device = next(models[0].parameters()).device
    was_training = [model.training for model in models]
    momenta = {}
    for model, bns in zip(models, groups):
        model.train()
        for m in bns:
            m.running_mean.zero_()
            m.running_var.fill_(1)
            momenta[m] = m.momentum
    n = 0
    num_batches = len(loader)
    counters = [deferred_bn_counters(model, forwards=0) for model in models]   # bumped once, by the forwards done
    cur = torch.cuda.current_stream(device) if streams else None
    with torch.no_grad():
        for c in counters:
            c.__enter__()
        try:
            if subset is not None:
                num_batches = int(num_batches * subset)
            for bi, (x, _) in enumerate(iter(loader)):
                if subset is not None and bi >= num_batches:
                    break
                x = x.to(device, non_blocking=True)
                b = x.size(0)
                mom = b / (n + b)
                for k, (model, bns) in enumerate(zip(models, groups)):
                    for m in bns:
                        m.momentum = mom
                    if streams:
                        streams[k].wait_stream(cur)
                        # the batch was allocated on `cur` (a shuffled gather, or the H2D copy of a host loader) and is
                        # read on the side stream: tell the caching allocator, or the block is handed to a later
                        # gather / copy on `cur` while this forward may still be reading it
                        x.record_stream(streams[k])
                        # (the members' forwards overlap on the device: K6's held form is for one launch at a time)
                        with torch.cuda.stream(streams[k]), fused_bn.several_streams():
                            model(x, **kwargs)
                    else:
                        model(x, **kwargs)
                n += b
                for c in counters:
                    c.forwards += 1
            if streams:
                for st in streams[:len(models)]:
                    cur.wait_stream(st)
        finally:
            for c in counters:
                c.__exit__(None, None, None)
    for model, bns, tr in zip(models, groups, was_training):
        for m in bns:
            m.momentum = momenta[m]
        model.train(tr)
    if fused_bn.held_in_use():          # K6's held form was opted into: a starved launch (NaN statistics) raises here, not later
        fused_bn.check_held(device)
