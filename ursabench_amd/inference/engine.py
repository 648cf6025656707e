"""The chain engine: one minibatch step = forward + loss + backward (stock PyTorch-ROCm:
MIOpen / rocBLAS) + ONE fused gfx950 update launch, optionally captured once into a hipGraph
and replayed for every full-size batch of the epoch.

Replaces the hot loop of URSABench/inference/sghmc.py:72-86 (and csghmc.py:80-93). What is
removed relative to the reference: 8 launches per parameter tensor per step (488 for
PreResNet-20), `optimizer.zero_grad()` (fused into the update), the per-step host sync
`loss.item()` (sghmc.py:82; the loss is accumulated on the device) and every host-side launch
gap (graph replay). Per-step scalars (lr, noise scale, noise on/off, Philox counter) live in a
device control block so the captured graph stays valid while they change; the update
launch advances the block itself.

Injected noise (parity runs: `eps_per_step`) goes through ONE persistent device buffer that the
captured update launch reads: the host refills it before every replay, so the replayed graph — the
path that is timed — is the path the reference comparisons execute.
"""
import torch

from .. import fused_block, fused_bn, fused_conv
from .._capture import capture, side_streams
from ..util import deferred_bn_counters


class ChainEngine:
    WARMUP_STEPS = 3

    def __init__(self, model, optimizer, loss_criterion, device, use_graph=None):
        self.model, self.opt, self.crit = model, optimizer, loss_criterion
        self.device = torch.device(device)
        if use_graph is None:
            use_graph = self.device.type == 'cuda'
        if use_graph and self.device.type != 'cuda':
            raise ValueError('hipGraph capture needs a HIP device')
        self.use_graph = use_graph
        self.loss_acc = torch.zeros((), device=self.device)     # sum_i loss_i * batch_i, on device
        self._params = list(optimizer.arena.params)
        self._slot = {id(p): i for i, p in enumerate(self._params)}     # parameter -> its arena view
        optimizer.ctl_zero_grad = False                         # the packed copy overwrites every gradient
        self._graph = None
        self._graph_eps = False          # whether the captured update launch reads the injected-noise buffer
        self._static = None
        self._eps_static = None
        self._eager_full_steps = 0
        # K7's first launches on a side stream beside the backward pass (fused_conv.Sink). OFF: measured SLOWER on this stack - 21
        # fork / join edges per captured step cost more than the overlap gains (2.90 -> 2.25 samples/s, 223 us of idle time per
        # step between dependent branches; profiles/r06_side_branch_ab.json). The pairing that works is inside ONE launch
        # (ursa_preact_bwd_pair_f32). Kept as a switch: it changes no bit (tests/test_fused_block_gpu.py).
        self.wgrad_side = False
        self.fused_head = True           # take model.forward_loss (K11) where it applies
        self.gate_probe = None           # fused_bn.GateProbe: parity runs against the reference CPU path only
        self._graph_probe = None
        self.stats = dict(graph_replays=0, eager_steps=0, captures=0)

    def invalidate(self):
        self._graph, self._static, self._eager_full_steps = None, None, 0

    def set_optimizer(self, optimizer):
        """update_hyp rebuilt the optimizer over the same arena: drop the captured graph."""
        self.opt = optimizer
        optimizer.ctl_zero_grad = False
        self.invalidate()

    def forward_backward(self, x, y, wgrad_side=None):
        """Forward + loss + backward, gradients packed into the arena. Returns what `finish` needs to put back
        the tensors that received no gradient (None almost always). wgrad_side=False: keep K7's launches on the current
        stream (ChainGroup: the device is already shared by the chains' branches)."""
        side = None
        if self.wgrad_side and wgrad_side is not False:
            side = side_streams(self.device, 2)[1]          # ([0]: the stream eager warm-up steps / single-chain groups run on)
        with deferred_bn_counters(self.model), fused_bn.probing(self.gate_probe), fused_conv.deferred(side) as pend:
            # (deferred_bn_counters: 19 one-element counter kernels -> one multi-tensor add)
            fl = getattr(self.model, 'forward_loss', None)
            loss = fl(x, y, self.crit) if fl is not None and self.fused_head else None      # K10 + K11: the loss without the logits
            fused_loss = loss is not None
            if not fused_loss:
                loss = self.crit(self.model(x), y)
        # Gradients: with p.grad = None autograd hands over its freshly computed tensors (no kernel);
        # ONE multi-tensor copy then packs them into the flat arena. Leaving the arena views in
        # p.grad instead makes autograd run one `grad += new` kernel per parameter tensor (61 launches
        # and a fused re-zeroing for PreResNet-20: +0.11 ms per step, tools/exp/step_variants.py).
        for p in self._params:
            p.grad = None
        if fused_loss:
            loss.backward(fused_block.one(self.device))     # a persistent 1.0: no ones_like fill, no scaling in the head's backward
        else:
            loss.backward()
        pend.join()                                 # the K7 branch meets the main stream again: its partial sums are reduced below
        grads = [p.grad for p in self._params]
        grad_views = self.opt.arena.grad_views
        keep = None
        if pend:
            # K7: the convolutions' weight gradients are still K-sliced partial sums; ONE launch reduces all of them, in a fixed
            # order, straight into their arena slots (no gradient tensor, no copy for them)
            slot = self._slot
            stray = [tuple(w.shape) for _, w in pend if id(w) not in slot]
            if stray:                                   # (re-registered after the engine was built, or left out of the optimizer)
                raise RuntimeError(f'convolution weights {stray} took K7\'s deferred launch but are not in the optimizer\'s arena: build the '
                                   f'engine / optimizer after the model\'s parameters are final')
            fused_conv.flush(pend, lambda w: grad_views[slot[id(w)]])
            done = {slot[id(w)] for _, w in pend}
            for i in done:
                if grads[i] is not None:            # the same weight also went through a call K7 does not cover
                    grad_views[i].add_(grads[i])
            rest = [i for i in range(len(grads)) if i not in done]
            if any(grads[i] is None for i in rest):
                pairs = [(grad_views[i], grads[i]) for i in rest if grads[i] is not None]
                if pairs:
                    torch._foreach_copy_([v for v, _ in pairs], [g for _, g in pairs])
                if getattr(self.opt, 'skip_grad_none', True):
                    keep = self.opt.arena.stash([i for i in rest if grads[i] is None])
            elif rest:
                torch._foreach_copy_([grad_views[i] for i in rest], [grads[i] for i in rest])
        elif any(g is None for g in grads):
            # a parameter that received no gradient (frozen / unused layer) is skipped by the reference
            # altogether (optim_sghmc.py:44-45): no prior pull, no noise. The flat launch covers it, so its
            # theta / momentum slices are copied aside and put back (device-to-device, part of the capture).
            pairs = [(v, g) for v, g in zip(grad_views, grads) if g is not None]
            torch._foreach_copy_([v for v, _ in pairs], [g for _, g in pairs])
            if getattr(self.opt, 'skip_grad_none', True):
                keep = self.opt.arena.stash([i for i, g in enumerate(grads) if g is None])
        else:
            torch._foreach_copy_(grad_views, grads)
        for p in self._params:
            p.grad = None
        self.loss_acc.add_(loss.detach(), alpha=x.shape[0])
        return keep

    def finish(self, keep):
        if keep is not None:
            self.opt.arena.unstash(keep)

    def _train_step(self, x, y, eps=None):
        keep = self.forward_backward(x, y)
        self.opt.ctl_step(eps=eps)
        self.finish(keep)

    def _capture(self, x, y):
        self._static = (torch.empty_like(x), torch.empty_like(y))
        self._static[0].copy_(x)
        self._static[1].copy_(y)
        g = torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog thread (one process per GPU jobs) may touch the HIP runtime while
        # this thread captures; only this thread's calls belong to the capture
        with capture(g):
            self._train_step(*self._static, eps=self._eps_static if self._graph_eps else None)
        self._graph = g
        self._graph_probe = self.gate_probe
        self.stats['captures'] += 1

    def run_epoch(self, loader, add_langevin_noise, sched=None, eps_per_step=None, gates_per_step=None):
        """One pass over `loader`. Returns the number of examples seen; self.loss_acc holds the
        summed loss (read it with .item() only when debugging: that is the one host sync).
        gates_per_step(k) -> the reference run's near-zero gate lists of minibatch step k for `self.gate_probe`
        (parity runs; loaded into the probe's persistent buffers before the step, observations collected after it)."""
        self.model.train()
        self.loss_acc.zero_()
        self.opt.ctl_begin(add_langevin_noise, sched)
        full = getattr(loader, 'batch_size', None)
        seen = steps = 0
        inject = eps_per_step is not None
        if inject and self._eps_static is None:
            self._eps_static = torch.zeros_like(self.opt.arena.theta)
        if self._graph is not None and (self._graph_eps != inject or self._graph_probe is not self.gate_probe):
            self._graph = None                       # the captured launches read / do not read the noise / gate buffers
        self._graph_eps = inject
        if gates_per_step is not None and self.gate_probe is None:
            raise ValueError('gates_per_step needs engine.gate_probe')
        eps_buf = self._eps_static if inject else None
        for bi, (x, y) in enumerate(loader):
            x = x.to(self.device, non_blocking=True)
            y = y.to(self.device, non_blocking=True)
            b = x.shape[0]
            if inject:
                self._eps_static.copy_(eps_per_step(steps))
            if gates_per_step is not None:
                self.gate_probe.load(gates_per_step(steps))
            if self.use_graph and b == full:
                if self._graph is None and self._eager_full_steps >= self.WARMUP_STEPS:
                    self._capture(x, y)            # records the step; the replay below executes it
                if self._graph is not None:
                    if self._static[0].shape != x.shape:
                        raise RuntimeError(f'batch shape changed {tuple(self._static[0].shape)} -> {tuple(x.shape)}')
                    self._static[0].copy_(x)
                    self._static[1].copy_(y)
                    self._graph.replay()
                    self.stats['graph_replays'] += 1
                else:
                    # warm-up steps are real steps, run on a side stream as capture will be
                    s = side_streams(self.device, 1)[0]
                    s.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(s):
                        self._train_step(x, y, eps_buf)
                    torch.cuda.current_stream(self.device).wait_stream(s)
                    self._eager_full_steps += 1
                    self.stats['eager_steps'] += 1
            else:
                self._train_step(x, y, eps_buf)
                self.stats['eager_steps'] += 1
            if gates_per_step is not None:
                self.gate_probe.collect()
            seen += b
            steps += 1
        self.opt.ctl_end(steps)
        if fused_bn.held_in_use():      # K6's held form was opted into (off by default): one small read per epoch; a starved
            fused_bn.check_held(self.device)   # launch - NaN outputs already - raises here instead of at the next predictive
        return seen
