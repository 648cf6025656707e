"""optimSGHMC — drop-in for URSABench/inference/optim_sghmc.py:7-68 over a flat HBM arena.

Same constructor, same `param_groups` keys, same `step(add_langevin_noise=True, closure=None)`,
same lazily created `state[p]['momentum_buffer']`; it stays a real torch.optim.Optimizer so
CosineAnnealingLR accepts it (sghmc.py:44). What changes is the execution: the parameters of a
group are re-homed into one flat fp32 buffer (FlatArena) and the whole update is ONE launch of
the gfx950 kernel `ursa_sgmcmc_step_f32` instead of 8 tiny torch ops per tensor.

Noise: the reference draws torch.randn_like per tensor from torch's global generator (:64).
Here it is Philox4x32-10 keyed by `seed` (default torch.initial_seed(), i.e. what
util.set_random_seed set) and the step counter; pass `eps=` (flat, arena layout) to inject a
captured noise vector instead — that is how parity with the reference is asserted.
"""
import ctypes
import math

import torch
from torch.optim.optimizer import Optimizer, required

from .. import _native
from ..arena import FlatArena

_GROUP_SALT = 0x9E3779B97F4A7C15


class optimSGHMC(Optimizer):

    def __init__(self, params, lr=required, momentum=0, dampening=0, weight_decay=0, num_training_samples=None,
                 nesterov=False, *, kernels=None, seed=None, module=None, arena=None, fuse_zero_grad=False):
        if lr is not required and lr < 0.0:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if momentum < 0.0:
            raise ValueError("Invalid momentum value: {}".format(momentum))
        if weight_decay < 0.0:
            raise ValueError("Invalid weight_decay value: {}".format(weight_decay))
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov,
                        num_training_samples=num_training_samples)
        super().__init__(params, defaults)
        self.kernels = kernels if kernels is not None else _native.default_kernels()
        self.seed = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        self.fuse_zero_grad = fuse_zero_grad
        if arena is not None:
            if len(self.param_groups) != 1:
                raise ValueError('arena= can only be shared with a single parameter group')
            self.arenas = [arena]
            arena.rebind()
        else:
            single = len(self.param_groups) == 1
            self.arenas = [FlatArena(g['params'], module=module if single else None) for g in self.param_groups]
        self._has_mom = [False] * len(self.param_groups)
        self._step = 0                 # Philox call index == number of updates applied so far
        self._ctl = None               # device control block (graph-replayable stepping)
        self._sched = None             # device (lr, c_noise) table for per-iteration schedules
        self.self_advance = True       # control-block stepping: the update launch advances the block (False: explicit advance launch)
        self.ctl_zero_grad = True      # control-block stepping re-zeroes the flat gradient (fused)
        self.skip_grad_none = True     # reference semantics for tensors whose .grad is None (optim_sghmc.py:44-45)

    def __setstate__(self, state):
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault('nesterov', False)

    @property
    def arena(self):
        return self.arenas[0]

    def adopt_device_state(self, old):
        """This optimizer replaces `old` over the same arena (update_hyp): take over its device control
        block and schedule table — a hipGraph captured with `old` holds their addresses, and dropping
        them would hand those addresses back to the caching allocator (use-after-free on replay) — and
        continue its Philox call counter (same key: a restart at 0 would replay the same noise)."""
        if old.arena is not self.arena:
            raise ValueError('adopt_device_state: the two optimizers do not share an arena')
        self._ctl, self._sched = old._ctl, old._sched
        old._ctl = old._sched = None
        if old.seed == self.seed:
            self._step = old._step

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad's contract, default included. set_to_none=True (what the reference's bare
        `optimizer.zero_grad()` means, sghmc.py:79): every .grad becomes None, so after backward() a tensor that received
        no gradient is still None and step() skips it exactly like optim_sghmc.py:44-45 (no prior pull, no noise);
        step() packs the fresh gradients into the arena with one multi-tensor copy. set_to_none=False: one memset of the
        flat gradient buffer, the .grad views stay bound and autograd accumulates into the arena directly (fastest
        drop-in loop, but an unused tensor then has a zero gradient, not None, and is updated by prior and noise)."""
        for a in self.arenas:
            if set_to_none:
                for p in a.params:
                    p.grad = None
            else:
                a.rebind()
                a.grad.zero_()

    def _register_momentum_views(self, gi):
        a = self.arenas[gi]
        for p, v in zip(a.params, a.layout.views(a.mom)):
            self.state[p]['momentum_buffer'] = v
        self._has_mom[gi] = True

    def _scalars(self, group, add_langevin_noise, gi):
        mu, lr, wd, n = group['momentum'], group['lr'], group['weight_decay'], group['num_training_samples']
        if group['nesterov']:
            raise NotImplementedError('nesterov is accepted but unused by every URSABench sampler; not implemented')
        if (wd != 0 or add_langevin_noise) and n is None:
            raise TypeError('num_training_samples is required (optim_sghmc.py:48,64 divide by it)')
        flags = 0
        if add_langevin_noise:
            flags |= _native.STEP_NOISE
        if wd != 0:
            flags |= _native.STEP_WD
        if mu != 0 and not self._has_mom[gi]:
            flags |= _native.STEP_FIRST
        if self.fuse_zero_grad:
            flags |= _native.STEP_ZERO_GRAD
        return dict(lr=float(lr), mu=float(mu), c_wd=(wd / n) if wd != 0 else 0.0,
                    c_noise=math.sqrt(2 * (1 - mu) * lr), n_train=float(n) if n is not None else 1.0, flags=flags)

    @torch.no_grad()
    def step(self, add_langevin_noise=True, closure=None, *, eps=None, snapshot=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            a = self.arenas[gi]
            # tensors without a gradient are skipped by the reference (optim_sghmc.py:44-45): the flat launch
            # covers them, so their theta / momentum slices are put back afterwards (rare: frozen layers)
            skipped = [i for i, p in enumerate(a.params) if p.grad is None] if self.skip_grad_none else []
            a.rebind()
            sc = self._scalars(group, add_langevin_noise, gi)
            mom = a.ensure_mom() if sc['mu'] != 0 else None
            keep = a.stash(skipped) if skipped else None
            seed = (self.seed ^ (gi * _GROUP_SALT)) & 0xFFFFFFFFFFFFFFFF
            self.kernels.sgmcmc_step(a.theta, a.grad, mom, seed=seed, step=self._step,
                                     eps=eps if gi == 0 else None, snapshot=snapshot if gi == 0 else None, **sc)
            if keep is not None:
                a.unstash(keep, snapshot if gi == 0 else None)
            if sc['mu'] != 0 and not self._has_mom[gi]:
                self._register_momentum_views(gi)
        self._step += 1
        return loss

    # ---- device-control-block stepping (used by the samplers; hipGraph-replayable) -----------
    def set_ctl_storage(self, view):
        """Keep this optimizer's control block in caller-provided device storage (one 64-byte slot of the block
        array a ChainGroup hands to the multi-chain launch). The current contents move along."""
        if view.numel() != _native.CTL_BYTES or view.dtype != torch.uint8:
            raise ValueError('control-block storage must be sizeof(ursa_step_ctl) uint8 elements')
        if self._ctl is not None:
            view.copy_(self._ctl)
        self._ctl = view

    def rehomed(self):
        """The arena moved (FlatArena.rehome): re-point the lazily created momentum_buffer state entries."""
        for gi in range(len(self.param_groups)):
            if self._has_mom[gi]:
                self._register_momentum_views(gi)

    def ctl_begin(self, add_langevin_noise, sched=None):
        """Upload this epoch's scalars. sched: optional float32 [steps, 2] of (lr, c_noise) per step."""
        if len(self.param_groups) != 1:
            raise NotImplementedError('graph stepping supports one parameter group')
        a, group = self.arena, self.param_groups[0]
        sc = self._scalars(group, add_langevin_noise, 0)
        a.ensure_mom()
        if sched is None:
            if self._sched is not None:
                raise ValueError('this optimizer was stepped with a schedule table before; keep passing one')
        else:
            # persistent buffer: the control block (and so a captured graph) holds its address and length
            if self._sched is None:
                self._sched = torch.zeros(sched.shape[0], 2, device=a.device)
            if self._sched.shape != sched.shape:
                raise ValueError(f'schedule table changed shape {tuple(self._sched.shape)} -> {tuple(sched.shape)}')
            self._sched.copy_(sched)
        # the update launch advances the block itself (two-level ticket tree on separate cache lines): no second launch
        self.self_advance = True
        c = _native.StepCtl(lr=sc['lr'], mu=sc['mu'], c_wd=sc['c_wd'], c_noise=sc['c_noise'], n_train=sc['n_train'],
                            flags=sc['flags'] | (_native.STEP_ADVANCE if self.self_advance else 0)
                            | (_native.STEP_ZERO_GRAD if self.ctl_zero_grad else 0),
                            seed=self.seed, step=self._step, sched_base=self._step,
                            sched=0 if self._sched is None else self._sched.data_ptr(),
                            sched_len=0 if self._sched is None else self._sched.shape[0])
        if sched is not None:
            c.lr = float(sched[0, 0])
            if sc['flags'] & _native.STEP_SGD:
                c.mu = float(sched[0, 1])         # SGD mode: the table's second column is the momentum
            else:
                c.c_noise = float(sched[0, 1])
        host = torch.frombuffer(bytearray(bytes(c)), dtype=torch.uint8)
        if self._ctl is None:
            self._ctl = torch.zeros(_native.CTL_BYTES, dtype=torch.uint8, device=a.device)
        self._ctl.copy_(host)
        self._ctl_mu = c.mu

    @torch.no_grad()
    def ctl_step(self, eps=None):
        """The update and the advance of the control block: ONE launch."""
        a = self.arena
        self.kernels.sgmcmc_step_ctl(a.theta, a.grad, a.mom, self._ctl, eps=eps)
        if not self.self_advance:
            self.kernels.step_ctl_advance(self._ctl)

    def ctl_end(self, n_steps):
        self._step += n_steps
        if n_steps:
            self._opt_called = True      # LRScheduler's "scheduler.step() before optimizer.step()" check
        if n_steps and self._ctl_mu != 0 and not self._has_mom[0]:
            self._register_momentum_views(0)
