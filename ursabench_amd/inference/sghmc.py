"""SGHMC / SGLD samplers — drop-ins for URSABench/inference/sghmc.py:14-115 and sgld.py:8-35.

State machine, hyper-parameter keys, scheduler quirks and return type (`list[nn.Module]`) are
the reference's; the hot loop runs on the ChainEngine (fused gfx950 update, hipGraph replay,
no per-step host sync) and a posterior sample is a device-to-device snapshot into a member
bank instead of deepcopy(model.cpu()) (sghmc.py:99).

Reference quirks kept on purpose: `burnt_in` is set True before the loop
(sghmc.py:67-69) so Langevin noise is on for every step including burn-in; the constructor's
CosineAnnealingLR has eta_min=0 while update_hyp's has eta_min=lr/2 (:44-45 vs :62-63).
"""
import torch
from torch.optim.lr_scheduler import CosineAnnealingLR

from ..arena import MemberBank
from ..util import get_loss_criterion, reset_model
from .engine import ChainEngine
from .inference_base import _Inference
from .optim_sghmc import optimSGHMC

try:                                   # optional, like the reference's wandb_debug flag
    import wandb
except Exception:                      # pragma: no cover
    wandb = None


class _ChainSampler(_Inference):
    """What SGHMC/SGLD/cSGHMC/cSGLD share: arena-backed optimizer, engine, member bank."""

    def _setup_chain(self, model, train_loader, model_loss, device, kernels, use_graph, lr, momentum, weight_decay,
                     seed=None):
        """seed: the chain's Philox key. None = torch.initial_seed() at construction (what
        util.set_random_seed set, experiment.py:170) — build chains that must be independent with distinct
        seed= values or a set_random_seed call between them; ChainGroup refuses equal keys."""
        if not isinstance(model, torch.nn.Module):
            raise NotImplementedError
        self.seed = int(torch.initial_seed() if seed is None else seed)
        self.model_loss = model_loss
        self.model = model.to(device)
        self.train_loader = train_loader
        self.device = device
        self.dataset_size = len(train_loader.dataset)
        self._kernels = kernels
        self.optimizer = optimSGHMC(params=self.model.parameters(), lr=lr, momentum=momentum,
                                    num_training_samples=self.dataset_size, weight_decay=weight_decay,
                                    kernels=kernels, module=self.model, seed=self.seed)
        self.arena = self.optimizer.arena
        self.loss_criterion = get_loss_criterion(loss=model_loss)
        self.engine = ChainEngine(self.model, self.optimizer, self.loss_criterion, device, use_graph)
        self.bank = MemberBank(self.arena)
        self.eps_provider = None        # callable(step_in_epoch) -> flat eps tensor: parity runs only
        self.gate_provider = None       # callable(step) -> the reference run's near-zero ReLU gate lists (with
        #                                 engine.gate_probe = fused_bn.GateProbe(...)): parity runs only

    def _new_optimizer(self, lr, momentum, weight_decay):
        """update_hyp rebuilds the optimizer (sghmc.py:57-58); the arena, the device control block and
        the schedule table (so every address a captured graph may hold) are reused, the momentum buffer
        restarts from the first-step rule. The chain keeps its Philox key and the call counter carries
        over, so a hyper-optimisation trial never replays the noise of the previous one."""
        old = self.optimizer
        self.optimizer = optimSGHMC(params=self.model.parameters(), lr=lr, momentum=momentum,
                                    num_training_samples=self.dataset_size, weight_decay=weight_decay,
                                    kernels=self._kernels, arena=self.arena, seed=self.seed)
        self.optimizer.adopt_device_state(old)
        self.engine.set_optimizer(self.optimizer)

    def _snapshot(self):
        return self.bank.snapshot(self.model)

    def _eps_for_epoch(self):
        if self.eps_provider is None:
            return None
        base = self.optimizer._step
        return lambda k: self.eps_provider(base + k)

    def _gates_for_epoch(self):
        if self.gate_provider is None:
            return None
        base = self.optimizer._step
        return lambda k: self.gate_provider(base + k)

    def _drive(self, epochs):
        """Run one sampler's epoch generator on its own engine and snapshot when it is exhausted."""
        try:
            noise, sched = next(epochs)
            while True:
                seen = self.engine.run_epoch(self.train_loader, noise, sched=sched, eps_per_step=self._eps_for_epoch(),
                                             gates_per_step=self._gates_for_epoch())
                noise, sched = epochs.send(seen)
        except StopIteration:
            pass
        return self._snapshot()

    def _debug_metrics(self, val_loader, seen, extra=None, wandb_debug=False):
        metrics = {'train_loss': self.engine.loss_acc.item() / self.dataset_size,
                   'val_loss': self.compute_val_loss(val_loader)}
        metrics.update(extra or {})
        print(metrics)
        if wandb_debug and wandb is not None:
            wandb.log(metrics)


class SGHMC(_ChainSampler):

    def __init__(self, hyperparameters, model=None, train_loader=None, model_loss='multi_class_linear_output',
                 device=torch.device('cpu'), *, kernels=None, use_graph=None, seed=None):
        """hyperparameters: {'lr', 'prior_std', 'num_samples', 'alpha', 'burn_in_epochs'} (sghmc.py:28-32)."""
        if hyperparameters == None:  # noqa: E711  (reference default, sghmc.py:23-25)
            hyperparameters = {'lr': 0.001, 'prior_std': 10, 'num_samples': 2, 'alpha': 0.1, 'burn_in_epochs': 10}
        super().__init__(hyperparameters, model, train_loader, device)
        self.lr = hyperparameters['lr']
        self.prior_std = hyperparameters['prior_std']
        self.num_samples = hyperparameters['num_samples']
        self.alpha = hyperparameters['alpha']
        self.burn_in_epochs = hyperparameters['burn_in_epochs']
        self._setup_chain(model, train_loader, model_loss, device, kernels, use_graph, lr=self.lr,
                          momentum=1 - self.alpha, weight_decay=1 / (self.prior_std ** 2), seed=seed)
        self.burnt_in = False
        self.epochs_run = 0
        self.lr_final = self.lr / 2
        self.optimizer_scheduler = CosineAnnealingLR(optimizer=self.optimizer,
                                                     T_max=(self.burn_in_epochs + self.num_samples), )

    def _read_hyp(self, hyperparameters):
        self.lr = hyperparameters['lr']
        self.prior_std = hyperparameters['prior_std']
        self.num_samples = hyperparameters['num_samples']
        self.burn_in_epochs = hyperparameters['burn_in_epochs']

    def update_hyp(self, hyperparameters):
        self._read_hyp(hyperparameters)
        self.alpha = hyperparameters['alpha']
        self.model = reset_model(self.model)
        self.burnt_in = False
        self.epochs_run = 0
        self._new_optimizer(self.lr, 1 - self.alpha, 1 / (self.prior_std ** 2))
        self.lr_final = self.lr / 2
        self.optimizer_scheduler = CosineAnnealingLR(optimizer=self.optimizer,
                                                     T_max=self.burn_in_epochs + self.num_samples,
                                                     eta_min=self.lr_final)

    def _epochs(self, val_loader=None, debug_val_loss=False, wandb_debug=False):
        """The epochs of the next sample as a generator: yields (noise, schedule table) for the driver to
        run, receives the number of examples seen, does the end-of-epoch bookkeeping (sghmc.py:67-98)."""
        if self.burnt_in is False:
            epochs = self.burn_in_epochs + 1
            self.burnt_in = True
        else:
            epochs = 1
        for epoch in range(epochs):
            noise = bool(epoch > 0.8 * epochs or self.burnt_in)          # sghmc.py:83 (always True)
            seen = yield noise, None
            self.optimizer_scheduler.step()
            if debug_val_loss:
                self._debug_metrics(val_loader, seen, {'lr': self.optimizer_scheduler.get_last_lr()}, wandb_debug)

    def sample_iterative(self, val_loader=None, debug_val_loss=False, wandb_debug=False):
        if not isinstance(self.model, torch.nn.Module):
            raise NotImplementedError
        return self._drive(self._epochs(val_loader, debug_val_loss, wandb_debug))

    def sample(self, num_samples=None, val_loader=None, debug_val_loss=False, wandb_debug=False):
        if num_samples is None:
            num_samples = self.num_samples
        if not isinstance(self.model, torch.nn.Module):
            raise NotImplementedError
        return [self.sample_iterative(val_loader=val_loader, debug_val_loss=debug_val_loss, wandb_debug=wandb_debug)
                for _ in range(num_samples)]


class SGLD(SGHMC):
    """sgld.py:8-35 — SGHMC with alpha forced to 1 (momentum 0 => the d_p.mul(-lr) branch,
    optim_sghmc.py:61-62, 12 B/param). Quirk kept: update_hyp rebuilds the optimizer but NOT the
    scheduler (sgld.py:25-35), so the learning rate stays constant afterwards."""

    def __init__(self, hyperparameters, model=None, train_loader=None, model_loss='multi_class_linear_output',
                 device=torch.device('cpu'), **kw):
        if hyperparameters == None:  # noqa: E711
            hyperparameters = {'lr': 0.001, 'prior_std': 10, 'num_samples': 2, 'alpha': 0.1, 'burn_in_epochs': 10}
        hyperparameters['alpha'] = 1.          # mutates the caller's dict, as sgld.py:22 does
        super().__init__(hyperparameters, model, train_loader, model_loss, device, **kw)

    def update_hyp(self, hyperparameters):
        self._read_hyp(hyperparameters)
        self.alpha = 1.
        self.model = reset_model(self.model)
        self._new_optimizer(self.lr, 1 - self.alpha, 1 / (self.prior_std ** 2))
        self.burnt_in = False
        self.epochs_run = 0
