"""SWA / SWAG — drop-ins for URSABench/inference/swa.py:13-178 and swag.py:12-147.

Reference flow: SGD(momentum, wd) epochs with a piecewise-linear LR (swa.py:92-101); after burn-in
every epoch's weights are flattened TO THE CPU and folded into running first/second moments
(swa.py:79-88); a sample is torch.normal(mean, sqrt(clamp(sq-mean^2, 1e-30))) on the CPU copied
tensor by tensor into `swag_model`, followed by a BatchNorm refresh pass and a CPU deep copy
(swag.py:84-126). Here the trajectory runs on the ChainEngine (FlatSGD in hipGraph replay), the
moments live in HBM in arena layout and are updated by one K2 launch, the draw is one K3 launch
straight into swag_model's arena (Philox noise), bn_update runs on the device and the sample is a
device-to-device snapshot.

reference_quirks=True (default) reproduces two defects of the reference bit for bit
(SURVEY.md fact 8): SWAG never increments `num_models_collected`, so each collect overwrites the
moments with the last iterate (variance clamps to 1e-30), and the Gaussian draw is discarded
(`weight_sample = self.weight_mean`, swag.py:98,118) — every "sample" is the last SGD iterate plus
a BN refresh. reference_quirks=False is SWAG as published: running moments over the collected
iterates and theta = mean + sqrt(var) * eps.

The PCA/covariance subspace (`subspaces.py`) is out of scope: `subspace` is a no-op sink and
full_cov=True raises (in the reference that branch reads a non-existent attribute).
"""
from copy import deepcopy

import torch

from ..arena import FlatArena, MemberBank
from .._capture import side_streams
from ..util import adjust_learning_rate, bn_update, bn_update_many, get_loss_criterion, reset_model
from .engine import ChainEngine
from .flat_sgd import FlatSGD
from .inference_base import _Inference

_DEFAULT_HYP = {'swag_lr': 0.001, 'swag_wd': 0.001, 'lr_init': 0.001, 'num_samples': 20, 'momentum': 0.1,
                'burn_in_epochs': 100, 'num_iterates': 50}


class _NullSubspace:
    """Stand-in for subspaces.Subspace: SWAG only feeds it a deviation vector (swa.py:89-90)."""

    def collect_vector(self, vector):
        pass

    def get_space(self):
        raise NotImplementedError('covariance subspaces are outside the SG-MCMC/BMA hot path')


class SWA(_Inference):
    def __init__(self, hyperparameters, model=None, train_loader=None, model_loss='multi_class_linear_output',
                 device=torch.device('cpu'), *, kernels=None, use_graph=None, reference_quirks=True, seed=None,
                 **subspace_kwargs):
        super().__init__(hyperparameters, model=None, train_loader=None, device=torch.device('cpu'))
        if hyperparameters == None:  # noqa: E711
            hyperparameters = dict(_DEFAULT_HYP)
        if not isinstance(model, torch.nn.Module):
            raise NotImplementedError
        self.hyperparameters = hyperparameters
        self.reference_quirks = reference_quirks
        self.device = device
        self._kernels = kernels
        self.model = model.to(device)
        self.swag_model = deepcopy(self.model)
        self.train_loader = train_loader
        self.loss_criterion = get_loss_criterion(loss=model_loss)
        self.dataset_size = len(train_loader.dataset)
        self.var_clamp = 1e-30
        self._read_hyp(hyperparameters)
        self.optimizer = FlatSGD(params=self.model.parameters(), lr=self.lr_init, momentum=self.momentum,
                                 weight_decay=self.swag_wd, kernels=kernels, module=self.model, seed=seed)
        self.arena = self.optimizer.arena
        self.kernels = self.optimizer.kernels
        self.swag_arena = FlatArena(self.swag_model.parameters(), module=self.swag_model)
        self.num_parameters = self.arena.num_parameters
        self.engine = ChainEngine(self.model, self.optimizer, self.loss_criterion, device, use_graph)
        self.bank = MemberBank(self.swag_arena)
        self.seed = self.optimizer.seed
        self._draws = 0
        self.eps_provider = None            # callable(draw_index) -> flat eps (arena layout): parity runs
        self.subspace = _NullSubspace()
        self.cov_factor = None
        self._reset_moments()

    # -- hyper-parameters / state ------------------------------------------------------------
    def _read_hyp(self, h):
        self.burn_in_epochs = h['burn_in_epochs']
        self.num_iterates = h['num_iterates']
        self.momentum = h['momentum']
        self.lr_init = h['lr_init']
        self.swag_lr = h['swag_lr']
        self.swag_wd = h['swag_wd']
        self.subspace_type = h.get('subspace_type', 'pca')

    def _reset_moments(self):
        self._mean = torch.zeros(self.arena.n, device=self.device)
        self._sq = torch.zeros(self.arena.n, device=self.device)
        self._std = None                    # sqrt(clamp(sq - mean^2)) of the CURRENT moments, filled on first draw
        self.num_models_collected = torch.zeros(1, dtype=torch.long)
        self.burnt_in = False
        self.epochs_run = 0

    @property
    def weight_mean(self):
        """[P] in model.parameters() order, like swa.py:25 (device-resident here)."""
        return self._mean[self.arena.layout.gather_index(self.device)]

    @property
    def sq_mean(self):
        return self._sq[self.arena.layout.gather_index(self.device)]

    def update_hyp(self, hyperparameters, **subspace_kwargs):
        self._reset_moments()
        self.hyperparameters = hyperparameters
        self._read_hyp(hyperparameters)
        self.model = reset_model(self.model)
        self.swag_model = reset_model(self.swag_model)
        old = self.optimizer
        self.optimizer = FlatSGD(params=self.model.parameters(), lr=self.lr_init, momentum=self.momentum,
                                 weight_decay=self.swag_wd, kernels=self._kernels, arena=self.arena, seed=self.seed)
        self.optimizer.adopt_device_state(old)      # keep the control block a captured graph may hold (like the chain samplers)
        self.engine.set_optimizer(self.optimizer)

    # -- the three SWA primitives ------------------------------------------------------------
    def _collect_model(self):
        """swa.py:79-90 — one K2 launch on the live arena; n = num_models_collected at call time."""
        n = self.num_models_collected.item()
        self.kernels.swag_collect(self._mean, self._sq, self.arena.theta, decay=n / (n + 1.0), denom=n + 1.0)
        self._std = None                    # the moments moved
        self.subspace.collect_vector(None)

    def _schedule(self, epoch):
        """swa.py:92-101."""
        t = epoch / self.burn_in_epochs
        lr_ratio = self.swag_lr / self.lr_init
        if t <= 0.5:
            factor = 1.0
        elif t <= 0.9:
            factor = 1.0 - (1.0 - lr_ratio) * (t - 0.5) / 0.4
        else:
            factor = lr_ratio
        return self.lr_init * factor

    def _set_swa(self):
        """swa.py:103-104 — swag_model <- mean (same arena layout: one device copy)."""
        self.swag_arena.theta.copy_(self._mean)

    def _get_mean_and_variance(self):
        """swa.py:106-108."""
        mean = self.weight_mean
        return mean, torch.clamp(self.sq_mean - mean ** 2, self.var_clamp)

    def get_space(self, export_cov_factor=True):
        mean, variance = self._get_mean_and_variance()
        if not export_cov_factor:
            return mean.clone(), variance.clone()
        raise NotImplementedError('covariance subspaces are outside the SG-MCMC/BMA hot path')

    def _train_epoch(self, val_loader, debug_val_loss, wandb_debug):
        adjust_learning_rate(self.optimizer, self._schedule(self.epochs_run))
        self.engine.run_epoch(self.train_loader, False)
        self.epochs_run += 1
        if debug_val_loss:
            print({'train_loss': self.engine.loss_acc.item() / self.dataset_size,
                   'val_loss': self.compute_val_loss(val_loader)})

    # -- SWA sampling (swa.py:123-178) -------------------------------------------------------
    def sample_iterative(self, update_bn_swa=True, val_loader=None, debug_val_loss=False, wandb_debug=False):
        if self.burnt_in is False:
            epochs = self.burn_in_epochs + 1
            self.burnt_in = True
        else:
            epochs = 1
        self.num_models_collected += 1             # before training (swa.py:130): first collect sees n = 1
        for _ in range(epochs):
            self._train_epoch(val_loader, debug_val_loss, wandb_debug)
        self._collect_model()
        if update_bn_swa:
            self._set_swa()
            bn_update(self.train_loader, self.swag_model, device=self.device)
        return self.swag_model

    def sample(self, num_samples=None, val_loader=None, debug_val_loss=False, wandb_debug=False):
        if num_samples is None:
            num_samples = self.num_iterates
        return [self.sample_iterative(update_bn_swa=(i == num_samples - 1), val_loader=val_loader,
                                      debug_val_loss=debug_val_loss, wandb_debug=wandb_debug)
                for i in range(num_samples)]


class SWAG(SWA):
    def __init__(self, hyperparameters, model=None, train_loader=None, model_loss='multi_class_linear_output',
                 device=torch.device('cpu'), **kw):
        super().__init__(hyperparameters, model=model, train_loader=train_loader, model_loss=model_loss,
                         device=device, **kw)
        if hyperparameters == None:  # noqa: E711
            hyperparameters = dict(_DEFAULT_HYP)
        self.num_samples = hyperparameters['num_samples']
        self.weight_variance = None

    def update_hyp(self, hyperparameters, **subspace_kwargs):
        super().update_hyp(hyperparameters, **subspace_kwargs)
        self.weight_variance = None
        self.num_samples = hyperparameters['num_samples']

    def _draw(self, theta_out):
        """One member: theta = mean + std * eps (swag.py:84-86 + swa.py:106-108). The standard deviation is the same for
        every member drawn from these moments: computed once (K3 `ursa_swag_std_f32`) and kept, so the per-member launch
        (`ursa_swag_draw_std_f32`, 12 B/param) carries no square roots — bit-identical to the fused draw."""
        if self._std is None:
            self._std = torch.empty_like(self._mean)
            self.kernels.swag_std(self._std, self._mean, self._sq, var_clamp=self.var_clamp, scale=1.0)
        eps = None if self.eps_provider is None else self.eps_provider(self._draws)
        self.kernels.swag_draw_std(theta_out, self._mean, self._std, seed=self.seed, draw=self._draws, eps=eps)

    def _draw_into_swag_model(self):
        if self.reference_quirks:
            self.swag_arena.theta.copy_(self._mean)              # swag.py:98 — the draw is discarded
        else:
            self._draw(self.swag_arena.theta)
        self._draws += 1

    def run_trajectory(self, val_loader=None, debug_val_loss=False, wandb_debug=False):
        """swag.py:55-83 — the SGD trajectory of the first sample_iterative call: burn_in_epochs +
        num_iterates epochs, moments collected after burn-in."""
        for epoch in range(self.burn_in_epochs + self.num_iterates):
            self._train_epoch(val_loader, debug_val_loss, wandb_debug)
            if epoch >= self.burn_in_epochs:
                self._collect_model()
                if not self.reference_quirks:
                    self.num_models_collected += 1
        self.adopt_moments()

    def adopt_moments(self):
        """Mark the moments in self._mean / self._sq as final (after run_trajectory, or after another rank's
        moments were received into them): later sample_iterative calls only draw."""
        self.burnt_in = True
        self._std = None
        _, self.weight_variance = self._get_mean_and_variance()

    def sample_iterative(self, update_bn=True, val_loader=None, debug_val_loss=False, wandb_debug=False,
                         full_cov=False):
        if full_cov:
            raise NotImplementedError('full_cov needs the covariance subspace, outside the hot path')
        if self.burnt_in is False:
            self.run_trajectory(val_loader, debug_val_loss, wandb_debug)
        self._draw_into_swag_model()
        if update_bn:
            bn_update(self.train_loader, self.swag_model, device=self.device)
        return self.bank.snapshot(self.swag_model)

    LANES = 4      # members whose BatchNorm refresh passes run concurrently in sample() ...
    GROUP_MIN_PARAMS = 4_000_000   # ... for networks at least this large: a small network's refresh pass is bound by the
    #                                host's launch rate, which extra streams do not raise (PreResNet-20, 3 samples: 3.05 s
    #                                one at a time, 3.25 s grouped; WideResNet-28-10: 5.90 -> 5.16 s per member)

    def _lanes(self, n):
        """Scratch copies of swag_model (own flat arenas) for concurrent draw + bn_update of several members."""
        lanes = getattr(self, '_lane_models', None)
        if lanes is None:
            lanes = self._lane_models = [(self.swag_model, self.swag_arena)]
        while len(lanes) < n:
            m = deepcopy(self.swag_model)
            lanes.append((m, FlatArena(m.parameters(), module=m)))
        return lanes[:n]

    def _sample_group(self, k):
        """k members at once: k draws (K3) into k scratch models, ONE pass over the training set refreshing the
        BatchNorm statistics of all k on k streams, k device snapshots. Each member is, bit for bit, what
        sample_iterative() returns for the same draw index."""
        lanes = self._lanes(k)
        lanes = lanes[1:] + lanes[:1]        # the group's LAST member is formed in swag_model itself, where the reference leaves it
        for model, arena in lanes:
            if self.reference_quirks:
                arena.theta.copy_(self._mean)                     # swag.py:98 — the draw is discarded
            else:
                self._draw(arena.theta)
            self._draws += 1
        on_hip = torch.device(self.device).type == 'cuda'
        main = self.swag_arena
        before = [b.clone() for _, b in main.ibufs]               # BatchNorm step counters of THE swag_model
        bn_update_many(self.train_loader, [m for m, _ in lanes], device=self.device,
                       streams=side_streams(self.device, k) if on_hip and k > 1 else None)
        # The reference refreshes one and the same swag_model for every sample, so its num_batches_tracked keeps
        # counting across samples (reset_bn leaves it alone, util.py:195-199): member j of this group carries
        # counter_before + (j + 1) * forwards, and the swag_model ends at counter_before + k * forwards.
        done = [b - b0 for (_, b), b0 in zip(main.ibufs, before)]  # forwards of one refresh pass, per counter
        out = []
        for j, (model, arena) in enumerate(lanes):
            row, irow = self.bank.new_row()
            with torch.no_grad():
                self.bank.theta_of(row).copy_(arena.theta)
                if arena.fbuf is not None and arena.fbuf.numel():
                    row[arena.layout.padded:].copy_(arena.fbuf)
                for dst, b0, d in zip(irow, before, done):
                    dst.copy_(b0 + (j + 1) * d)
            out.append(self.bank.materialise(row, irow, self.swag_model))
        with torch.no_grad():
            for (_, b), b0, d in zip(main.ibufs, before, done):
                b.copy_(b0 + k * d)
        return out

    def sample(self, num_samples=None, val_loader=None, debug_val_loss=False, wandb_debug=False, full_cov=False):
        if num_samples is None:
            num_samples = self.num_samples
        if full_cov:
            raise NotImplementedError('full_cov needs the covariance subspace, outside the hot path')
        if self.LANES <= 1 or num_samples <= 1 or self.num_parameters < self.GROUP_MIN_PARAMS:
            return [self.sample_iterative(update_bn=True, val_loader=val_loader, debug_val_loss=debug_val_loss,
                                          wandb_debug=wandb_debug, full_cov=full_cov) for _ in range(num_samples)]
        if self.burnt_in is False:
            self.run_trajectory(val_loader, debug_val_loss, wandb_debug)
        out = []
        while len(out) < num_samples:
            out.extend(self._sample_group(min(self.LANES, num_samples - len(out))))
        return out
