"""cSGHMC / cSGLD — drop-ins for URSABench/inference/csghmc.py:13-127 and csgld.py:9-36.

Same optimizer update (K1); what differs from SGHMC is host-side schedule only: a cosine
learning rate with warm restarts evaluated PER ITERATION (csghmc.py:64-72), Langevin noise only
in the tail of each cycle (:89-93) and samples emitted only in the last
`num_samples_per_cycle` epochs of a cycle (:106). The per-iteration (lr, noise-scale) pairs of
an epoch are computed on the host in float64 exactly as the reference does and uploaded as one
small table; the update kernel's control block walks it on the device, so the epoch still runs
as hipGraph replays with no host round trip.
"""
import math

import numpy as np
import torch

from ..util import reset_model
from .sghmc import _ChainSampler


class cSGHMC(_ChainSampler):

    def __init__(self, hyperparameters, model=None, train_loader=None, model_loss='multi_class_linear_output',
                 device=torch.device('cpu'), *, kernels=None, use_graph=None, seed=None):
        if hyperparameters == None:  # noqa: E711  (csghmc.py:17-19)
            hyperparameters = {'lr_0': 0.001000, 'prior_std': 10.1000, 'num_samples_per_cycle': 5, 'cycle_length': 20,
                               'burn_in_epochs': 5, 'num_cycles': 10, 'alpha': 1., }
        super().__init__(hyperparameters, model, train_loader, device)
        self._read_hyp(hyperparameters)
        self.alpha = hyperparameters['alpha']
        self.batch_size = train_loader.batch_size
        # float, and over-counts by one (csghmc.py:30-31): kept, it shifts the restart points
        self.num_batch = max(1, len(train_loader.dataset) / self.batch_size + 1)
        self._setup_chain(model, train_loader, model_loss, device, kernels, use_graph, lr=self.lr_0,
                          momentum=1 - self.alpha, weight_decay=1 / (self.prior_std ** 2), seed=seed)
        self.burnt_in = False
        self.epochs_run = 0
        self.dataloader_batch_size = self.train_loader.batch_size
        self._derive()

    def _read_hyp(self, h):
        self.lr_0 = h['lr_0']
        self.prior_std = h['prior_std']
        self.num_samples_per_cycle = h['num_samples_per_cycle']
        self.cycle_length = h['cycle_length']
        self.burn_in_epochs = h['burn_in_epochs']
        self.num_cycles = h['num_cycles']

    def _derive(self):
        self.total_epochs = self.cycle_length * self.num_cycles
        self.total_iterations = self.total_epochs * self.num_batch
        assert ((self.cycle_length - self.burn_in_epochs - self.num_samples_per_cycle) > 0)

    def update_hyp(self, hyperparameters):
        self._read_hyp(hyperparameters)
        self.alpha = hyperparameters['alpha']
        self._rebuild()

    def _rebuild(self):
        self.model = reset_model(self.model)
        self._new_optimizer(self.lr_0, 1 - self.alpha, 1 / (self.prior_std ** 2))
        self.burnt_in = False
        self.epochs_run = 0
        assert ((self.cycle_length - self.burn_in_epochs - self.num_samples_per_cycle) > 0)

    def _adjust_learning_rate(self, optimizer, epoch, batch_idx):
        """csghmc.py:64-72, verbatim arithmetic (numpy float64, float floor-division)."""
        rcounter = epoch * self.num_batch + batch_idx
        cos_inner = np.pi * (rcounter % (self.total_iterations // self.num_cycles))
        cos_inner /= self.total_iterations // self.num_cycles
        cos_out = np.cos(cos_inner) + 1
        lr = 0.5 * cos_out * self.lr_0
        for param_group in optimizer.param_groups:
            param_group['lr'] = lr
        return lr

    def _epoch_table(self):
        mu = self.optimizer.param_groups[0]['momentum']
        rows = []
        for b in range(len(self.train_loader)):
            lr = float(self._adjust_learning_rate(self.optimizer, self.epochs_run, b))
            rows.append((lr, math.sqrt(2 * (1 - mu) * lr)))
        self.lr = rows[-1][0]                       # what the reference leaves in self.lr / param_groups
        return torch.tensor(rows, dtype=torch.float32)

    def _epochs(self, val_loader=None, debug_val_loss=False, wandb_debug=False):
        """Epoch generator (see SGHMC._epochs): runs epochs until one falls in the collecting tail of a cycle
        (csghmc.py:77-111)."""
        while True:
            noise = (self.epochs_run % self.cycle_length) + 1 > (self.cycle_length - self.burn_in_epochs
                                                                 - self.num_samples_per_cycle)
            table = self._epoch_table()
            self.optimizer.param_groups[0]['lr'] = float(table[0, 0])   # ctl_begin reads the epoch's first lr here
            seen = yield bool(noise), table
            self.optimizer.param_groups[0]['lr'] = self.lr
            self.epochs_run += 1
            print('Epoch: ', self.epochs_run, ' lr: ', self.lr)
            if debug_val_loss:
                self._debug_metrics(val_loader, seen, None, wandb_debug)
            if ((self.epochs_run - 1) % self.cycle_length) >= (self.cycle_length - self.num_samples_per_cycle):
                return

    def sample_iterative(self, val_loader=None, debug_val_loss=False, wandb_debug=False):
        if not isinstance(self.model, torch.nn.Module):
            raise NotImplementedError
        return self._drive(self._epochs(val_loader, debug_val_loss, wandb_debug))

    def sample(self, num_samples=None, val_loader=None, debug_val_loss=False, wandb_debug=False):
        if num_samples is None:
            num_samples = self.num_samples_per_cycle * self.num_cycles
        if not isinstance(self.model, torch.nn.Module):
            raise NotImplementedError
        return [self.sample_iterative(val_loader=val_loader, debug_val_loss=debug_val_loss, wandb_debug=wandb_debug)
                for _ in range(num_samples)]


class cSGLD(cSGHMC):
    """csgld.py:9-36 — alpha forced to 1."""

    def __init__(self, hyperparameters, model=None, train_loader=None, model_loss='multi_class_linear_output',
                 device=torch.device('cpu'), **kw):
        if hyperparameters == None:  # noqa: E711
            hyperparameters = {'lr_0': 0.001000, 'prior_std': 10.1000, 'num_samples_per_cycle': 5, 'cycle_length': 20,
                               'burn_in_epochs': 5, 'num_cycles': 10, 'alpha': 1., }
        hyperparameters['alpha'] = 1.
        super().__init__(hyperparameters, model, train_loader, model_loss, device, **kw)

    def update_hyp(self, hyperparameters):
        self._read_hyp(hyperparameters)
        self.alpha = 1.
        self._rebuild()
