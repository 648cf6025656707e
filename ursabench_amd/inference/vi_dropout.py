"""MCdropout — drop-in for URSABench/inference/vi_dropout.py:13-131.

The reference swaps the given model for its `<Class>_dropout` sibling (a FRESH random init; the `dropout`
hyper-parameter is read but 0.2 is hard-coded, vi_dropout.py:21), trains it with SGD(momentum, weight decay)
under OneCycleLR stepped once per MINIBATCH (:59-61,107; CosineAnnealingLR after update_hyp, :83-84) and
returns the same live model `num_samples` times: the dropout masks stay on in eval mode (F.dropout's default
training=True in the `_dropout` models), so T forwards of that one model are the MC-dropout ensemble.

Here the trajectory runs on the ChainEngine with FlatSGD (K1 in SGD mode, one launch per step, hipGraph
replay); the per-minibatch (lr, momentum) pairs of an epoch are taken from the very torch scheduler the
reference uses, stepped on the host, and uploaded as the device schedule table the update kernel's control
block walks — so the epoch still replays one graph with no host round trip. Evaluation needs nothing
special: the tasks run a module that is not bank-resident eagerly, and every forward draws fresh masks.
"""
import inspect

import torch
from torch.optim.lr_scheduler import CosineAnnealingLR, OneCycleLR

from .. import models
from ..util import get_loss_criterion, reset_model
from .engine import ChainEngine
from .flat_sgd import FlatSGD
from .inference_base import _Inference


def change_to_dropout_model(model, dropout):
    """vi_dropout.py:13-23 — `<Class>_dropout(dropout=0.2, **ctor-arguments read back from the model)`.
    Quirk kept: the `dropout` argument is ignored."""
    signature = inspect.signature(model.__init__)
    kwargs = {key: getattr(model, key) for key in signature.parameters.keys()}
    model_cfg = getattr(models, model.__class__.__name__ + '_dropout')
    return model_cfg(dropout=0.2, **kwargs)


class MCdropout(_Inference):
    def __init__(self, hyperparameters, model=None, train_loader=None, model_loss='multi_class_linear_output',
                 device=torch.device('cpu'), *, kernels=None, use_graph=None):
        if hyperparameters == None:  # noqa: E711  (vi_dropout.py:36-38)
            hyperparameters = {'lr': 0.1, 'epochs': 10, 'dropout': 0.2, 'lengthscale': 0.01, 'num_samples': 10,
                               'momentum': 0.9, 'weight_decay': 0}
        super().__init__(hyperparameters, model, train_loader, device)
        self.lr = hyperparameters['lr']
        self.num_samples = hyperparameters['num_samples']
        self.burn_in_epochs = hyperparameters['epochs']
        self.dropout = hyperparameters['dropout']
        self.momentum = hyperparameters['momentum']
        self.model = change_to_dropout_model(model, self.dropout).to(device)
        self.train_loader = train_loader
        self.device = device
        self.dataset_size = len(train_loader.dataset)
        self.weight_decay = self._weight_decay(hyperparameters)
        self._kernels = kernels
        self.optimizer = FlatSGD(self.model.parameters(), lr=self.lr, momentum=self.momentum,
                                 weight_decay=self.weight_decay, kernels=kernels, module=self.model)
        self.arena = self.optimizer.arena
        self.loss_criterion = get_loss_criterion(loss=model_loss)
        self.engine = ChainEngine(self.model, self.optimizer, self.loss_criterion, device, use_graph)
        self.burnt_in = False
        self.epochs_run = 0
        self.lr_final = self.lr / 100.
        self.optimizer_scheduler = OneCycleLR(optimizer=self.optimizer, max_lr=self.lr * 5,
                                              steps_per_epoch=len(self.train_loader),
                                              epochs=self.burn_in_epochs + self.num_samples)

    def _weight_decay(self, h):
        if h['weight_decay'] != 0:
            return h['weight_decay']
        return h['lengthscale'] ** 2 * (1 - self.dropout) / (2. * self.dataset_size)      # vi_dropout.py:53

    def update_hyp(self, hyperparameters):
        self.lr = hyperparameters['lr']
        self.num_samples = hyperparameters['num_samples']
        self.epochs = hyperparameters['epochs']            # quirk: the loop keeps using burn_in_epochs (:68 vs :103)
        self.dropout = hyperparameters['dropout']
        self.momentum = hyperparameters['momentum']
        self.weight_decay = self._weight_decay(hyperparameters)
        old = self.optimizer
        self.optimizer = FlatSGD(self.model.parameters(), lr=self.lr, momentum=self.momentum,
                                 weight_decay=self.weight_decay, kernels=self._kernels, arena=self.arena)
        self.optimizer.adopt_device_state(old)
        self.engine.set_optimizer(self.optimizer)
        self.model = reset_model(self.model).to(self.device)
        self.burnt_in = False
        self.epochs_run = 0
        self.lr_final = self.lr / 2
        self.optimizer_scheduler = CosineAnnealingLR(optimizer=self.optimizer,
                                                     T_max=self.burn_in_epochs + self.num_samples, eta_min=self.lr_final)

    def _epoch_table(self):
        """(lr, momentum) of every minibatch of the next epoch: the scheduler is stepped here, on the host,
        exactly as often as the reference steps it inside its loop (vi_dropout.py:107)."""
        group = self.optimizer.param_groups[0]
        self.optimizer._opt_called = True          # LRScheduler's step-order check: the steps follow, on the device
        rows = []
        for _ in range(len(self.train_loader)):
            rows.append((float(group['lr']), float(group['momentum'])))
            self.optimizer_scheduler.step()
        after = (group['lr'], group['momentum'])
        return torch.tensor(rows, dtype=torch.float32), rows[0], after

    def sample_iterative(self, val_loader=None, debug_val_loss=False, wandb_debug=False):
        if not issubclass(self.model.__class__, torch.nn.Module):
            raise NotImplementedError
        if self.burnt_in is False:
            epochs = self.burn_in_epochs + 1
            self.burnt_in = True
        else:
            epochs = 1
        group = self.optimizer.param_groups[0]
        for _ in range(epochs):
            table, first, after = self._epoch_table()
            group['lr'], group['momentum'] = first          # ctl_begin reads the epoch's first pair here
            self.engine.run_epoch(self.train_loader, False, sched=table)
            group['lr'], group['momentum'] = after          # what the reference's scheduler leaves behind
            self.epochs_run += 1
            if debug_val_loss:
                print({'train_loss': self.engine.loss_acc.item() / self.dataset_size,
                       'val_loss': self.compute_val_loss(val_loader)})
        return self.model

    def sample(self, num_samples=None, val_loader=None, debug_val_loss=False, wandb_debug=False):
        if num_samples is None:
            num_samples = self.num_samples
        if not issubclass(self.model.__class__, torch.nn.Module):
            raise NotImplementedError
        return [self.sample_iterative(val_loader=val_loader, debug_val_loss=debug_val_loss, wandb_debug=wandb_debug)
                for _ in range(num_samples)]
