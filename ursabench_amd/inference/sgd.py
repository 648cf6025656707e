"""SGD — the "one-sample" baseline of URSABench/inference/sgd.py:19-113, so `getattr(inference, 'SGD')`
resolves. Not an MCMC method: SGD-momentum epochs with cosine annealing, then the live model is
returned (the same object for every requested sample). The trajectory runs on the ChainEngine with
FlatSGD (K1 SGD mode). Quirks kept: update_hyp stores `epochs` but the loop keeps using the
constructor's `burn_in_epochs`; eta_min is lr/100 in the constructor and lr/2 after update_hyp.
"""
import torch
from torch.optim.lr_scheduler import CosineAnnealingLR

from ..util import get_loss_criterion, reset_model
from .engine import ChainEngine
from .flat_sgd import FlatSGD
from .inference_base import _Inference


class SGD(_Inference):
    def __init__(self, hyperparameters, model=None, train_loader=None, model_loss='multi_class_linear_output',
                 device=torch.device('cpu'), *, kernels=None, use_graph=None):
        if hyperparameters == None:  # noqa: E711
            hyperparameters = {'lr': 0.1, 'epochs': 10, 'momentum': 0.9, 'weight_decay': 0.001}
        super().__init__(hyperparameters, model, train_loader, device)
        if not isinstance(model, torch.nn.Module):
            raise NotImplementedError
        self.lr = hyperparameters['lr']
        self.num_samples = 1
        self.burn_in_epochs = hyperparameters['epochs']
        self.momentum = hyperparameters['momentum']
        self.weight_decay = hyperparameters['weight_decay']
        self.model = model.to(device)
        self.dataset_size = len(train_loader.dataset)
        self._kernels = kernels
        self.optimizer = FlatSGD(self.model.parameters(), lr=self.lr, momentum=self.momentum,
                                 weight_decay=self.weight_decay, kernels=kernels, module=self.model)
        self.arena = self.optimizer.arena
        self.loss_criterion = get_loss_criterion(loss=model_loss)
        self.engine = ChainEngine(self.model, self.optimizer, self.loss_criterion, device, use_graph)
        self.burnt_in = False
        self.epochs_run = 0
        self.lr_final = self.lr / 100.
        self.optimizer_scheduler = CosineAnnealingLR(optimizer=self.optimizer, T_max=self.burn_in_epochs + self.num_samples,
                                                     eta_min=self.lr_final)

    def update_hyp(self, hyperparameters):
        self.lr = hyperparameters['lr']
        self.num_samples = 1
        self.epochs = hyperparameters['epochs']
        self.momentum = hyperparameters['momentum']
        self.weight_decay = hyperparameters['weight_decay']
        old = self.optimizer
        self.optimizer = FlatSGD(self.model.parameters(), lr=self.lr, momentum=self.momentum,
                                 weight_decay=self.weight_decay, kernels=self._kernels, arena=self.arena)
        self.optimizer.adopt_device_state(old)      # keep the control block a captured graph may hold (like the chain samplers)
        self.engine.set_optimizer(self.optimizer)
        self.model = reset_model(self.model).to(self.device)
        self.burnt_in = False
        self.epochs_run = 0
        self.lr_final = self.lr / 2
        self.optimizer_scheduler = CosineAnnealingLR(optimizer=self.optimizer, T_max=self.burn_in_epochs + self.num_samples,
                                                     eta_min=self.lr_final)

    def sample_iterative(self, val_loader=None, debug_val_loss=False, wandb_debug=False):
        if self.burnt_in is False:
            epochs = self.burn_in_epochs + 1
            self.burnt_in = True
        else:
            epochs = 0
        for _ in range(epochs):
            self.engine.run_epoch(self.train_loader, False)
            if debug_val_loss:
                print({'train_loss': self.engine.loss_acc.item() / self.dataset_size,
                       'val_loss': self.compute_val_loss(val_loader)})
            self.optimizer_scheduler.step()
        return self.model

    def sample(self, num_samples=None, val_loader=None, debug_val_loss=False, wandb_debug=False):
        if num_samples is None:
            num_samples = self.num_samples
        return [self.sample_iterative(val_loader=val_loader, debug_val_loss=debug_val_loss, wandb_debug=wandb_debug)
                for _ in range(num_samples)]
