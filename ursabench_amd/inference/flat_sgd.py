"""FlatSGD — torch.optim.SGD(momentum, weight_decay) over the flat arena, for the SWA/SWAG
training trajectory (URSABench/inference/swa.py:41-42, swag.py:55-70). Same param_groups keys as
torch.optim.SGD; one launch of the K1 kernel in SGD mode per step, bit-identical to
torch.optim.SGD's single-tensor update (tests/golden/sgd_steps.npz)."""
import torch
from torch.optim.optimizer import required

from .. import _native
from .optim_sghmc import optimSGHMC


class FlatSGD(optimSGHMC):
    def __init__(self, params, lr=required, momentum=0, dampening=0, weight_decay=0, nesterov=False, **kw):
        if dampening != 0 or nesterov:
            raise NotImplementedError('FlatSGD implements dampening=0, nesterov=False (all the reference uses)')
        super().__init__(params, lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay,
                         num_training_samples=None, nesterov=False, **kw)

    def _scalars(self, group, add_langevin_noise, gi):
        mu, lr, wd = group['momentum'], group['lr'], group['weight_decay']
        flags = _native.STEP_SGD
        if wd != 0:
            flags |= _native.STEP_WD
        if mu != 0 and not self._has_mom[gi]:
            flags |= _native.STEP_FIRST
        if self.fuse_zero_grad:
            flags |= _native.STEP_ZERO_GRAD
        return dict(lr=float(lr), mu=float(mu), c_wd=float(wd), c_noise=0.0, n_train=1.0, flags=flags)

    @torch.no_grad()
    def step(self, closure=None, **kw):
        return super().step(add_langevin_noise=False, closure=closure, **kw)
