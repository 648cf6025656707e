"""The sampler plug-in API — same surface as URSABench/inference/inference_base.py:12-56."""
import torch

from ..util import get_loss_criterion


class _Inference:
    """Base class of inference wrapper (inference_base.py:12-44)."""

    def __init__(self, hyperparameters, model=None, train_loader=None, device=torch.device('cpu'),
                 model_loss='multi_class_linear_output'):
        self.model = model
        self.hyperparameters = hyperparameters
        self.train_loader = train_loader
        self.device = device
        self.loss_criterion = get_loss_criterion(loss=model_loss)

    def update_hyp(self, hyperparameters):
        raise NotImplementedError

    def sample_iterative(self):
        raise NotImplementedError

    def sample(self):
        raise NotImplementedError

    # north_star calls the sampling entry point `sample_theta`; no such symbol exists in the
    # reference — kept as an alias of sample().
    def sample_theta(self, *args, **kwargs):
        return self.sample(*args, **kwargs)

    def compute_val_loss(self, val_loader=None):
        """inference_base.py:46-56 — mean CE over a loader in eval mode. The per-batch `.item()`
        of the reference becomes one sync at the end (loss summed on the device)."""
        with torch.no_grad():
            n, total = 0, None
            self.model.eval()
            for x, y in val_loader:
                x = x.to(self.device)
                l = self.loss_criterion(self.model(x), y.to(self.device)) * len(x)
                total = l if total is None else total + l
                n += len(x)
            return total.item() / n
