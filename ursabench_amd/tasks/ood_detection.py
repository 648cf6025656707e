"""OODDetection — drop-in for URSABench/tasks/ood_detection.py:11-130. Accumulates the SMOOTHED
softmax and its entropy on an in-distribution and an out-of-distribution loader; AUROC of
total / model uncertainty with label 1 = OOD (:118-125)."""
import numpy as np
import torch

from ..util import compute_predictive_entropy
from . import metrics as M
from .task_base import EnsembleAccumulator, _Task, as_member_list

__all__ = ['OODDetection']


class OODDetection(_Task):
    def __init__(self, data_loader=None, num_classes=None, device=torch.device('cpu'), *, kernels=None,
                 process_group=None, acc_kw=None):
        super().__init__(data_loader, num_classes, device)
        self.in_distribution_loader = data_loader['in_distribution_test']
        self.out_distribution_loader = data_loader['out_distribution_test']
        self.num_classes = num_classes
        self.device = device
        self.process_group = process_group
        self._in = EnsembleAccumulator(self.in_distribution_loader, num_classes, device, kernels, smoothed=True, **(acc_kw or {}))
        self._out = EnsembleAccumulator(self.out_distribution_loader, num_classes, device, kernels, smoothed=True, **(acc_kw or {}))
        self.reset()

    def _publish(self):
        """Collectives (one per loader) happen here, i.e. in update_statistics only."""
        p, e, _, n = self._in.reduced(self._local_count, self.process_group)
        self.in_distribution_ensemble_proba, self.in_distribution_data_uncertainty = p, e
        p, e, _, _ = self._out.reduced(self._local_count, self.process_group)
        self.out_distribution_ensemble_proba, self.out_distribution_data_uncertainty = p, e
        self.num_samples_collected = n

    def reset(self):
        self._in.reset()
        self._out.reset()
        self._local_count = 0
        self.in_distribution_total_uncertainty = None
        self.out_distribution_total_uncertainty = None
        self.in_distribution_model_uncertainty = None
        self.out_distribution_model_uncertainty = None
        self.num_samples_collected = 0
        self.in_distribution_ensemble_proba, self.in_distribution_data_uncertainty, _ = self._in.local()
        self.out_distribution_ensemble_proba, self.out_distribution_data_uncertainty, _ = self._out.local()

    def update_statistics(self, models, output_performance=True):
        members = as_member_list(models)
        self._local_count += len(members)
        self._in.accumulate(members)
        self._out.accumulate(members)
        self._publish()
        if output_performance:
            return self.get_performance_metrics()

    def get_performance_metrics(self):
        S = self.num_samples_collected
        self.in_distribution_total_uncertainty = compute_predictive_entropy(self.in_distribution_ensemble_proba / S)
        self.out_distribution_total_uncertainty = compute_predictive_entropy(self.out_distribution_ensemble_proba / S)
        self.in_distribution_model_uncertainty = \
            self.in_distribution_total_uncertainty - self.in_distribution_data_uncertainty / S
        self.out_distribution_model_uncertainty = \
            self.out_distribution_total_uncertainty - self.out_distribution_data_uncertainty / S
        labels = np.concatenate([np.ones(len(self.out_distribution_loader.dataset)),
                                 np.zeros(len(self.in_distribution_loader.dataset))])
        total = np.concatenate([self.out_distribution_total_uncertainty.numpy(),
                                self.in_distribution_total_uncertainty.numpy()])
        model = np.concatenate([self.out_distribution_model_uncertainty.numpy(),
                                self.in_distribution_model_uncertainty.numpy()])
        return {'total_uncertainty_auroc': M.roc_auc(labels, total), 'model_uncertainty_auroc': M.roc_auc(labels, model)}
