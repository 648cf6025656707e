"""Prediction task — drop-in for URSABench/tasks/prediction.py:12-149.

Same constructor, public state (`ensemble_proba [N,C]`, `expected_data_uncertainty [N]`,
`num_samples_collected`, `targets`) and the same 11 metrics with the same keys; the
accumulation runs on the GPU (EnsembleAccumulator) and the public CPU tensors are refreshed
once per update_statistics call. Quirks kept: the sum is over RAW softmax while the entropy is
of the SMOOTHED softmax (:60-63); reset() does not clear expected_data_uncertainty (:33-35).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .. import util
from . import metrics as M
from .task_base import EnsembleAccumulator, _Task, as_member_list

__all__ = ['Prediction']


class Prediction(_Task):
    supported_metric_list = ['error_rate', 'nll', 'll', 'brier_score', 'ece', 'misclass_model_uncertainty_auroc',
                             'misclass_model_uncertainty_aucpr', 'misclass_total_uncertainty_auroc',
                             'misclass_total_uncertainty_aucpr', 'misclass_confidence_auroc',
                             'misclass_confidence_aucpr']

    def __init__(self, dataloader, num_classes, device, metric_list, *, kernels=None, process_group=None,
                 acc_kw=None):
        super().__init__(dataloader, num_classes, device)
        self.data_loader = dataloader['in_distribution_test']
        self.num_classes = num_classes
        self.device = device
        self.process_group = process_group
        self.num_samples_collected = 0
        self._local_count = 0
        self.required_metric_list = self.supported_metric_list if metric_list == 'ALL' else metric_list
        assert all(metric in self.supported_metric_list for metric in self.required_metric_list)
        self.targets = torch.cat([y.cpu() for _, y in self.data_loader])       # prediction.py:28-31
        self._acc = EnsembleAccumulator(self.data_loader, num_classes, device, kernels, smoothed=False, **(acc_kw or {}))
        self.ensemble_proba, self.expected_data_uncertainty, _ = self._acc.local()      # zeros; no collective here

    def _publish(self):
        """The one collective of the path (see EnsembleAccumulator.reduced): update_statistics only."""
        self.ensemble_proba, self.expected_data_uncertainty, _, self.num_samples_collected = \
            self._acc.reduced(self._local_count, self.process_group)

    def reset(self):
        """prediction.py:33-35: clears the probabilities and the count, NOT expected_data_uncertainty
        (neither the public tensor nor the accumulator behind it). Local: no collective."""
        self._local_count = 0
        self.num_samples_collected = 0
        self._acc.reset(entropy_too=False)
        self.ensemble_proba = torch.zeros_like(self.ensemble_proba)

    def update_statistics(self, models, output_performance=True, smoothing=True):
        members = as_member_list(models)
        self._local_count += len(members)
        self._acc.accumulate(members)
        self._publish()
        if output_performance:
            return self.get_performance_metrics(output_performance, smoothing)

    def get_performance_metrics(self, output_performance=False, smoothing=True):
        S = self.num_samples_collected
        mean_t = self.ensemble_proba / S
        mean = mean_t.numpy()
        tgt = self.targets.numpy()
        smooth = util.central_smoothing(mean_t).numpy()
        edu = (self.expected_data_uncertainty / S).numpy()
        out = {}
        for metric in self.required_metric_list:
            if metric == 'error_rate':
                out[metric] = 1 - np.mean(np.argmax(mean, axis=1) == tgt)
            elif metric in ('nll', 'll'):
                logp = torch.log(util.central_smoothing(mean_t) if smoothing else mean_t)
                nll = F.nll_loss(logp, self.targets).item()
                out[metric] = -nll if metric == 'll' else nll
            elif metric == 'brier_score':
                out[metric] = M.brier(mean, tgt)
            elif metric == 'ece':
                out[metric] = M.ece(mean, tgt)
            else:                                    # misclass_{model_uncertainty,total_uncertainty,confidence}_{auroc,aucpr}
                crit = {'model': 'model_uncertainty', 'total': 'entropy', 'confidence': 'confidence'}[metric.split('_')[1]]
                wrong, score = M.misclassification_scores(smooth, tgt, crit, edu)
                out[metric] = (M.roc_auc if metric.endswith('auroc') else M.average_precision)(wrong, score)
        if output_performance:
            if len(self.required_metric_list) != 1:
                raise RuntimeError('Multiple metrics in metric list not suitable for output_performance = True')
            return float(out[self.required_metric_list[0]])
        return out
