"""PredictionDistilled / OODDetectionDistilled — drop-ins for URSABench/tasks/prediction_distilled.py:11-145 and
ood_detection_distilled.py:11-116: the evaluation of a DISTILLED posterior, i.e. exactly two student networks,
`models = [prediction_model, expected_data_uncertainty_model]`, instead of an ensemble: the first one's softmax is
the predictive, exp() of the second one's scalar output is the expected data uncertainty. There is no ensemble
reduction here (one softmax per row), so this is not on the K5 path; it exists so that the `tasks` namespace holds every
name the reference's harness can ask for (`getattr(tasks, args.task)`, experiment.py:82). The accumulation stays on the
device; the public CPU tensors are refreshed once per update_statistics call. Quirks kept: every call counts ONE sample
whatever it holds (:41-43); the out-of-distribution half ignores the second model and takes the entropy of the first
one's smoothed softmax (ood_detection_distilled.py:82-85); PredictionDistilled.reset() leaves the uncertainty alone."""
import torch
import torch.nn.functional as F

from ..util import central_smoothing, compute_predictive_entropy
from .ood_detection import OODDetection
from .prediction import Prediction

__all__ = ['PredictionDistilled', 'OODDetectionDistilled']


def _two_students(models):
    if not isinstance(models, list):
        raise Exception('Need exactly two models here')                 # prediction_distilled.py:64
    if not all(isinstance(m, torch.nn.Module) for m in models):
        raise NotImplementedError
    if len(models) < 2:
        raise Exception('Need exactly two models here')
    return models[0].eval(), models[1].eval()


class PredictionDistilled(Prediction):
    def update_statistics(self, models, output_performance=True, smoothing=True):
        student, uncertainty = _two_students(models)
        self.num_samples_collected += 1
        dev = self.device
        proba = torch.zeros(len(self.data_loader.dataset), self.num_classes, device=dev)
        ent = torch.zeros(len(self.data_loader.dataset), device=dev)
        with torch.no_grad():
            start = 0
            for x, _ in self.data_loader:
                x = x.to(dev)
                end = start + len(x)
                proba[start:end] = F.log_softmax(student(x), dim=-1).exp_()
                ent[start:end] = uncertainty(x).exp().reshape(-1)
                start = end
        self.ensemble_proba = self.ensemble_proba + proba.cpu()
        self.expected_data_uncertainty = self.expected_data_uncertainty + ent.cpu()
        if output_performance:
            return self.get_performance_metrics(output_performance, smoothing)

    def reset(self):
        self.num_samples_collected = 0
        self.ensemble_proba = torch.zeros_like(self.ensemble_proba)


class OODDetectionDistilled(OODDetection):
    def update_statistics(self, models, output_performance=True):
        student, uncertainty = _two_students(models)
        self.num_samples_collected += 1
        dev = self.device
        with torch.no_grad():
            for loader, inside in ((self.in_distribution_loader, True), (self.out_distribution_loader, False)):
                proba = torch.zeros(len(loader.dataset), self.num_classes, device=dev)
                ent = torch.zeros(len(loader.dataset), device=dev)
                start = 0
                for x, _ in loader:
                    x = x.to(dev)
                    end = start + len(x)
                    q = central_smoothing(F.log_softmax(student(x), dim=-1).exp_())
                    proba[start:end] = q
                    ent[start:end] = uncertainty(x).exp().reshape(-1) if inside else compute_predictive_entropy(q)
                    start = end
                if inside:
                    self.in_distribution_ensemble_proba = self.in_distribution_ensemble_proba + proba.cpu()
                    self.in_distribution_data_uncertainty = self.in_distribution_data_uncertainty + ent.cpu()
                else:
                    self.out_distribution_ensemble_proba = self.out_distribution_ensemble_proba + proba.cpu()
                    self.out_distribution_data_uncertainty = self.out_distribution_data_uncertainty + ent.cpu()
        if output_performance:
            return self.get_performance_metrics()
