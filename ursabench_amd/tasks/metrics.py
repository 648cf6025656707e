"""The metric surface of URSABench/tasks/prediction.py:79-267, computed from the accumulators
the gfx950 reduction kernel produced. Own implementations (numpy only; scikit-learn is not a
dependency): AUROC and average precision follow sklearn.metrics.roc_auc_score /
average_precision_score semantics (ties share a threshold), which is what the reference calls
(prediction.py:4,243,267; ood_detection.py:4,124-125)."""
import numpy as np


def roc_auc(labels, scores):
    """Area under the ROC curve; equals the tie-corrected Mann-Whitney statistic. NaN when only
    one class is present (sklearn raises/warns there)."""
    labels = np.asarray(labels).astype(bool)
    scores = np.asarray(scores, dtype=np.float64)
    n_pos, n_neg = int(labels.sum()), int((~labels).sum())
    if n_pos == 0 or n_neg == 0:
        return float('nan')
    order = np.argsort(-scores, kind='mergesort')
    s, l = scores[order], labels[order]
    last = np.r_[np.nonzero(np.diff(s))[0], s.size - 1]            # last index of each distinct threshold
    tps = np.r_[0.0, np.cumsum(l)[last]]
    fps = np.r_[0.0, np.cumsum(~l)[last]]
    tpr, fpr = tps / n_pos, fps / n_neg
    return float(np.sum(np.diff(fpr) * (tpr[1:] + tpr[:-1]) * 0.5))


def average_precision(labels, scores):
    """sum_k (R_k - R_{k-1}) P_k over distinct thresholds, descending score."""
    labels = np.asarray(labels).astype(bool)
    scores = np.asarray(scores, dtype=np.float64)
    n_pos = int(labels.sum())
    if n_pos == 0:
        return float('nan')
    order = np.argsort(-scores, kind='mergesort')
    s, l = scores[order], labels[order]
    last = np.r_[np.nonzero(np.diff(s))[0], s.size - 1]
    tps = np.cumsum(l)[last].astype(np.float64)
    fps = np.cumsum(~l)[last].astype(np.float64)
    precision = tps / (tps + fps)
    recall = tps / n_pos
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def ece(preds, targets, n_bins=15):
    """prediction.py:152-182 — 15 equal-width confidence bins, (lower, upper]."""
    edges = np.linspace(0, 1, n_bins + 1)
    conf, pred = preds.max(1), preds.argmax(1)
    hit = pred == targets
    total = 0.0
    for lo, hi in zip(edges[:-1], edges[1:]):
        inb = np.logical_and(conf > lo, conf <= hi)
        frac = np.mean(inb)
        if frac > 0:
            total += np.abs(np.mean(conf[inb]) - np.mean(hit[inb])) * frac
    return total


def brier(preds, targets):
    """prediction.py:185-194 — float64 one-hot."""
    onehot = np.zeros(preds.shape)
    onehot[np.arange(len(targets)), targets] = 1.0
    return np.mean(np.sum((preds - onehot) ** 2, axis=1))


def misclassification_scores(preds, targets, criterion, expected_data_uncertainty=None):
    """prediction.py:222-243 — (is_misclassified, score) for the three criteria."""
    wrong = preds.argmax(1) != np.asarray(targets)       # top-1 (prediction.py:197-219 with topk=1)
    if criterion == 'entropy':
        score = np.sum(-preds * np.log(preds), axis=1)
    elif criterion == 'confidence':
        score = -preds.max(axis=1)
    elif criterion == 'model_uncertainty':
        score = np.sum(-preds * np.log(preds), axis=1) - expected_data_uncertainty
    else:
        raise NotImplementedError
    return wrong, score
