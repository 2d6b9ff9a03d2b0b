"""Counterpart of /root/reference/utils/metric_utils.py:4-37 (host-side numpy; semantics frozen:
21 thresholds arange(0,1.05,.05), strict `>`, TP = ((2T-O)==1), recall=1 without ground truth,
precision=1 without positives, AP = sum P[i]*(R[i]-R[i+1]))."""
import numpy as np

THRESHOLDS = np.arange(0.00, 1.05, 0.05)


def compute_recall_precision(O, T):
    O = np.asarray(O)
    T = np.asarray(T)
    tp = np.count_nonzero((2 * T - O) == 1)
    num_gt = T.sum()
    num_pos = O.sum()
    recall = float(tp) / float(num_gt) if num_gt > 0 else 1
    prec = float(tp) / float(num_pos) if num_pos > 0 else 1
    return recall, prec


def calculate_metrics(output, target):
    n = min(output.shape[0], target.shape[0])
    T, O = target[:n], output[:n]
    rp = [compute_recall_precision(np.where(O > th, 1, 0), T) for th in THRESHOLDS]
    recalls = np.array([r for r, _ in rp])
    precisions = np.array([p for _, p in rp])
    AP = np.sum(precisions[:-1] * (recalls[:-1] - recalls[1:]))
    return recalls, precisions, AP


def f_score(recll, precision, precision_importance_factor=1):
    b2 = precision_importance_factor ** 2
    return (1 + b2) * recll * precision / (b2 * recll + precision + 1e-9)
