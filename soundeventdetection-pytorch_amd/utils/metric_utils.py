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


def metric_counts_device(output, target, raw_logits=False, return_probs=False):
    """Device-side counting behind calculate_metrics: `output` (frames, K) and `target` (frames', K)
    are CUDA tensors; returns (tp[21], positives[21], gt_sum[, probs]) as torch tensors still on
    the device (sed_metric_counts; no host copy of the frame probabilities)."""
    import torch
    from .. import _lib as L
    if not (output.is_cuda and target.is_cuda):
        raise RuntimeError("metric_counts_device needs CUDA tensors (there is no CPU path)")
    out = output.detach().float().contiguous()
    tgt = target.detach().float().contiguous()
    if out.dim() != 2 or tgt.dim() != 2 or out.shape[1] != tgt.shape[1]:
        raise ValueError(f"expected (frames, K) tensors, got {tuple(out.shape)} / {tuple(tgt.shape)}")
    nth = len(THRESHOLDS)
    n = min(out.shape[0], tgt.shape[0])
    dev = out.device
    counts = torch.zeros(nth, 2, dtype=torch.int64, device=dev)       # uint64 on the device side; < 2^63
    gt = torch.zeros(1, dtype=torch.float64, device=dev)
    ws = torch.empty(L.lib().sed_metric_counts_ws_bytes(nth) // 8 + 1, dtype=torch.float64, device=dev)
    probs = torch.empty(n, out.shape[1], dtype=torch.float32, device=dev) if return_probs else None
    import ctypes
    ths = (ctypes.c_double * nth)(*[float(t) for t in THRESHOLDS])
    L.check(L.lib().sed_metric_counts(L.ptr(out), L.ptr(tgt), L.ptr(probs), ctypes.cast(ths, ctypes.c_void_p), nth,
                                      1 if raw_logits else 0, L.ptr(counts), L.ptr(gt), L.ptr(ws),
                                      out.shape[0], tgt.shape[0], out.shape[1],
                                      torch.cuda.current_stream().cuda_stream), "metric_counts")
    res = (counts[:, 0], counts[:, 1], gt)
    return res + (probs,) if return_probs else res


def metrics_from_counts(tp, positives, gt_sum):
    """compute_recall_precision's divisions + the AP sum on the 21 (TP, positives) pairs."""
    tp = np.asarray(tp, dtype=np.int64)
    positives = np.asarray(positives, dtype=np.int64)
    num_gt = float(gt_sum)
    recalls = np.array([float(t) / num_gt if num_gt > 0 else 1 for t in tp], dtype=np.float64)
    precisions = np.array([float(t) / float(p) if p > 0 else 1 for t, p in zip(tp, positives)], dtype=np.float64)
    AP = np.sum(precisions[:-1] * (recalls[:-1] - recalls[1:]))
    return recalls, precisions, AP


def calculate_metrics_device(output, target, raw_logits=False):
    """calculate_metrics for CUDA tensors: counting on the device, 43 numbers to the host."""
    tp, pos, gt = metric_counts_device(output, target, raw_logits=raw_logits)
    import torch
    host = torch.cat([tp.double(), pos.double(), gt]).cpu().numpy()
    n = len(THRESHOLDS)
    return metrics_from_counts(host[:n].astype(np.int64), host[n:2 * n].astype(np.int64), host[2 * n])
