"""Per-frame average precision (step_recognition/utils/metrics.py:25-62), SURVEY.md section 8 f3.

Two implementations of the same definition, including the reference's quirk of always ignoring class index 0 as
"background" (metrics.py:48):
  * `perframe_average_precision`        - host side, sklearn, exactly like the reference (numpy inputs);
  * `perframe_average_precision_torch`  - the same arithmetic restated on torch tensors (sort + scans per class, float64),
    used by `Evaluate` so that the [frames x classes] score matrix of an eval pass never leaves the device: at the
    Assembly101-O test split's size (2.3 M frames x 86 classes) the sklearn path takes tens of seconds after a 0.13 s
    forward pass.  It follows sklearn's `average_precision_score` = sum over DISTINCT thresholds of (R_k - R_{k-1}) P_k
    (ties in the scores share one threshold), so it agrees with the host version to float64 rounding."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


def calibrated_average_precision_score(y_true, y_score):
    """cAP (metrics.py:10-22)."""
    y_true_sorted = y_true[np.argsort(-y_score)]
    tp = y_true_sorted.astype(float)
    fp = np.abs(y_true_sorted.astype(float) - 1)
    tps, fps = np.cumsum(tp), np.cumsum(fp)
    ratio = np.sum(tp == 0) / np.sum(tp)
    cprec = tps / (tps + fps / (ratio + np.finfo(float).eps) + np.finfo(float).eps)
    return np.sum(cprec[tp == 1]) / np.sum(tp)


def perframe_average_precision(prediction, ground_truth, class_names, postprocessing=None, metrics="AP"):
    from sklearn.metrics import average_precision_score

    result = OrderedDict()
    ground_truth = np.asarray(ground_truth)
    prediction = np.asarray(prediction)
    if postprocessing is not None:
        ground_truth, prediction = postprocessing(ground_truth, prediction)
    if metrics == "AP":
        compute_score = average_precision_score
    elif metrics == "cAP":
        compute_score = calibrated_average_precision_score
    else:
        raise RuntimeError("Unknown metrics: {}".format(metrics))
    ignore_index = {0}
    result["per_class_AP"] = OrderedDict()
    result["num"] = OrderedDict()
    for idx, class_name in enumerate(class_names):
        if idx not in ignore_index and np.any(ground_truth[:, idx]):
            ap = compute_score(ground_truth[:, idx], prediction[:, idx])
            result["per_class_AP"][class_name] = ap
            result["num"][class_name] = (f"[true: {int(np.sum(ground_truth[:, idx]))}, "
                                         f"pred:{int(np.sum(prediction[:, idx]))}, AP:{ap * 100:.1f}]")
    result["mean_AP"] = np.mean(list(result["per_class_AP"].values()))
    return result


def average_precision_torch(y_true, y_score):
    """sklearn.metrics.average_precision_score for one class on torch tensors (any device), float64."""
    import torch
    y_score = y_score.to(torch.float64)
    s, idx = torch.sort(y_score, descending=True, stable=True)
    y = y_true[idx].to(torch.float64)
    n = y.numel()
    tps_all = torch.cumsum(y, 0)
    last = torch.ones(n, dtype=torch.bool, device=y.device)
    if n > 1:
        last[:-1] = s[:-1] != s[1:]                       # end of every run of equal scores = one threshold
    tps = tps_all[last]
    cnt = torch.arange(1, n + 1, device=y.device, dtype=torch.float64)[last]
    precision = tps / cnt
    recall = tps / tps_all[-1]
    prev = torch.cat([torch.zeros(1, dtype=torch.float64, device=y.device), recall[:-1]])
    return float(((recall - prev) * precision).sum())


def average_precision_columns_torch(y_true, y_score):
    """sklearn.metrics.average_precision_score for EVERY class at once: y_true (bool) and y_score [frames, classes] on any
    device -> float64 [classes] (NaN where a class has no positive).  One stable sort along the frame axis, cumulative sums and
    a running maximum: no per-class Python loop and a single host transfer for the whole metric.  Thresholds are the ends of
    runs of equal scores (ties share one threshold), exactly as sklearn's precision_recall_curve builds them."""
    import torch
    s, idx = torch.sort(y_score.to(torch.float64), dim=0, descending=True, stable=True)
    y = torch.gather(y_true.to(torch.float64), 0, idx)
    n = y.shape[0]
    tps = torch.cumsum(y, 0)
    last = torch.ones_like(y, dtype=torch.bool)
    if n > 1:
        last[:-1] = s[:-1] != s[1:]
    cnt = torch.arange(1, n + 1, device=y.device, dtype=torch.float64)[:, None]
    total = tps[-1:].clamp(min=1e-300)
    recall = tps / total
    precision = tps / cnt
    zero = torch.zeros((), dtype=torch.float64, device=y.device)
    r_last = torch.where(last, recall, zero)
    prev = torch.cummax(r_last, 0).values                      # recall at the most recent threshold at or before i (recall is monotone)
    prev = torch.cat([torch.zeros_like(prev[:1]), prev[:-1]], 0)
    ap = torch.where(last, (recall - prev) * precision, zero).sum(0)
    return torch.where(tps[-1] > 0, ap, torch.full_like(ap, float("nan")))


def perframe_average_precision_torch(prediction, ground_truth, class_names, postprocessing=None, metrics="AP"):
    """`perframe_average_precision` on torch tensors [frames, classes] (prediction: scores, ground_truth: one/multi-hot)."""
    import torch
    if postprocessing is not None or metrics != "AP":
        return perframe_average_precision(prediction.cpu().numpy(), ground_truth.cpu().numpy(), class_names, postprocessing, metrics)
    result = OrderedDict()
    result["per_class_AP"] = OrderedDict()
    result["num"] = OrderedDict()
    truth = ground_truth != 0
    # everything the report needs in ONE device -> host transfer: AP, positives and score mass per class
    stats = torch.stack([average_precision_columns_torch(truth, prediction), truth.sum(0).to(torch.float64),
                         prediction.to(torch.float64).sum(0), ground_truth.to(torch.float64).sum(0)]).cpu().numpy()
    for idx, class_name in enumerate(class_names):
        if idx != 0 and stats[1, idx] > 0:                        # class 0 is skipped as "background" (utils/metrics.py:48)
            ap = float(stats[0, idx])
            result["per_class_AP"][class_name] = ap
            result["num"][class_name] = (f"[true: {int(stats[3, idx])}, pred:{int(stats[2, idx])}, AP:{ap * 100:.1f}]")
    result["mean_AP"] = np.mean(list(result["per_class_AP"].values()))
    return result
