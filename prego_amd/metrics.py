"""Per-frame average precision, host side (step_recognition/utils/metrics.py:25-62).
Not on the frames/s path; kept on the CPU with sklearn exactly like the
reference, including its quirk of always ignoring class index 0 as "background" (metrics.py:48)."""
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
