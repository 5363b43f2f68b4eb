"""Per-frame average precision, the metric behind `metric: 'AP'` of both shipped configs (the reference computes it in
step_recognition/utils/metrics.py:25-62 with one sklearn call per class), SURVEY.md section 8 row f3.

The definition is sklearn's `average_precision_score`: per class, thresholds at the DISTINCT score values in descending order
(ties share a threshold), AP = sum_k (R_k - R_{k-1}) P_k; classes without a positive are skipped and so is class 0, which the
reference always treats as background (metrics.py:44-48).  Two routes to the same numbers:

  * `perframe_average_precision`         - numpy on the host, all classes at once (one argsort of the score matrix); the CPU
                                           tests hold it to sklearn.
  * `perframe_average_precision_device`  - the HIP path: `prego_perframe_ap` (csrc/metrics.hip: segmented radix sort + scan, one
                                           segment per class), used by `Evaluate` so that the [frames x classes] score matrix of
                                           an eval pass never leaves the device.  No CPU fallback.

Only `metric: 'AP'` exists here: the reference's calibrated variant (cAP) is unreachable from the shipped configs."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


def average_precision_columns(scores: np.ndarray, positive: np.ndarray) -> np.ndarray:
    """AP of every column: scores [frames, classes] (any float dtype), positive [frames, classes] bool -> float64 [classes],
    NaN where a column has no positive."""
    scores = np.asarray(scores)
    positive = np.asarray(positive, dtype=bool)
    n = scores.shape[0]
    order = np.argsort(-scores, axis=0, kind="stable")
    ranked = np.take_along_axis(scores, order, axis=0)
    hits = np.cumsum(np.take_along_axis(positive, order, axis=0), axis=0, dtype=np.int64)      # positives at or above each rank
    closes = np.ones_like(positive)                                                            # rank closes a run of equal scores
    closes[:-1] = ranked[:-1] != ranked[1:]
    # positives at the previous threshold: the running maximum of `hits` over the earlier run ends (hits is monotone)
    at_end = np.where(closes, hits, 0)
    before = np.zeros_like(hits)
    before[1:] = np.maximum.accumulate(at_end, axis=0)[:-1]
    seen = np.arange(1, n + 1, dtype=np.float64)[:, None]
    total = hits[-1].astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        ap = np.where(closes, (hits - before) * (hits / seen), 0.0).sum(axis=0) / total
    return np.where(total > 0, ap, np.nan)


def _report(ap, n_true, score_sum, class_names):
    """the reference's result dict: per-class AP (classes 1.. that have a positive), its log strings, and their mean"""
    res = OrderedDict(per_class_AP=OrderedDict(), num=OrderedDict())
    for c in range(1, len(class_names)):                    # class 0 = background, never scored (metrics.py:44-48)
        if n_true[c] > 0:
            name = class_names[c]
            res["per_class_AP"][name] = float(ap[c])
            res["num"][name] = f"[true: {int(n_true[c])}, pred:{int(score_sum[c])}, AP:{ap[c] * 100:.1f}]"
    res["mean_AP"] = np.mean(list(res["per_class_AP"].values()))
    return res


def perframe_average_precision(prediction, ground_truth, class_names, postprocessing=None, metrics="AP"):
    """Host path: prediction / ground_truth [frames, classes] (lists of rows or arrays, as trainer/eval.py collects them)."""
    if metrics != "AP":
        raise RuntimeError(f"Unknown metrics: {metrics} (prego_amd implements 'AP', the metric of the shipped configs)")
    truth = np.asarray(ground_truth)
    pred = np.asarray(prediction)
    if postprocessing is not None:
        truth, pred = postprocessing(truth, pred)
    return _report(average_precision_columns(pred, truth != 0), (truth != 0).sum(0), pred.sum(0, dtype=np.float64), class_names)


def perframe_average_precision_device(prediction, ground_truth, class_names, postprocessing=None, metrics="AP"):
    """Device path: prediction / ground_truth fp32 CUDA tensors [frames, classes]; the sort and the scan run in
    libprego_amd.so (`prego_perframe_ap`), one small device -> host transfer brings back AP, positives and score mass."""
    import ctypes as C

    import torch

    from . import _lib
    from ._lib import PregoError, check
    if postprocessing is not None or metrics != "AP":
        raise PregoError("perframe_average_precision_device: metric 'AP' without postprocessing only (what both shipped configs use)")
    if not (prediction.is_cuda and ground_truth.is_cuda):
        raise PregoError("perframe_average_precision_device needs CUDA tensors; there is no CPU fallback "
                         "(host arrays: perframe_average_precision)")
    pred = prediction.detach().to(torch.float32).contiguous()
    truth = ground_truth.detach().to(torch.float32).contiguous()
    if pred.dim() != 2 or pred.shape != truth.shape or pred.shape[1] != len(class_names):
        raise PregoError(f"perframe_average_precision_device: shapes {tuple(pred.shape)} / {tuple(truth.shape)} for {len(class_names)} classes")
    n, ncls = pred.shape
    lib = _lib.load()
    dev = pred.device
    ws = torch.empty(lib.prego_perframe_ap_workspace_bytes(n, ncls), dtype=torch.uint8, device=dev)
    out = torch.empty((3, ncls), dtype=torch.float64, device=dev)          # AP | positives (int64 bits) | score sums
    with torch.cuda.device(dev):
        check(lib.prego_perframe_ap(C.c_void_p(pred.data_ptr()), C.c_void_p(truth.data_ptr()), n, ncls, C.c_void_p(out[0].data_ptr()),
                                    C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    host = out.cpu().numpy()
    return _report(host[0], host[1].view(np.int64), host[2], class_names)
