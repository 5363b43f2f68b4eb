"""Per-frame average precision, the metric behind `metric: 'AP'` of both shipped configs (the reference computes it in
step_recognition/utils/metrics.py:25-62 with one sklearn call per class), SURVEY.md section 8 row f3.

The definition is sklearn's `average_precision_score`: per class, thresholds at the DISTINCT score values in descending order
(ties share a threshold), AP = sum_k (R_k - R_{k-1}) P_k; classes without a positive are skipped and so is class 0, which the
reference always treats as background (metrics.py:44-48).  Two routes to the same numbers:

  * `perframe_average_precision`         - numpy on the host, all classes at once (one argsort of the score matrix); the CPU
                                           tests hold it to sklearn.
  * `perframe_average_precision_device`  - the HIP path: `prego_perframe_ap` / `prego_perframe_ap_labels` (csrc/metrics.hip: the
                                           positives of every class sorted, every score counted against them), used by `Evaluate`
                                           so that the [frames x classes] score matrix of an eval pass never leaves the device.
                                           No CPU fallback.

`metric: 'cAP'` (the reference's calibrated variant for TVSeries, metrics.py:10-22; unreachable from the shipped configs) runs on the
host path only; `Evaluate` moves the matrices to the host for it."""
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


def calibrated_average_precision_columns(scores: np.ndarray, positive: np.ndarray) -> np.ndarray:
    """cAP of every column (metrics.py:10-22 of the reference, all classes at once): precision re-weighted by the negative /
    positive ratio w, cprec_k = TP_k / (TP_k + FP_k / (w + eps) + eps), averaged over the ranks that hold a positive.  Rows of equal
    score keep their input order (stable sort; the reference's unstable argsort leaves ties to chance)."""
    scores = np.asarray(scores)
    positive = np.asarray(positive, dtype=bool)
    eps = np.finfo(float).eps
    order = np.argsort(-scores, axis=0, kind="stable")
    tp = np.take_along_axis(positive, order, axis=0).astype(np.float64)
    tps = np.cumsum(tp, axis=0)
    fps = np.cumsum(1.0 - tp, axis=0)
    total = tp.sum(axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        ratio = (tp.shape[0] - total) / total
        cprec = tps / (tps + fps / (ratio + eps) + eps)
        cap = (cprec * tp).sum(axis=0) / total
    return np.where(total > 0, cap, np.nan)


def _report(ap, n_true, score_sum, class_names):
    """the reference's result dict: per-class AP (classes 1.. that have a positive), its log strings, and their mean"""
    res = OrderedDict(per_class_AP=OrderedDict(), num=OrderedDict())
    for c in range(1, len(class_names)):                    # class 0 = background, never scored (metrics.py:44-48)
        if n_true[c] > 0:
            name = class_names[c]
            res["per_class_AP"][name] = float(ap[c])
            res["num"][name] = f"[true: {int(n_true[c])}, pred:{int(score_sum[c])}, AP:{ap[c] * 100:.1f}]"
    with np.errstate(invalid="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            res["mean_AP"] = np.mean(list(res["per_class_AP"].values())) if res["per_class_AP"] else float("nan")
    return res


def perframe_ap_raw(pred: np.ndarray, truth: np.ndarray, metrics="AP"):
    """(AP or cAP per column, positives per column, score mass per column) of host matrices [frames, classes]: the three vectors the
    report is built from (class-sharded multi-rank eval computes them for its own columns and gathers only these)"""
    cols = average_precision_columns if metrics == "AP" else calibrated_average_precision_columns
    if pred.shape[0] == 0:
        z = np.zeros(pred.shape[1])
        return z, z.astype(np.int64), z.copy()
    return cols(pred, truth != 0), (truth != 0).sum(0), pred.sum(0, dtype=np.float64)


def report_from_raw(ap, n_true, score_sum, class_names):
    return _report(np.asarray(ap), np.asarray(n_true), np.asarray(score_sum), class_names)


def perframe_average_precision(prediction, ground_truth, class_names, postprocessing=None, metrics="AP"):
    """Host path: prediction / ground_truth [frames, classes] (lists of rows or arrays, as trainer/eval.py collects them)."""
    if metrics not in ("AP", "cAP"):
        raise RuntimeError(f"Unknown metrics: {metrics}")
    truth = np.asarray(ground_truth)
    pred = np.asarray(prediction)
    if postprocessing is not None:
        truth, pred = postprocessing(truth, pred)
    if truth.ndim != 2 or truth.shape[0] == 0:          # an empty eval set: the reference's np.mean([]) = nan, no per-class entries
        return _report(np.zeros(len(class_names)), np.zeros(len(class_names), np.int64), np.zeros(len(class_names)), class_names)
    cols = average_precision_columns if metrics == "AP" else calibrated_average_precision_columns
    return _report(cols(pred, truth != 0), (truth != 0).sum(0), pred.sum(0, dtype=np.float64), class_names)


def perframe_ap_raw_device(pred, truth):
    """device counterpart of perframe_ap_raw (metric 'AP'): fp32 CUDA matrices [frames, classes] (truth: or class ids [frames]) -> three host vectors"""
    ncls = int(pred.shape[1])
    fin = perframe_average_precision_device(pred, truth, [str(i) for i in range(ncls)], None, "AP", defer=True, raw=True)
    return fin()


def perframe_average_precision_device(prediction, ground_truth, class_names, postprocessing=None, metrics="AP", defer=False, raw=False):
    """Device path: prediction / ground_truth fp32 CUDA tensors [frames, classes] - or ground_truth = an integer CUDA tensor [frames],
    one class id per frame (what one-hot targets say; an id outside the classes = no positive) -; the sort and the scan run in
    libprego_amd.so (`prego_perframe_ap`), one small device -> host transfer brings back AP, positives and score mass.
    defer=True: the kernels are only ENQUEUED and a function is returned that waits for them and builds the report (Evaluate writes
    its output file while the GPU sorts)."""
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
    by_label = ground_truth.dim() == 1 and not ground_truth.dtype.is_floating_point      # one class id per frame (prego_perframe_ap_labels)
    truth = ground_truth.detach().to(torch.int32 if by_label else torch.float32).contiguous()
    if pred.dim() != 2 or pred.shape[1] != len(class_names) or (truth.shape != pred.shape[:1] if by_label else truth.shape != pred.shape):
        raise PregoError(f"perframe_average_precision_device: shapes {tuple(pred.shape)} / {tuple(truth.shape)} for {len(class_names)} classes")
    n, ncls = pred.shape
    if n == 0:                                          # an empty eval set: same empty report as the host path
        z = (np.zeros(ncls), np.zeros(ncls, np.int64), np.zeros(ncls))
        rep = z if raw else _report(*z, class_names)
        return (lambda: rep) if defer else rep
    lib = _lib.load()
    dev = pred.device
    ws = torch.empty(lib.prego_perframe_ap_workspace_bytes(n, ncls), dtype=torch.uint8, device=dev)
    out = torch.empty((3, ncls), dtype=torch.float64, device=dev)          # AP | positives (int64 bits) | score sums
    with torch.cuda.device(dev):
        fn = lib.prego_perframe_ap_labels if by_label else lib.prego_perframe_ap
        check(fn(C.c_void_p(pred.data_ptr()), C.c_void_p(truth.data_ptr()), n, ncls, C.c_void_p(out[0].data_ptr()),
                 C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    def finish(_keep=(pred, truth, ws)):
        host = out.cpu().numpy()
        if raw:
            return host[0].copy(), host[1].view(np.int64).copy(), host[2].copy()
        return _report(host[0], host[1].view(np.int64), host[2], class_names)
    return finish if defer else finish()
