"""Per-frame argmax -> 200-frame majority vote -> step sequence (utils/aggregate.py:46-90), the wire format
into step_anticipation.  `aggregate` is the host (numpy) form over the JSON the eval loop wrote;
`aggregate_device` takes the int32 per-frame argmax tensors the head kernel left in HBM and runs the majority vote there
(`prego_window_vote`, csrc/postproc.hip): one int32 per 200-frame window crosses PCIe instead of one per frame."""
from __future__ import annotations

import json

import numpy as np


def _changes(a):
    r = [i for i in range(1, len(a)) if a[i] != a[i - 1]]
    r.append(len(a))
    return r


def _dedup(a):
    r = [a[0]]
    for i in range(1, len(a)):
        if a[i] != a[i - 1]:
            r.append(a[i])
    return r


def aggregate(data: dict, output_path: str | None = None, window_size: int = 200) -> dict:
    out = {}
    for key, value in data.items():
        pred = np.asarray(value["pred"])
        gt = list(value["gt"])
        new = np.zeros_like(pred)
        for s in range(0, len(pred), window_size):
            e = min(s + window_size, len(pred))
            new[s:e] = np.argmax(np.bincount(pred[s:e]))       # ties: lowest class id
        out[key] = {"pred": [int(v) for v in _dedup(list(new))], "gt": [int(v) for v in _dedup(gt)],
                    "changes_pred": _changes(list(new)), "changes_gt": _changes(gt)}
    if output_path:
        with open(output_path, "w") as fp:
            json.dump(out, fp)
    return out


def aggregate_device(preds: dict, gts: dict, output_path: str | None = None, window_size: int = 200, n_classes: int | None = None) -> dict:
    """preds: {vid: int32 CUDA tensor [T]} (per-frame argmax on the device); gts: {vid: sequence of per-frame ground-truth ids}.
    n_classes: the model's num_classes (ids outside [0, n_classes) are an error, as np.bincount's negative ids are in
    utils/aggregate.py:60); None = 128, the kernel's limit.  Same result dict as `aggregate` / utils/aggregate.py:46-90."""
    if n_classes is None:
        n_classes = 128
    import ctypes as C

    import torch

    from . import _lib
    from ._lib import PregoError, check
    from .engine import _stream_ptr
    lib = _lib.load()
    votes = {}
    for key, pred in preds.items():
        if not pred.is_cuda or pred.dtype != torch.int32 or not pred.is_contiguous() or pred.dim() != 1:
            raise PregoError(f"aggregate_device: pred[{key!r}] must be a contiguous int32 CUDA vector")
        T = pred.numel()
        v = torch.empty(((T + window_size - 1) // window_size,), dtype=torch.int32, device=pred.device)
        with torch.cuda.device(pred.device):
            check(lib.prego_window_vote(C.c_void_p(pred.data_ptr()), T, window_size, n_classes, C.c_void_p(v.data_ptr()),
                                        C.c_void_p(_stream_ptr(pred.device))))
        votes[key] = (v, T)
    out = {}
    for key, (v, T) in votes.items():
        w = v.cpu().numpy()
        if (w < 0).any():
            raise PregoError(f"aggregate_device: pred[{key!r}] holds a class id outside [0, {n_classes}) "
                             "(np.bincount of utils/aggregate.py:60 would raise / count a class the model does not have)")
        gt = list(gts[key])
        # the per-frame sequence is constant inside a window: duplicates and change points follow from the window values
        keep = np.concatenate(([True], w[1:] != w[:-1]))
        changes = [int(i) * window_size for i in np.nonzero(keep)[0][1:]] + [T]
        out[key] = {"pred": [int(x) for x in w[keep]], "gt": [int(x) for x in _dedup(gt)], "changes_pred": changes,
                    "changes_gt": _changes(gt)}
    if output_path:
        with open(output_path, "w") as fp:
            json.dump(out, fp)
    return out
