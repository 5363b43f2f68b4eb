"""Per-frame argmax -> 200-frame majority vote -> step sequence (utils/aggregate.py:46-90), the wire format
into step_anticipation.  Host side, numpy."""
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
