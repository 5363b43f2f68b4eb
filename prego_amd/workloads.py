"""Synthetic workloads shaped like BASELINE.json's configs (no datasets ship with the reference)."""
from __future__ import annotations

import numpy as np

from . import weights as W

# lengths of the 15 Epic-tent-O test videos, read off the reference's shipped eval dump
# (output_miniRoad/output_miniROAD.json: 187 959 frames) - the only real length statistics in the repo
EPIC_TENT_TEST_LENGTHS = [3702, 5157, 6971, 9015, 9874, 10046, 10735, 11734, 11744, 12807, 12976, 13001, 17280,
                          21803, 31114]
ASSEMBLY101_TEST_CLIPS = 182     # step_recognition/data_info/video_list.json, ASSEMBLY101-O test_session_set


def assembly101_eval_lengths(n_clips: int = ASSEMBLY101_TEST_CLIPS, seed: int = 20) -> list:
    """Assembly101-O test-split-sized clip list.  The real lengths are unknown (features are not shipped),
    so lengths are drawn once, seeded, from the Epic-tent-O length distribution with +-10 % jitter
    (SURVEY.md section 8d says to do exactly this, and to say so)."""
    u = W.uniform01((n_clips, 2), seed, "assembly101.lengths")
    base = np.array(EPIC_TENT_TEST_LENGTHS)[(u[:, 0] * len(EPIC_TENT_TEST_LENGTHS)).astype(int)]
    return [int(round(b * (0.9 + 0.2 * j))) for b, j in zip(base, u[:, 1])]


# ---- synthetic *learnable* action videos (fixture G11: parity on TRAINED weights) --------------------------------------------
# The uniform-noise features above carry no signal, so nothing can be trained on them.  These videos have one prototype per
# class under unit noise (TSN-like: post-ReLU, non-negative), piecewise-constant label tracks with a background class 0
# (`utils/metrics.py:48` ignores it) and a share of degraded frames, so that a trained model shows the whole range from flat to
# saturated softmax rows.  Everything is a function of (seed, name): the build container trains the imported reference on them
# (`oracle/train_g11.py`), the GPU box regenerates the same test videos for the parity tests.

def action_prototypes(num_classes: int, seed: int = 20, dim: int = 2048) -> np.ndarray:
    return W.normal((num_classes, dim), seed, f"actions.proto.{num_classes}.{dim}")


def action_labels(T: int, num_classes: int, seed: int, name: str, max_extra: int = 600) -> np.ndarray:
    """Piecewise-constant class ids [T]: segments of 24 .. 24 + max_extra frames (short ones more likely), a quarter of them background."""
    n_seg = T // 24 + 2
    u = W.uniform01((n_seg, 3), seed, name + ".seg")
    labels = np.empty(T, np.int64)
    t = i = 0
    while t < T:
        dur = 24 + int(u[i, 0] * u[i, 0] * max_extra)
        labels[t:t + dur] = 0 if u[i, 1] < 0.25 else 1 + int(u[i, 2] * (num_classes - 1))
        t += dur
        i += 1
    return labels


def action_video(T: int, num_classes: int, seed: int, name: str, snr: float = 0.15, max_extra: int = 600):
    """(rgb f32 [T, 2048] >= 0, labels int64 [T]).  30 % of the frames carry a weakened prototype (amplitude uniform in [0, snr))."""
    labels = action_labels(T, num_classes, seed, name, max_extra)
    P = action_prototypes(num_classes)
    q = W.uniform01((T,), seed, name + ".q")
    amp = snr * np.where(q < 0.3, q / 0.3, 1.0)
    x = W.normal((T, 2048), seed, name + ".noise") + (amp[:, None] * P[labels]).astype(np.float32)
    return np.maximum(x, 0.0).astype(np.float32), labels


def onehot(labels: np.ndarray, num_classes: int) -> np.ndarray:
    t = np.zeros((labels.shape[0], num_classes), np.float32)
    t[np.arange(labels.shape[0]), labels] = 1.0
    return t
