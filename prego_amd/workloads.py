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
