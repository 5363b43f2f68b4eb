"""Host-side cost model of a MiniROAD pass / training step on one MI355X, and what it predicts for 1 / 2 / 4 / 8 GPUs.

Pure Python (no GPU, no library): `bench.py` prints the table into every line (`predicted`), `DESIGN.md` section 10 quotes it, and the
world-N dry-run test checks the sharding it is computed from.  The constants are the ones `csrc/miniroad.cpp` chooses the pass with
(`prego_miniroad_forward`: chunked = recurrence estimate + rows x (projection flops at 1.4 PFLOP/s + 3 ns) + 30 us per chunk; split =
max(steps x 2.0 us, rows x (flops at 1.4 PFLOP/s + pack bytes at 5.3 TB/s + 1.5 ns) x 8 / (8 - R)) + 1.5 ms), with the measured /
estimated ratios of round 6's devices folded in (chunked 0.98, split 0.93): a prediction, to be held against the driver's SCALE file.

Why strong scaling of the eval metric stops at ~1.6x: clips are independent but a clip is sequential (rnn.py:49,60-61: h_t needs
h_t-1).  However many GPUs share the one 182-clip set, the rank that owns the longest clip (33 981 frames in the bench workload)
runs 33 981 recurrence steps of ~1.6-1.8 us = 55-61 ms, against 93-96 ms for the whole set on one GPU.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

from .data import shard_clips

GEMM_NS_PER_FRAME = 2.0 * (4096 * 2048 + 2048 * 3072) / 1.4e15 * 1e9        # 20.97 ns: both projections at 1.4 PFLOP/s
PACK_NS_PER_FRAME = 4096 * (4.0 + 2.0) / 5.3e12 * 1e9                        # 4.64 ns: fp32 features in, 16-bit rows out at 5.3 TB/s
STEP_US_CHUNKED = {16: 1.80, 8: 1.70, 4: 1.59}                                # per recurrence step at <= n live columns per group
STEP_US_SPLIT = 2.0                                                           # in a split pass, beside the feed-forward launch
ROWS_PER_CHUNK = 49152
MEASURED_OVER_ESTIMATED = {"chunked": 0.98, "split": 0.93}                    # round-6 devices: 119.7 / 94.8 ms measured


def _lpt_max_load(lens: Sequence[int], slots: int) -> int:
    """continuous batching: clips longest-first onto the lightest of `slots` recurrence slots; the pass is as long as the fullest slot"""
    load = [0] * max(1, slots)
    for t in sorted((int(x) for x in lens), reverse=True):
        i = min(range(len(load)), key=load.__getitem__)
        load[i] += t
    return max(load) if lens else 0


def predict_eval_pass_ms(lens: Sequence[int]) -> Dict[str, float]:
    """one pass (features resident in HBM, rgb + flow) over `lens` on one GPU: the chunked and the split estimate and the one the
    library would take"""
    n, frames = len(lens), int(sum(int(x) for x in lens))
    if n == 0:
        return {"ms": 0.0, "pass": "none", "chunked_ms": 0.0, "split_ms": None, "sequential_steps": 0, "frames": 0, "clips": 0}
    slots = min(128, n)
    steps_c = _lpt_max_load(lens, slots)
    per_group = (slots + 7) // 8
    step_us = STEP_US_CHUNKED[16] if per_group > 8 else STEP_US_CHUNKED[8] if per_group > 4 else STEP_US_CHUNKED[4]
    chunks = -(-frames // ROWS_PER_CHUNK)
    chunked = (steps_c * step_us * 1e-3 + frames * (GEMM_NS_PER_FRAME + 3.0) * 1e-6 + 0.03 * chunks) * MEASURED_OVER_ESTIMATED["chunked"]
    split = None
    steps_s = None
    if n >= 48 and frames >= 262144:
        for r in (3, 4):
            if n < 16 * r:
                continue
            st = _lpt_max_load(lens, 16 * r)
            e = (max(st * STEP_US_SPLIT * 1e-3, frames * (GEMM_NS_PER_FRAME + PACK_NS_PER_FRAME + 1.5) * 1e-6 * 8.0 / (8 - r)) + 1.5) * \
                MEASURED_OVER_ESTIMATED["split"]
            if split is None or e < split:
                split, steps_s = e, st
    use_split = split is not None and split < 0.98 * chunked
    return {"ms": split if use_split else chunked, "pass": "split" if use_split else "chunked", "chunked_ms": chunked, "split_ms": split,
            "sequential_steps": steps_s if use_split else steps_c, "frames": frames, "clips": n}


def predict_eval_scaling(lens_of_rank0: Sequence[int], lens_fn=None, worlds: Sequence[int] = (1, 2, 4, 8)) -> Dict[str, List[dict]]:
    """frames/s of the eval metric at N GPUs.  weak: every rank its own clip list of the eval-set size (lens_fn(rank), bench.py's
    default); strong: the ONE list of rank 0 sharded by data.shard_clips.  Pass time = the slowest rank's."""
    out = {"weak": [], "strong": []}
    base = None
    for w in worlds:
        per = [predict_eval_pass_ms(lens_fn(r) if lens_fn is not None else lens_of_rank0) for r in range(w)]
        ms = max(p["ms"] for p in per)
        fps = sum(p["frames"] for p in per) / ms * 1e3
        out["weak"].append({"n_gpus": w, "ms_per_step": ms, "frames_per_s": fps})
        shards = [[lens_of_rank0[i] for i in shard_clips(lens_of_rank0, w, r)] for r in range(w)]
        per = [predict_eval_pass_ms(s) for s in shards]
        ms = max(p["ms"] for p in per)
        fps = sum(p["frames"] for p in per) / ms * 1e3
        if base is None:
            base = fps
        out["strong"].append({"n_gpus": w, "ms_per_step": ms, "frames_per_s": fps, "speedup": fps / base,
                              "passes": sorted({p["pass"] for p in per}), "max_sequential_steps": max(p["sequential_steps"] for p in per),
                              "frames_per_rank": [p["frames"] for p in per]})
    for row in out["weak"]:
        row["speedup"] = row["frames_per_s"] / out["weak"][0]["frames_per_s"]
    longest = max(int(x) for x in lens_of_rank0) if len(lens_of_rank0) else 0
    out["strong_bound"] = {"longest_clip_frames": longest, "floor_ms": longest * STEP_US_CHUNKED[4] * 1e-3,
                           "why": "a clip's recurrence is sequential (rnn.py:60-61); the rank that owns the longest clip runs that many steps "
                                  "whatever N is, so strong scaling of ONE eval set saturates at one-GPU time / floor"}
    return out


# training step (BASELINE configs[2]): measured 1.181 ms at 16 windows per GPU and 1.009 ms at 2 (profiles/r06_bench_line_train*.json;
# round 5: 1.258 / 1.072): 128 + 128 sequential recurrence / BPTT steps are the fixed part, the GEMMs the part that scales with the batch
TRAIN_FIXED_MS, TRAIN_MS_PER_WINDOW = 0.984, 0.0123
GRAD_BYTES = 71_704_920
XGMI_LINK_GBS = 153.0          # per direction and link; 7 links per GPU, point to point


def predict_train_step_ms(local_batch: int, world: int, rings: int = 4, compress_bf16: bool = False) -> Dict[str, float]:
    """step time of window-sharded data parallel training.  The gradient all-reduce runs in three sub-buckets from inside the backward
    (head 0.35 MB before the BPTT, GRU 37.8 MB under the layer1 / LayerNorm tail, layer1 33.6 MB behind the backward): the last
    sub-bucket is exposed, the GRU one partly (the tail it hides under is ~0.2 ms).  Ring all-reduce of S bytes over N ranks moves
    2 (N - 1) / N x S per rank; RCCL runs several rings over the 7 point-to-point links (`rings`; 1 = the single-ring floor)."""
    compute = TRAIN_FIXED_MS + TRAIN_MS_PER_WINDOW * local_batch
    if world <= 1:
        return {"ms": compute, "compute_ms": compute, "exposed_allreduce_ms": 0.0}
    wire = 0.5 if compress_bf16 else 1.0
    t = lambda nbytes: 2.0 * (world - 1) / world * nbytes * wire / (XGMI_LINK_GBS * 1e9 * rings) * 1e3 + 0.03      # + launch / sync latency
    gru, l1 = t(37.8e6), t(33.6e6)
    exposed = max(0.0, gru - 0.2) + l1
    return {"ms": compute + exposed, "compute_ms": compute, "exposed_allreduce_ms": exposed}


def predict_train_scaling(global_batch: int = 16, local_batch: int = 16, worlds: Sequence[int] = (1, 2, 4, 8)) -> Dict[str, List[dict]]:
    out = {"strong": [], "weak": []}
    for w in worlds:
        lb = max(1, global_batch // w)
        p = predict_train_step_ms(lb, w)
        out["strong"].append({"n_gpus": w, "local_batch": lb, "ms_per_step": p["ms"], "frames_per_s": lb * w * 128 / p["ms"] * 1e3,
                              "exposed_allreduce_ms": p["exposed_allreduce_ms"]})
        p = predict_train_step_ms(local_batch, w)
        out["weak"].append({"n_gpus": w, "local_batch": local_batch, "ms_per_step": p["ms"],
                            "frames_per_s": local_batch * w * 128 / p["ms"] * 1e3, "exposed_allreduce_ms": p["exposed_allreduce_ms"]})
    for k in out:
        for row in out[k]:
            row["speedup"] = row["frames_per_s"] / out[k][0]["frames_per_s"]
    return out
