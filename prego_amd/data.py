"""Feature feeder replacing step_recognition/datasets/dataset.py:24-135.

Same on-disk format ({root}/{annotation_type|rgb_type}/{vid}.npy: float [T, 2048] features, [T, C] one-hot
targets), same items `(rgb[T,2048] f32, flow[T,2048] f32, target[T,C] f32, vid, start, end)`, same training
windowing (front zero-pad window_size-1, stride 4, random phase re-drawn by `_init_features()` each epoch,
dataset.py:53-55,96-123) and whole-video test items (dataset.py:120-123).

Differences on purpose: no `ipdb` breakpoints (dataset.py:108,112), a missing video is skipped without
mutating the list being iterated (dataset.py:87-94 skips the element after each failure), pads are float32
(the reference's np.zeros pads are float64 and get re-cast per item), and when flow_type is
'flow_anet_resnet50' the flow half - which the reference overwrites with zeros (dataset.py:63-69) - is a
stride-0 zero tensor instead of a materialised [T,2048] array.

cfg['feature_dtype'] (not a reference key; default 'fp32'): 'fp16' / 'bf16' keep the TEST-mode features in that 16-bit type
in host memory - converted once at load, round to nearest even, the conversion the pack kernel would do on the device - so the
eval loop ships half the bytes per frame over PCIe to a model whose operand type it is (PREGO_FWD_IN16).  Training items stay
fp32 (the training kernels take fp32 features)."""
from __future__ import annotations

import json
import os.path as osp

import numpy as np
import torch
import torch.utils.data as data

from .config import FEATURE_SIZES
from .registry import DATA_LAYERS

# datasets/dataset.py:100-107: `_init_features` removes this Assembly101-O session from `self.vids` in every mode (it is in
# ASSEMBLY101-O's test_session_set, so the reference scores 181 test videos, not 182).  cfg['exclude_videos'] overrides.
REFERENCE_EXCLUDED_VIDEOS = ("nusar-2021_action_both_9056-b08a_9056_user_id_2021-02-22_141934",)


class StepRecognitionDataset(data.Dataset):
    def __init__(self, cfg, mode="train"):
        self.root_path = cfg["root_path"]
        self.mode = mode
        self.training = mode == "train"
        self.window_size = cfg["window_size"]
        self.stride = cfg["stride"]
        self.num_classes = cfg["num_classes"]
        self.annotation_type = cfg["annotation_type"]
        self.rgb_type = cfg["rgb_type"]
        self.flow_type = cfg["flow_type"]
        self.zero_flow = cfg["flow_type"] == "flow_anet_resnet50"      # dataset.py:63-69
        fd = cfg.get("feature_dtype", "fp32")
        if fd not in ("fp32", "fp16", "bf16"):
            raise ValueError(f"feature_dtype {fd!r}: expected fp32, fp16 or bf16")
        self.feature_dtype = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[fd if not self.training else "fp32"]
        vids = json.load(open(cfg["video_list_path"]))[cfg["data_name"]][mode + "_session_set"]
        self.excluded = tuple(cfg.get("exclude_videos", REFERENCE_EXCLUDED_VIDEOS))
        self.vids, self.removed = [], 0
        self.target_all, self.rgb_inputs, self.flow_inputs = {}, {}, {}
        pad = self.window_size - 1
        d_flow = FEATURE_SIZES[self.flow_type]
        for vid in vids:
            try:
                target = np.load(osp.join(self.root_path, self.annotation_type, vid + ".npy")).astype(np.float32)
                rgb = np.load(osp.join(self.root_path, self.rgb_type, vid + ".npy")).astype(np.float32)
                if self.zero_flow:
                    flow = None
                else:
                    sub = "assembly_optical_flow_BNInception/" + vid + "/assembling.npy"
                    flow = np.load(osp.join(self.root_path, self.flow_type, sub)).astype(np.float32)
            except Exception as e:     # same policy as the reference: drop videos without features
                print("---- Exception in loading video ", e)
                self.removed += 1
                continue
            if self.training:
                target = np.concatenate((np.zeros((pad, self.num_classes), np.float32), target), 0)
                rgb = np.concatenate((np.zeros((pad, rgb.shape[1]), np.float32), rgb), 0)
                if flow is not None:
                    flow = np.concatenate((np.zeros((pad, d_flow), np.float32), flow), 0)
            if vid in self.excluded:       # dataset.py:100-107 (loaded, then dropped from the item list)
                continue
            self.vids.append(vid)
            if self.feature_dtype != torch.float32:          # once per video, at load: torch's CPU casts round to nearest even
                rgb = torch.from_numpy(rgb).to(self.feature_dtype)
                flow = None if flow is None else torch.from_numpy(flow).to(self.feature_dtype)
            self.target_all[vid], self.rgb_inputs[vid], self.flow_inputs[vid] = target, rgb, flow
        self._zero_row = torch.zeros(1, d_flow, dtype=self.feature_dtype)
        self._init_features()

    def _init_features(self):
        """re-draw the window phase (main.py:100 calls this after every epoch)"""
        self.inputs = []
        for vid in self.vids:
            n = self.target_all[vid].shape[0]
            if self.training:
                seed = np.random.randint(self.stride)
                for start, end in zip(range(seed, n, self.stride), range(seed + self.window_size, n + 1, self.stride)):
                    self.inputs.append((vid, start, end))
            else:
                self.inputs.append((vid, 0, n))

    def __getitem__(self, index):
        # index >= len: a PAD entry of EpochWindowSampler (short last global batch of a data-parallel epoch) = window index - len
        # with an all-zero target, i.e. zero loss and zero gradient under OadLoss (loss.py:28-29)
        pad = index >= len(self.inputs)
        vid, start, end = self.inputs[index - len(self.inputs) if pad else index]
        as_t = (lambda a: a) if self.feature_dtype != torch.float32 else torch.from_numpy
        rgb = as_t(self.rgb_inputs[vid][start:end])
        flow = self.flow_inputs[vid]
        flow = self._zero_row.expand(end - start, -1) if flow is None else as_t(flow[start:end])
        target = torch.from_numpy(self.target_all[vid][start:end])
        if pad:
            target = torch.zeros_like(target)
        return rgb, flow, target, vid, start, end

    def __len__(self):
        return len(self.inputs)


for _name in ("ASSEMBLY101-O", "EPIC-TENT-O"):
    DATA_LAYERS.register(_name, StepRecognitionDataset)


def build_data_loader(cfg, mode):
    """datasets/dataset_builder.py:15-24.  Under torch.distributed the TRAIN loader shards the windows across ranks
    (DistributedSampler pads the last batch so that every rank takes the same number of steps: the loss is a batch mean,
    criterions/loss.py:30-31, so equal local batches + a gradient all-reduce-mean reproduce the single-process step with
    global batch = world * batch_size; set cfg['batch_size'] = 16 / world to keep the reference's global batch)."""
    import torch.distributed as dist
    ds = DATA_LAYERS[cfg["data_name"]](cfg, mode)
    sampler = None
    if mode == "train" and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        sampler = EpochWindowSampler(ds, cfg["batch_size"])
    return data.DataLoader(dataset=ds, batch_size=cfg["batch_size"] if mode == "train" else cfg["test_batch_size"],
                           shuffle=(mode == "train" and sampler is None), sampler=sampler,
                           num_workers=cfg["num_workers"], pin_memory=True)


class EpochWindowSampler(data.Sampler):
    """DistributedSampler for a dataset whose length changes every epoch (`_init_features()` re-draws the window phase,
    main.py:100, so len(dataset) moves by a few windows): the permutation, the padding and the per-rank share are
    recomputed from the CURRENT dataset length at every `__iter__`, and the epoch number seeds the shuffle
    (`set_epoch`, called by train_one_epoch).

    Global step j takes windows order[j * Bg : (j + 1) * Bg] (Bg = world * batch_size); rank r its contiguous block of batch_size
    of them, so all ranks run the same number of steps.  The reference's DataLoader has no drop_last (dataset_builder.py:17-23): its
    last batch is SHORT and its loss is the mean over the windows that exist.  Here the short last global batch is filled up with
    PAD entries - index + len(dataset), which StepRecognitionDataset turns into that window with an all-zero target: zero loss and
    zero gradient under OadLoss (loss.py:28-29, F.normalize of a zero row) - and `step_weight(j)` = Bg / (real windows of step j)
    is what the trainer multiplies the averaged gradient (and the logged loss) with, so the step equals the reference's mean over
    the real windows.  No window is counted twice."""

    def __init__(self, dataset, batch_size: int = 1, seed: int = 0):
        import torch.distributed as dist
        self.dataset = dataset
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.batch_size = int(batch_size)
        self.seed = seed
        self.epoch = 0

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def _steps(self, n):
        bg = self.world * self.batch_size
        return (n + bg - 1) // bg

    def __len__(self):
        return self._steps(len(self.dataset)) * self.batch_size

    def step_weight(self, step: int) -> float:
        """global batch size / number of real (non-pad) windows in global step `step` of the current epoch"""
        n, bg = len(self.dataset), self.world * self.batch_size
        real = min(bg, n - step * bg)
        return bg / real if real > 0 else 1.0

    def __iter__(self):
        n = len(self.dataset)
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        order = torch.randperm(n, generator=g).tolist()
        bg, b = self.world * self.batch_size, self.batch_size
        total = self._steps(n) * bg
        # pads: index + n = "that window, zero target"; cyclic, so that a dataset smaller than the padding (n < total - n: a tiny set
        # against world x batch) still gives every rank the same number of entries (a short order would hang the gradient all-reduce)
        order += [order[k % n] + n for k in range(total - n)] if n > 0 else []
        mine = []
        for j in range(0, total, bg):
            mine += order[j + self.rank * b: j + (self.rank + 1) * b]
        return iter(mine)


# ---- clip sharding for data-parallel inference --------------------------------
def shard_clips(lengths, world_size: int, rank: int):
    """Length-balanced partition of the clip list: longest first onto the currently lightest rank (T varies
    8x on Epic-tent-O).  Deterministic; every rank computes the same partition.  Returns the clip indices
    of `rank`, in the original order."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    owner = [0] * len(lengths)
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += int(lengths[i])
    return [i for i in range(len(lengths)) if owner[i] == rank]
