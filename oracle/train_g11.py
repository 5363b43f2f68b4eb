#!/usr/bin/env python3
"""Fixture G11: parity on TRAINED weights.  Runs only in the build container (needs /root/reference).

The reference ships no checkpoint (README.md:48-53 points at Drive), so this script makes one the way the reference would:
its own `build_data_loader` / `THUMOSDataset` (datasets/dataset.py:24-135, windows of 128, stride 4, front padding, zero flow
as the shipped yaml's `flow_anet_resnet50` branch has it), its own `build_model`, `build_criterion`, AdamW exactly as
main.py:62-67 builds it (lr 1e-4, weight decay 0.05) and its own `train_one_epoch` (trainer/train.py:6-29), dropout 0.2 active,
on a synthetic learnable feature tree (`prego_amd/workloads.py:action_video`) written to a temp dir.  Initial weights come from
the build-owned generator (seed 20, head gain 1), so `trained - init` is a reproducible delta.

    python oracle/train_g11.py train a101 500 6      # C = 86, 6 epochs of 500 steps (~0.5 s each on 8 cores) -> gpurun_out/_scratch/
    python oracle/train_g11.py train epic 400 3      # C = 12
    python oracle/train_g11.py pack a101             # quantise the delta -> tests/golden/g11_weights_a101.npz
    python oracle/train_g11.py eval a101             # reference Evaluate on the PACKED weights -> tests/golden/g11_eval_a101.npz

The 72 MB fp32 tensors cannot be committed.  What is committed is `init(seed) + dequant(delta)`: the trained delta of every
tensor on a 4-bit grid (15 levels over +-3.5 sigma, clipped) - and THAT model is the fixture model: the reference's Evaluate is
run on exactly those weights, the GPU tests rebuild exactly those weights (`prego_amd.weights.g11_state_dict`).  The pack step
reports the reference's loss / accuracy before and after quantisation, so "still a trained model" is checked, not assumed.
Only data is written; no reference source text is copied anywhere.
"""
from __future__ import annotations

import json
import logging
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

from gen_golden import _stub_modules, _load, REF, OUT       # noqa: E402
from prego_amd import weights as W                           # noqa: E402
from prego_amd import workloads as WL                        # noqa: E402
from prego_amd.config import assembly101_cfg, epic_tent_cfg  # noqa: E402

SCRATCH = os.path.join(REPO, "gpurun_out", "_scratch")       # never travels to a GPU box, never committed
CFGS = {"a101": assembly101_cfg, "epic": epic_tent_cfg}
TRAIN_VIDEO_FRAMES = 800
# the eval videos of the fixture: two short ones, one past 4 096 frames, the longest Epic-tent-O length
EVAL_LENGTHS = {"a101": [300, 517, 5000, 31114], "epic": [190, 1007, 4500, 31114]}


def _write_tree(root, cfg, tag, n_videos):
    C = cfg["num_classes"]
    for sub in ("target_perframe", "rgb_anet_resnet50", "rgb_as_flow/rgb_anet_resnet50"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    vids = []
    for i in range(n_videos):
        vid = f"g11_{tag}_train_{i}"
        rgb, lab = WL.action_video(TRAIN_VIDEO_FRAMES, C, 20, f"g11.{tag}.train.{i}", max_extra=100)
        np.save(os.path.join(root, "rgb_anet_resnet50", vid + ".npy"), rgb)
        # dataset.py:62-69 loads this file for its shape only and zeroes it
        np.save(os.path.join(root, "rgb_as_flow/rgb_anet_resnet50", vid + ".npy"), np.zeros((TRAIN_VIDEO_FRAMES, 2048), np.float32))
        np.save(os.path.join(root, "target_perframe", vid + ".npy"), WL.onehot(lab, C))
        vids.append(vid)
    vl = os.path.join(root, "video_list.json")
    json.dump({cfg["data_name"]: {"train_session_set": vids, "test_session_set": vids[:1],
                                  "class_index": [f"c{k}" for k in range(C)]}}, open(vl, "w"))
    return vl


def train(tag, steps, epochs=1):
    from model import build_model
    from criterions import build_criterion
    from trainer import build_trainer
    from datasets import build_data_loader
    cfg = CFGS[tag]()
    n_videos = (steps * cfg["batch_size"] * cfg["stride"] + TRAIN_VIDEO_FRAMES - 1) // TRAIN_VIDEO_FRAMES
    tmp = tempfile.mkdtemp(dir=SCRATCH)
    try:
        vl = _write_tree(tmp, cfg, tag, n_videos)
        cfg.update(root_path=tmp, video_list_path=vl, num_workers=0, device="cpu")
        np.random.seed(20)
        torch.manual_seed(20)
        loader = build_data_loader(cfg, "train")
        model = _load(build_model(cfg, "cpu"), W.miniroad_state_dict(cfg, seed=20))
        crit = build_criterion(cfg, "cpu")
        optimizer = torch.optim.AdamW([{"params": model.parameters(), "initial_lr": cfg["lr"]}], lr=cfg["lr"],
                                      weight_decay=cfg["weight_decay"])           # main.py:62-67
        train_one_epoch = build_trainer(cfg)
        print(f"{tag}: {len(loader)} steps per epoch, {n_videos} videos", flush=True)

        # the reference's loop has no step limit and no logging hook: count steps through the criterion it calls once per step
        losses = []
        t0 = time.time()

        class _Crit(torch.nn.Module):
            def forward(self, out, target):
                l = crit(out, target)
                losses.append(float(l))
                if len(losses) % 10 == 0:
                    print(f"  step {len(losses)} loss {np.mean(losses[-10:]):.4f}  {time.time() - t0:.0f} s", flush=True)
                return l
        for epoch in range(1, epochs + 1):                      # main.py:88-100 (the per-epoch eval is not needed here)
            train_one_epoch(loader, model, _Crit(), optimizer, None, epoch, "cpu")
            loader.dataset._init_features()
        os.makedirs(SCRATCH, exist_ok=True)
        np.savez(os.path.join(SCRATCH, f"g11_{tag}_trained.npz"), losses=np.array(losses),
                 **{k: v.numpy() for k, v in model.state_dict().items()})
        print(f"{tag}: trained {len(losses)} steps, loss {losses[0]:.3f} -> {np.mean(losses[-10:]):.3f}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


LEVELS = 7          # symmetric grid -7 .. 7 (15 levels, 4 bits)
CLIP_SIGMA = 3.5


def pack(tag):
    cfg = CFGS[tag]()
    tr = np.load(os.path.join(SCRATCH, f"g11_{tag}_trained.npz"))
    init = W.miniroad_state_dict(cfg, seed=20)
    save = {"steps": np.int64(len(tr["losses"])), "loss_curve": tr["losses"].astype(np.float32)}
    for k, w0 in init.items():
        d = (tr[k].astype(np.float64) - w0.astype(np.float64)).reshape(-1)
        scale = CLIP_SIGMA * float(d.std()) / LEVELS
        q = np.clip(np.rint(d / scale), -LEVELS, LEVELS).astype(np.int8)
        u = (q + LEVELS).astype(np.uint8)
        if u.size % 2:
            u = np.append(u, np.uint8(LEVELS))
        save["q." + k] = (u[0::2] | (u[1::2] << 4)).astype(np.uint8)
        save["scale." + k] = np.float32(scale)
        err = d - q.astype(np.float64) * scale
        print(f"  {k:28s} |delta| rms {d.std():.2e} (init rms {w0.std():.2e})  quantisation rms {err.std():.2e}")
    path = os.path.join(OUT, f"g11_weights_{tag}.npz")
    np.savez_compressed(path, **save)
    print(f"{tag}: packed -> {path} ({os.path.getsize(path) / 1e6:.1f} MB)")
    # trained vs packed under the reference: same held-out video, loss of the last-frame rows the way OadLoss takes them
    from model import build_model
    rgb, lab = WL.action_video(1500, cfg["num_classes"], 20, f"g11.{tag}.heldout")
    x = torch.from_numpy(rgb)[None]
    for name, sd in (("init", init), ("trained", {k: tr[k] for k in init}), ("packed", W.g11_state_dict(tag))):
        model = _load(build_model(cfg, "cpu"), sd).eval()
        with torch.no_grad():
            p = model(x, torch.zeros_like(x))["logits"][0].numpy()
        nll = float(-np.log(np.maximum(p[np.arange(len(lab)), lab], 1e-30)).mean())
        print(f"  {name:8s} held-out NLL {nll:.3f}  frame accuracy {float((p.argmax(1) == lab).mean()):.3f}  mean top-1 prob {float(p.max(1).mean()):.3f}")


def evaluate(tag):
    """The reference's Evaluate (trainer/eval.py:30-84) on the packed weights: JSON ids, mAP, every frame's argmax and margin,
    probabilities at sampled frames."""
    from model import build_model
    from trainer import build_eval
    cfg = CFGS[tag]()
    C = cfg["num_classes"]
    lens = EVAL_LENGTHS[tag]
    tmp = tempfile.mkdtemp(dir=SCRATCH)
    vl = os.path.join(tmp, "video_list.json")
    json.dump({cfg["data_name"]: {"class_index": [f"c{k}" for k in range(C)]}}, open(vl, "w"))
    cfg.update(eval="g11.pth", video_list_path=vl)
    model = _load(build_model(cfg, "cpu"), W.g11_state_dict(tag))
    items = []
    for i, T in enumerate(lens):
        rgb, lab = WL.action_video(T, C, 20, f"g11.{tag}.eval.{i}")
        items.append((rgb, lab))

    class _DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(items)

        def __getitem__(self, i):
            rgb, lab = items[i]
            return torch.from_numpy(rgb), torch.zeros(rgb.shape), torch.from_numpy(WL.onehot(lab, C)), f"g11_{tag}_eval_{i}", 0, len(lab)
    loader = torch.utils.data.DataLoader(_DS(), batch_size=1, shuffle=False)
    # capture the probabilities the loop extends its list with: wrap the model's forward
    probs = []
    fwd = model.forward

    def _fwd(r, f):
        out = fwd(r, f)
        if not model.training:
            probs.append(out["logits"][0].numpy().copy())
        return out
    model.forward = _fwd
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        mAP = build_eval(cfg)(model, loader, logging.getLogger("g11"), "cpu")
        js = json.load(open("output_miniRoad/output_miniROAD.json"))
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
    save = {"mAP": np.float64(mAP), "lengths": np.array(lens, np.int64)}
    for i, T in enumerate(lens):
        p = probs[i]
        srt = np.sort(p, 1)
        idx = np.linspace(0, T - 1, 96).astype(np.int64)
        assert js[f"g11_{tag}_eval_{i}"]["pred"] == p.argmax(1).tolist()
        save[f"pred{i}"] = np.array(js[f"g11_{tag}_eval_{i}"]["pred"], np.int16)
        save[f"gt{i}"] = np.array(js[f"g11_{tag}_eval_{i}"]["gt"], np.int16)
        save[f"margin{i}"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)
        save[f"top1{i}"] = srt[:, -1].astype(np.float16)
        save[f"sample_idx{i}"] = idx
        save[f"sample_probs{i}"] = p[idx].astype(np.float32)
        print(f"  video {i} T {T}: accuracy {float((p.argmax(1) == items[i][1]).mean()):.3f}  mean top-1 {float(srt[:, -1].mean()):.3f} "
              f" margins < 1e-3: {int((save[f'margin{i}'] < 1e-3).sum())}  min margin {float(save[f'margin{i}'].min()):.2e}")
    np.savez_compressed(os.path.join(OUT, f"g11_eval_{tag}.npz"), **save)
    print(f"{tag}: mAP {mAP:.4f}")


if __name__ == "__main__":
    os.makedirs(SCRATCH, exist_ok=True)
    _stub_modules()
    torch.set_num_threads(int(os.environ.get("G11_THREADS", "8")))
    cmd, tag = sys.argv[1], sys.argv[2]
    if cmd == "train":
        train(tag, int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    elif cmd == "pack":
        pack(tag)
    elif cmd == "eval":
        evaluate(tag)
