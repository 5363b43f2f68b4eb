"""`Evaluate` ("OAD") behind the reference's EVAL registry (step_recognition/trainer/eval.py:15-84):
the per-frame inference loop of `main.py --eval`.

Same call contract: `Evaluate(cfg)(model, dataloader, logger, device) -> mean_AP`, same side effect
(`output_miniRoad/output_miniROAD.json` = {vid: {"pred": [...], "gt": [...]}} when cfg['eval'] is set,
eval.py:51-65).  What changes is how the loop runs: the reference pushes ONE video per forward through
a batch-1 GRU and copies [T, C] probabilities back per video; here whole videos are batched (up to the
engine's clip capacity / a frame budget), advanced together by the ragged HIP path, and argmax is
taken on the device.  The FPS log line is computed correctly (the reference shadows its timer with the
loader's `start` field, eval.py:35-36,77-80)."""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from .metrics import perframe_average_precision, perframe_average_precision_device
from .registry import EVAL


@EVAL.register("OAD")
class Evaluate(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.data_processing = None          # thumos_postprocessing applies to THUMOS only (eval.py:19-21)
        if "THUMOS" in cfg["data_name"]:
            raise NotImplementedError("THUMOS post-processing is outside the PREGO datasets")
        self.metric = cfg["metric"]
        self.cfg = cfg
        self.all_class_names = json.load(open(cfg["video_list_path"]))[cfg["data_name"].split("_")[0]]["class_index"]
        self.max_frames_per_batch = int(cfg.get("eval_frames_per_batch", 4_000_000))
        self.output_dir = cfg.get("eval_output_dir", "output_miniRoad")
        self.last_fps = None
        self._copy_stream = None             # side stream for the H2D feature copies (double buffering against the compute stream)

    def _zero_flow(self, model) -> bool:
        """the flow half is identically zero - never shipped, its half of layer1's K never multiplied (exact): the model was told so
        (cfg['assume_zero_flow']), or the config names the flow type the reference's loader overwrites with zeros (dataset.py:63-69)"""
        known_zero = bool(getattr(model, "assume_zero_flow", False)) or self.cfg.get("flow_type") == "flow_anet_resnet50"
        return known_zero and bool(getattr(model, "use_rgb", True)) and bool(self.cfg.get("eval_skip_zero_flow", True))

    @staticmethod
    def _features(model, x):
        """fp32 as the reference ships them (eval.py:40-45), or - when the feeder already holds the model's 16-bit operand type
        (cfg['feature_dtype']) - unchanged: half the PCIe bytes per frame, no conversion on either side"""
        keep = {"fp16": torch.float16, "bf16": torch.bfloat16}.get(getattr(model, "compute_dtype", None))
        return x.contiguous() if (keep is not None and x.dtype == keep) else x.float().contiguous()

    @staticmethod
    def _json_int_lists(output) -> bytes:
        """the reference's output file ({vid: {"pred": [...], "gt": [...]}}, eval.py:59-65) written from the int arrays directly:
        class ids through a table of pre-formatted byte strings (json.dump walks 2 x frames Python ints one at a time: 0.25 s for
        0.8 M frames, five times the forward pass).  Same JSON value; numbers are padded with blanks, which JSON allows."""
        lut4 = np.frombuffer(b"".join(("%3d," % i).encode() for i in range(100)), dtype=np.uint32)         # one 4-byte gather per id < 100
        lut5 = np.frombuffer(b"".join(("%4d," % i).encode() for i in range(1000)), dtype=np.uint8).reshape(1000, 5)

        def arr(a):
            a = np.asarray(a)
            if a.size == 0:
                return b"[]"
            lo, hi = int(a.min()), int(a.max())
            if lo < 0 or hi >= 1000:
                return json.dumps(a.tolist()).encode()
            body = lut4[a].tobytes() if hi < 100 else np.take(lut5, a, axis=0).tobytes()
            return b"[" + body[:-1] + b"]"
        parts = [json.dumps(str(vid)).encode() + b': {"pred": ' + arr(v["pred"]) + b', "gt": ' + arr(v["gt"]) + b"}" for vid, v in output.items()]
        return b"{" + b", ".join(parts) + b"}"

    def _enqueue(self, model, sub, device):
        """H2D of one sub-batch on the copy stream + its forward on the compute stream; nothing here waits for the GPU"""
        zero_flow = self._zero_flow(model)
        dev = torch.device(device)
        if dev.type == "cuda":
            # H2D on a side stream: these copies run while whatever was enqueued before is still computing (the loader's
            # pin_memory=True makes them true async DMA); the compute stream waits on one event per sub-batch
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(dev)
            cur = torch.cuda.current_stream(dev)
            with torch.cuda.stream(self._copy_stream):
                rgb = [b[0].to(dev, non_blocking=True) for b in sub]
                flow = None if zero_flow else [b[1].to(dev, non_blocking=True) for b in sub]
                tgt = [b[2].to(dev, non_blocking=True) for b in sub]
                ready = torch.cuda.Event()
                ready.record(self._copy_stream)
            cur.wait_event(ready)
            for t in rgb + (flow or []) + tgt:
                t.record_stream(cur)         # allocated on the copy stream, consumed on the compute stream
        else:
            rgb = [b[0].to(device) for b in sub]
            flow = None if zero_flow else [b[1].to(device) for b in sub]
            tgt = [b[2].to(device) for b in sub]
        probs, args, _ = model.forward_clips(rgb, flow, want_probs=True, want_argmax=True)
        return probs, args, tgt

    def _flush(self, model, batch, pred_scores, gt_targets, output, device):
        if not batch:
            return
        # A batch runs as long as its longest video's recurrence; its H2D copy would sit in front of that.  Large batches go in two
        # sub-batches ordered by length: the few longest videos first (little to copy, the long critical path), then the bulk of
        # the bytes, whose copy runs under the first sub-batch's recurrence.  Results return to the loader's order below.
        order = list(range(len(batch)))
        parts = [order]
        frames = [int(b[0].shape[0]) for b in batch]
        if len(batch) >= 16 and torch.device(device).type == "cuda" and self.cfg.get("eval_split_by_length", True):
            order.sort(key=lambda i: -frames[i])
            # share of the frames in the first part: its forward should last about as long as the second part's copy.  Measured on
            # the 60-video bench set (scripts/eval_e2e_bench.py): fp32 features 0.2: 4.23, 0.5: 4.72, 0.6: 4.41 M frames/s;
            # 16-bit features (half the bytes) 0.2: 5.87, 0.5: 5.62
            frac = self.cfg.get("eval_split_fraction")
            if frac is None:
                frac = 0.5 if batch[0][0].element_size() >= 4 else 0.2
            budget, k, acc = float(frac) * sum(frames), 0, 0
            while k < len(order) - 1 and acc + frames[order[k]] <= budget:
                acc += frames[order[k]]
                k += 1
            if k >= 1:
                parts = [order[:k], order[k:]]
        res = {}
        for part in parts:
            probs, args, tgt = self._enqueue(model, [batch[i] for i in part], device)
            for i, p, a, t in zip(part, probs, args, tgt):
                res[i] = (p, a, t)
        want_json = self.cfg["eval"] is not None
        # ONE device -> host copy of the whole batch's argmax, after everything has been enqueued (the first wait for the GPU)
        arg_host = torch.cat([res[i][1] for i in range(len(batch))]).cpu().numpy() if want_json else None
        o = 0
        for i, (r, f, target, vid) in enumerate(batch):
            p, a, t = res[i]
            # the [T, C] score and target matrices stay torch tensors on the model's device (one entry per video, concatenated
            # once at the end); the reference extends Python lists by one row object per frame (eval.py:46-49)
            pred_scores.append(p)
            gt_targets.append(t)
            if want_json:
                n = int(a.shape[0])
                output[vid] = {"pred": arg_host[o:o + n], "gt": torch.argmax(target, dim=1).numpy()}      # int arrays; text only at the end
                o += n
                self.last_device_argmax[vid] = a          # int32 on the device: input of aggregate_device (utils/aggregate.py)
        batch.clear()

    def eval(self, model, dataloader, logger, device):
        model.eval()
        output = {}
        self.last_device_argmax = {}
        max_clips = model.max_clips          # MROAD: the engine's clip capacity; ViTEnc (`model: 'Transformer'`): its own bound
        # data-parallel eval: videos are independent, so under torch.distributed rank r takes every world-th video of
        # the loader's order (no data-path collective); rank 0 gathers the per-video results once at the end
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        skip_flow = self._zero_flow(model)
        with torch.no_grad():
            pred_scores, gt_targets = [], []
            per_video = []                       # (loader position, n_frames) to restore the loader's order on rank 0
            t_begin = time.time()
            batch, frames, pos = [], 0, 0
            for rgb_input, flow_input, target, vid, start, end in dataloader:
                # loader items carry a leading batch dim of test_batch_size == 1 (dataset_builder.py:19)
                for b in range(rgb_input.shape[0]):
                    mine = (pos % world) == rank
                    pos += 1
                    if not mine:
                        continue
                    name = vid[b] if isinstance(vid, (list, tuple)) else vid
                    batch.append((self._features(model, rgb_input[b]), None if skip_flow else self._features(model, flow_input[b]),
                                  target[b], name))
                    per_video.append((pos - 1, int(rgb_input.shape[1])))
                    frames += rgb_input.shape[1]
                if len(batch) >= max_clips or frames >= self.max_frames_per_batch:
                    self._flush(model, batch, pred_scores, gt_targets, output, device)
                    frames = 0
            self._flush(model, batch, pred_scores, gt_targets, output, device)
            model.check()
            if world > 1:
                gathered = [None] * world if rank == 0 else None
                dist.gather_object((per_video, [p.cpu().numpy() for p in pred_scores], [g.cpu().numpy() for g in gt_targets], output),
                                   gathered, dst=0)
                if rank != 0:
                    return float("nan")          # only rank 0 reports (main.py logs / checkpoints on rank 0)
                chunks = []
                output = {}
                for pv, ps, gs, out in gathered:
                    for (p_idx, n), pm, gm in zip(pv, ps, gs):       # one [n, C] matrix per video
                        chunks.append((p_idx, torch.from_numpy(pm), torch.from_numpy(gm)))
                    output.update(out)
                chunks.sort(key=lambda c: c[0])
                pred_scores = [c[1] for c in chunks]
                gt_targets = [c[2] for c in chunks]
            if self.cfg["eval"] is not None:
                os.makedirs(self.output_dir, exist_ok=True)
                with open(os.path.join(self.output_dir, "output_miniROAD.json"), "wb") as file:
                    file.write(self._json_int_lists(output))
            t_end = time.time()
            pred_all = torch.cat(pred_scores, 0) if pred_scores else torch.zeros((0, len(self.all_class_names)))
            gt_all = torch.cat(gt_targets, 0).to(pred_all.device) if gt_targets else torch.zeros_like(pred_all)
            num_frames = int(gt_all.shape[0])
            if torch.device(device).type == "cuda":
                # sort + scan per class in libprego_amd.so (prego_perframe_ap); after a multi-rank gather the matrices are host
                # tensors on rank 0 and go back to its GPU first
                result = perframe_average_precision_device(pred_all.to(device), gt_all.to(device), self.all_class_names,
                                                           self.data_processing, self.metric)
            else:       # only reachable with a stand-in model on a CPU box (the gloo tests of the sharding logic)
                result = perframe_average_precision(pred_all.numpy(), gt_all.numpy(), self.all_class_names, self.data_processing, self.metric)
            time_taken = max(t_end - t_begin, 1e-9)
            self.last_fps = num_frames / time_taken
            logger.info(f"Processed {num_frames} frames in {time_taken:.1f} seconds ({self.last_fps:.1f} FPS)")
        return result["mean_AP"]

    def aggregate_last(self, output_path=None, window_size: int = 200):
        """utils/aggregate.py:46-90 on the per-frame argmax the last eval left in HBM (`last_device_argmax`), with this config's
        number of classes; ground truth from the output JSON's "gt" lists"""
        from .aggregate import aggregate_device
        js = json.load(open(os.path.join(self.output_dir, "output_miniROAD.json")))
        return aggregate_device(self.last_device_argmax, {k: v["gt"] for k, v in js.items()}, output_path, window_size,
                                n_classes=len(self.all_class_names))

    def forward(self, model, dataloader, logger, device):
        return self.eval(model, dataloader, logger, device)
