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

from .metrics import (perframe_ap_raw, perframe_ap_raw_device, perframe_average_precision, perframe_average_precision_device,
                      report_from_raw)
from .registry import EVAL


@EVAL.register("OAD")
class Evaluate(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.data_processing = None          # thumos_postprocessing applies to THUMOS only (eval.py:19-21)
        if "THUMOS" in cfg["data_name"]:
            raise NotImplementedError("THUMOS post-processing is outside the PREGO datasets")
        self.metric = cfg["metric"]
        if self.metric not in ("AP", "cAP"):          # known before the eval pass runs, not after it (utils/metrics.py:38-44)
            raise RuntimeError(f"Unknown metrics: {self.metric}")
        self.cfg = cfg
        self.all_class_names = json.load(open(cfg["video_list_path"]))[cfg["data_name"].split("_")[0]]["class_index"]
        # frames per forward.  The loop is pipelined one batch deep (eval: _flush / _collect): while batch k's features cross the link and
        # its forward runs, the host waits for batch k - 1's ids and formats their JSON text.  Measured on the 182-video bench set
        # (2.3 M frames, 16-bit features): ONE batch 226 ms, two batches of 1.2 M frames 238 ms - every forward pays its own fill and tail
        # (~12 ms) and the text it hides is 6 ms -, so the default keeps an eval set of this size in one forward
        self.max_frames_per_batch = int(cfg.get("eval_frames_per_batch", 4_000_000))
        if "eval_piece_frames" in cfg:
            self.PIECE_FRAMES = int(cfg["eval_piece_frames"])
        self.output_dir = cfg.get("eval_output_dir", "output_miniRoad")
        self.last_fps = None
        self._copy_stream = None             # side stream for the H2D feature copies (double buffering against the compute stream)
        self._ap_stream = None               # side stream of the metric's kernels (enqueued behind the last forward)

    @staticmethod
    def _new_copy_stream(dev):
        """The H2D copies must run BESIDE the forward's kernels.  HIP maps streams onto a handful of hardware queues round-robin in
        creation order, and two streams that share a queue execute in submission order: in a process that had created a few streams
        before (bench.py after its main pass) the copy stream landed on the compute stream's queue and the whole transfer ran in FRONT
        of the forward (149 instead of 89 ms for the 60-video set).  A queue has one priority, so a high-priority stream never shares
        the queue of the normal-priority compute stream."""
        return torch.cuda.Stream(dev, priority=-1)

    def _zero_flow(self, model, dataloader=None) -> bool:
        """the flow half of EVERY video is identically zero - never shipped, its half of layer1's K never multiplied (exact).  Known
        from what the data IS, not from a config string: the model was told so (cfg['assume_zero_flow']), or the loader's dataset
        says it zeroes the flow stream (StepRecognitionDataset.zero_flow = the reference's dataset.py:63-69 behaviour).  A loader
        that supplies real flow under the same flow_type name has its flow used.  Single tensors are also checked, see _flow_zero."""
        known_zero = bool(getattr(model, "assume_zero_flow", False)) or bool(getattr(getattr(dataloader, "dataset", None), "zero_flow", False))
        return known_zero and bool(getattr(model, "use_rgb", True)) and bool(self.cfg.get("eval_skip_zero_flow", True))

    def _flow_zero(self, model, flow) -> bool:
        """one video's flow tensor [T, D] is provably all zeros at no cost: a stride-0 expansion of a zero row (what a dataset that
        zeroes the flow stream hands out before collation)"""
        if not (bool(getattr(model, "use_rgb", True)) and bool(self.cfg.get("eval_skip_zero_flow", True))):
            return False
        return flow.dim() == 2 and flow.shape[0] > 0 and (flow.stride(0) == 0 or flow.shape[0] == 1) and not bool(flow[0].any())

    @staticmethod
    def _features(model, x):
        """fp32 as the reference ships them (eval.py:40-45), or - when the feeder already holds the model's 16-bit operand type
        (cfg['feature_dtype']) - unchanged: half the PCIe bytes per frame, no conversion on either side"""
        keep = {"fp16": torch.float16, "bf16": torch.bfloat16}.get(getattr(model, "compute_dtype", None))
        return x.contiguous() if (keep is not None and x.dtype == keep) else x.float().contiguous()

    @staticmethod
    def _json_int_lists(output, as_parts: bool = False):
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
        return parts if as_parts else b"{" + b", ".join(parts) + b"}"

    def _targets_to_device(self, targets, dev):
        """The loader's target rows [T, C] of a batch's videos -> device tensors, enqueued on the current (copy) stream.  Host fp32
        one-hot targets (what the reference's dataset yields) travel as ONE class id per frame: `prego_onehot_labels` reduces them in the
        loader's memory (a few host threads, while the GPU is busy with the features) and says per video whether the ids tell everything
        the rows do - 4 bytes per frame over the link instead of 4 x classes (0.79 GB for the 182-video eval set: 14 ms of a 200 ms
        pass).  Such a video's entry is an int32 vector [T]; any other video's rows are copied as they are ([T, C])."""
        out = [None] * len(targets)
        host = [i for i, t in enumerate(targets) if t.device.type == "cpu" and t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous()
                and t.shape[1] == len(self.all_class_names)]
        if host and self.cfg.get("eval_label_targets", True):
            import ctypes as C
            from . import _lib
            from ._lib import check
            nv = len(host)
            ptrs = (C.c_void_p * nv)(*[targets[i].data_ptr() for i in host])
            rows = (C.c_int64 * nv)(*[int(targets[i].shape[0]) for i in host])
            total = sum(int(targets[i].shape[0]) for i in host)
            labels = torch.empty(max(total, 1), dtype=torch.int32, pin_memory=dev.type == "cuda")
            flags = (C.c_int32 * nv)()
            check(_lib.load().prego_onehot_labels(nv, ptrs, rows, len(self.all_class_names), C.c_void_p(labels.data_ptr()), flags))
            labels_dev = labels.to(dev, non_blocking=True)
            self._keep_labels = labels                       # the pinned source of an async copy
            o = 0
            for k, i in enumerate(host):
                n = int(targets[i].shape[0])
                if flags[k]:
                    out[i] = labels_dev[o:o + n]
                o += n
        for i, t in enumerate(targets):
            if out[i] is None:
                out[i] = t.to(dev, non_blocking=True)
        return out

    def _gt_matrix(self, t):
        """a video's ground truth as the [T, C] matrix the loader gave (class-id vectors of _targets_to_device expanded again)"""
        if t.dim() == 2:
            return t
        m = torch.zeros((t.shape[0], len(self.all_class_names)), dtype=torch.float32, device=t.device)
        m[torch.arange(t.shape[0], device=t.device), t.long()] = 1
        return m

    def _enqueue(self, model, sub, device):
        """H2D of one sub-batch on the copy stream + its forward on the compute stream; nothing here waits for the GPU"""
        dev = torch.device(device)
        flows = [b[1] for b in sub]
        if all(f is None for f in flows):
            flows = None
        if dev.type == "cuda":
            # H2D on a side stream: these copies run while whatever was enqueued before is still computing (the loader's
            # pin_memory=True makes them true async DMA); the compute stream waits on one event per sub-batch
            if self._copy_stream is None:
                self._copy_stream = self._new_copy_stream(dev)
            cur = torch.cuda.current_stream(dev)
            with torch.cuda.stream(self._copy_stream):
                rgb = [b[0].to(dev, non_blocking=True) for b in sub]
                flow = None if flows is None else [None if f is None else f.to(dev, non_blocking=True) for f in flows]
                tgt = self._targets_to_device([b[2] for b in sub], dev)
                ready = torch.cuda.Event()
                ready.record(self._copy_stream)
            cur.wait_event(ready)
            for t in rgb + [f for f in (flow or []) if f is not None] + tgt:
                t.record_stream(cur)         # allocated on the copy stream, consumed on the compute stream
        else:
            rgb = [b[0].to(device) for b in sub]
            flow = None if flows is None else [None if f is None else f.to(device) for f in flows]
            tgt = self._targets_to_device([b[2] for b in sub], dev)
        probs, args, _ = model.forward_clips(rgb, flow, want_probs=True, want_argmax=True)
        return probs, args, tgt

    # link-fed eval: frames per H2D piece; one feed event per ~EVENT_BYTES copied.  A copy costs ~10 us of fixed time whatever its size, so
    # large pieces keep the link busier (182-video set, 16-bit features: 2 048 frames 226 ms to the last id, 4 096: 220, 8 192: 215.5) -
    # but the recurrence cannot pass a piece's first step before the piece has landed, so whatever a slot's LAST piece covers is run after
    # the link has gone idle: the tail of every video goes in TAIL_PIECE_FRAMES pieces
    PIECE_FRAMES = 8192
    TAIL_PIECE_FRAMES = 2048
    EVENT_BYTES = 64 << 20

    def _pieces_of(self, T):
        """frame ranges [a, b) one video's features are copied in"""
        P, Q = self.PIECE_FRAMES, min(self.TAIL_PIECE_FRAMES, self.PIECE_FRAMES)
        out, a = [], 0
        while T - a > P + Q:
            out.append((a, a + P))
            a += P
        while a < T:
            out.append((a, min(a + Q, T)))
            a += Q
        return out

    def _enqueue_link_fed(self, model, batch, device):
        """ONE forward over the whole batch while its features are still arriving (MROAD.link_fed_eval; pinned host features).
        The engine says at which step every video starts in its recurrence slot (plan_starts: a schedule costed for a link-bound
        feed - about frames / longest-video slots, every slot alive to the end, so rows are needed at the rate the link delivers
        them); the features are cut into pieces of PIECE_FRAMES frames, copied in the order the packed pipeline needs them
        (piece (video, a) at step start[video] + a), and a feed event is recorded every EVENT_BYTES; the library's packing stream waits
        for the events a chunk needs (set_feed_events).  The recurrence's critical path - the longest video - is paid once, with the
        copy under it, instead of copy + forward one after the other (round 3: 46 % of the PCIe floor)."""
        dev = torch.device(device)
        eng = model.engine()
        if self._copy_stream is None:
            self._copy_stream = self._new_copy_stream(dev)
        cur = torch.cuda.current_stream(dev)
        lens = [int(b[0].shape[0]) for b in batch]
        any_flow = any(b[1] is not None for b in batch)
        row_bytes = sum(int(t.shape[1]) * t.element_size() for t in (batch[0][0],) + ((next(b[1] for b in batch if b[1] is not None),) if any_flow else ()))
        start, n_steps = eng.plan_starts(lens, row_bytes)
        pieces = sorted((start[i] + a, i, a, b_) for i, T in enumerate(lens) for a, b_ in self._pieces_of(T))
        upto, events, acc = [], [], 0
        with torch.cuda.stream(self._copy_stream):
            # the destination tensors belong to the COPY stream (allocated inside its context): the copies of this batch follow the
            # previous batch's copies at once - they do not wait for the compute stream, which is still running the previous forward
            rgb = [torch.empty(b[0].shape, dtype=b[0].dtype, device=dev) for b in batch]
            flow = [None if b[1] is None else torch.empty(b[1].shape, dtype=b[1].dtype, device=dev) for b in batch] if any_flow else None
            for k, (need, i, a, b) in enumerate(pieces):
                rgb[i][a:b].copy_(batch[i][0][a:b], non_blocking=True)
                acc += (b - a) * int(rgb[i].shape[1]) * rgb[i].element_size()
                if flow is not None and flow[i] is not None:
                    flow[i][a:b].copy_(batch[i][1][a:b], non_blocking=True)
                    acc += (b - a) * int(flow[i].shape[1]) * flow[i].element_size()
                last = k + 1 == len(pieces)
                if acc >= self.EVENT_BYTES or last:
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                    events.append(ev)
                    upto.append(2 ** 31 - 1 if last else pieces[k + 1][0])      # every piece needed before that step has been copied
                    acc = 0
        for t in rgb + [f for f in (flow or []) if f is not None]:
            t.record_stream(cur)                               # allocated on the copy stream (inside its context), read on `cur`
        self._feed_events = events                           # keep the hipEvent_t objects alive until the stream has used them
        eng.set_feed_events(upto, events, row_bytes)
        try:
            probs, args, _ = model.forward_clips(rgb, flow, want_probs=True, want_argmax=True)
        except BaseException:
            eng.set_feed_events([], [], 0)                     # a forward that never reached the library must not leave its feed events armed for the next one
            raise
        # the targets are only needed behind the forward: reduced / copied now that everything else is enqueued, behind the features
        with torch.cuda.stream(self._copy_stream):
            tgt = self._targets_to_device([b[2] for b in batch], dev)
            ready = torch.cuda.Event()
            ready.record(self._copy_stream)
        for t in tgt:
            t.record_stream(cur)
        cur.wait_event(ready)
        return probs, args, tgt

    def _flush(self, model, batch, device):
        """enqueue one batch (H2D + forward + argmax ids D2H): nothing here waits for the GPU.  Returns the record _collect finishes."""
        if not batch:
            return None
        # Pinned host features + a model that can be fed while it runs: ONE link-fed forward (_enqueue_link_fed).  Otherwise a batch
        # runs as long as its longest video's recurrence and its H2D copy would sit in front of that, so large batches go in two
        # sub-batches ordered by length: the few longest videos first (little to copy, the long critical path), then the bulk of
        # the bytes, whose copy runs under the first sub-batch's recurrence.  Results return to the loader's order below.
        order = list(range(len(batch)))
        parts = [order]
        frames = [int(b[0].shape[0]) for b in batch]
        on_gpu = torch.device(device).type == "cuda"
        link_fed = on_gpu and bool(getattr(model, "link_fed_eval", False)) and bool(self.cfg.get("eval_link_fed", True)) and len(batch) >= 2 and \
            all(b[0].device.type == "cpu" and b[0].is_pinned() and (b[1] is None or (b[1].device.type == "cpu" and b[1].is_pinned())) for b in batch)
        if len(batch) >= 16 and on_gpu and not link_fed and self.cfg.get("eval_split_by_length", True):
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
        if link_fed:
            probs, args, tgt = self._enqueue_link_fed(model, batch, device)
            for i, (p, a, t) in enumerate(zip(probs, args, tgt)):
                res[i] = (p, a, t)
        else:
            for part in parts:
                probs, args, tgt = self._enqueue(model, [batch[i] for i in part], device)
                for i, p, a, t in zip(part, probs, args, tgt):
                    res[i] = (p, a, t)
        want_json = self.cfg["eval"] is not None
        # np.argmax of the one-hot targets (eval.py:55) on the device, where they are anyway for the AP kernel (on the host it reads
        # frames x classes floats through one core: 25 ms for the bench set); ONE device -> host copy of the whole batch's pred and gt
        # ids is ENQUEUED here (into pinned memory, behind the forward) and waited for in _collect - one batch later, so that the next
        # batch's copies and forward are already running while the host waits for these ids and formats their text
        rec = {"items": [(vid, res[i]) for i, (r, f, target, vid) in enumerate(batch)], "ids": None, "ev": None, "link_fed": link_fed,
               "src": list(batch), "feed": getattr(self, "_feed_events", None)}       # the loader's tensors / the feed events stay alive until _collect
        if want_json:
            pred_ids = torch.cat([res[i][1] for i in range(len(batch))])
            gt_ids = torch.cat([res[i][2] if res[i][2].dim() == 1 else torch.argmax(res[i][2], dim=1)
                                for i in range(len(batch))]).to(pred_ids.dtype)
            ids_dev = torch.stack([pred_ids, gt_ids])
            if ids_dev.is_cuda:
                ids_host = torch.empty(ids_dev.shape, dtype=ids_dev.dtype, pin_memory=True)
                ids_host.copy_(ids_dev, non_blocking=True)
                if ids_dev.dtype == torch.int32 and len(self.all_class_names) <= 1000:
                    # the JSON text of the ids as well (prego_format_ids: four bytes "%3d," per id), so that behind the last frame
                    # the host only cuts it per video: formatting 2 x 2.3 M numbers through a numpy table took 10-15 ms there
                    from . import _lib
                    from ._lib import check
                    import ctypes as C
                    ids_dev = ids_dev.contiguous()
                    text_dev = torch.empty(ids_dev.shape, dtype=torch.int32, device=ids_dev.device)
                    bad_dev = torch.zeros(1, dtype=torch.int32, device=ids_dev.device)
                    with torch.cuda.device(ids_dev.device):
                        check(_lib.load().prego_format_ids(C.c_void_p(ids_dev.data_ptr()), ids_dev.numel(), C.c_void_p(text_dev.data_ptr()),
                                                           C.c_void_p(bad_dev.data_ptr()),
                                                           C.c_void_p(torch.cuda.current_stream(ids_dev.device).cuda_stream)))
                    text_host = torch.empty(ids_dev.shape, dtype=torch.int32, pin_memory=True)
                    bad_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                    text_host.copy_(text_dev, non_blocking=True)
                    bad_host.copy_(bad_dev, non_blocking=True)
                    rec["text"], rec["bad"], rec["_keep_text"] = text_host, bad_host, (text_dev, bad_dev)
                ev = torch.cuda.Event()
                ev.record()
                rec["ids"], rec["ev"], rec["_keep"] = ids_host, ev, ids_dev
            else:                                  # a stand-in model on a CPU box (the gloo tests of the sharding logic)
                rec["ids"] = ids_dev
        batch.clear()
        return rec

    def _collect(self, rec, pred_scores, gt_targets, output, json_parts):
        """the host half of a batch, one batch behind its launch: wait for its ids, cut them per video, format the videos' JSON text"""
        if rec is None:
            return
        ids = None
        if rec["ids"] is not None:
            if rec["ev"] is not None:
                rec["ev"].synchronize()
            ids = rec["ids"].numpy()
        elif rec["link_fed"]:
            self._copy_stream.synchronize()      # the copies read the loader's pinned tensors: keep them alive until then
        text = None
        if ids is not None and rec.get("text") is not None and int(rec["bad"][0]) == 0:
            text = rec["text"].numpy().view(np.uint8).reshape(2, -1, 4)      # [pred | gt][frame] = b"%3d," (prego_format_ids)
        o = 0
        for vid, (p, a, t) in rec["items"]:
            if ids is not None:
                n = int(a.shape[0])
                output[vid] = {"pred": ids[0, o:o + n], "gt": ids[1, o:o + n]}      # int arrays; text below
                if json_parts is not None:
                    if json_parts:
                        json_parts.append(b", ")
                    if text is not None and n > 0:
                        # the video's slice of the device-made text, in place: every list's last comma becomes its bracket
                        tp, tg = text[0, o:o + n], text[1, o:o + n]
                        tp[-1, 3] = tg[-1, 3] = 0x5D
                        json_parts += [json.dumps(str(vid)).encode() + b': {"pred": [', tp, b', "gt": [', tg, b"}"]
                    else:
                        json_parts += self._json_int_lists({vid: output[vid]}, as_parts=True)
                o += n
                self.last_device_argmax[vid] = a          # int32 on the device: input of aggregate_device (utils/aggregate.py)

    def _cat_targets(self, gt_targets, matrix=False):
        """the eval set's ground truth: one class-id vector [frames] when every video came as class ids (and the caller takes them), else
        the [frames, C] matrix"""
        if not matrix and all(t.dim() == 1 for t in gt_targets):
            return torch.cat(gt_targets, 0)
        return torch.cat([self._gt_matrix(t) for t in gt_targets], 0)

    @staticmethod
    def _scores(rec, pred_scores, gt_targets):
        """the [T, C] score and target matrices of a batch's videos, taken at launch time: they stay torch tensors on the model's device
        (one entry per video, concatenated once at the end; the reference extends Python lists by one row object per frame,
        eval.py:46-49), so the metric's kernels can be enqueued behind the last forward before the host waits for anything"""
        if rec is not None:
            for _vid, (p, _a, t) in rec["items"]:
                pred_scores.append(p)
                gt_targets.append(t)

    def eval(self, model, dataloader, logger, device):
        model.eval()
        output = {}
        self.last_device_argmax = {}
        max_clips = model.max_clips          # MROAD: the engine's clip capacity; ViTEnc (`model: 'Transformer'`): its own bound
        # data-parallel eval: videos are independent, so under torch.distributed rank r takes every world-th video of
        # the loader's order (no data-path collective); rank 0 gathers the per-video results once at the end
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        skip_flow = self._zero_flow(model, dataloader)
        with torch.no_grad():
            pred_scores, gt_targets = [], []
            per_video = []                       # (loader position, n_frames) to restore the loader's order on rank 0
            t_begin = time.time()
            t0_ = time.perf_counter()
            self.phase_log = []                  # (phase, host seconds since the start of this eval): where an end-to-end eval spends its time
            mark = lambda name: self.phase_log.append((name, time.perf_counter() - t0_))
            batch, frames, pos = [], 0, 0
            pending = None                       # the batch whose launch is out and whose host half (_collect) is still due
            json_parts = [] if (self.cfg["eval"] is not None and world == 1) else None      # per-video JSON text, formatted batch by batch
            for rgb_input, flow_input, target, vid, start, end in dataloader:
                # loader items carry a leading batch dim of test_batch_size == 1 (dataset_builder.py:19)
                for b in range(rgb_input.shape[0]):
                    mine = (pos % world) == rank
                    pos += 1
                    if not mine:
                        continue
                    name = vid[b] if isinstance(vid, (list, tuple)) else vid
                    fl = None if (skip_flow or self._flow_zero(model, flow_input[b])) else self._features(model, flow_input[b])
                    batch.append((self._features(model, rgb_input[b]), fl, target[b], name))
                    per_video.append((pos - 1, int(rgb_input.shape[1])))
                    frames += rgb_input.shape[1]
                if len(batch) >= max_clips or frames >= self.max_frames_per_batch:
                    mark("loader")
                    rec = self._flush(model, batch, device)                 # batch k: enqueued ...
                    self._scores(rec, pred_scores, gt_targets)
                    mark("launch")
                    self._collect(pending, pred_scores, gt_targets, output, json_parts)      # ... while the host finishes batch k - 1
                    mark("collect")
                    pending = rec
                    frames = 0
            mark("loader")
            rec = self._flush(model, batch, device)
            self._scores(rec, pred_scores, gt_targets)
            finish_ap = None
            if world == 1 and torch.device(device).type == "cuda" and self.metric == "AP" and pred_scores:
                # the metric in libprego_amd.so (prego_perframe_ap), ENQUEUED here behind the last forward and collected behind the
                # output file: the GPU ranks the positives while the host waits for the ids and writes the JSON
                # ... on a stream of its own behind an event, so that model.check() below (which drains the forward's stream to read its
                # timeout word) does not wait for the metric as well
                if self._ap_stream is None:
                    self._ap_stream = torch.cuda.Stream(torch.device(device))
                fwd_done = torch.cuda.Event()
                fwd_done.record(torch.cuda.current_stream(torch.device(device)))
                with torch.cuda.stream(self._ap_stream):
                    self._ap_stream.wait_event(fwd_done)
                    pred_all = torch.cat(pred_scores, 0)
                    gt_all = self._cat_targets(gt_targets).to(pred_all.device)
                    ap_fn = perframe_average_precision_device(pred_all.to(device), gt_all.to(device), self.all_class_names,
                                                              self.data_processing, self.metric, defer=True)

                def finish_ap(_fn=ap_fn, _s=self._ap_stream):
                    with torch.cuda.stream(_s):           # the result's device -> host copy follows the kernels on THEIR stream
                        return _fn()
            mark("launch")
            self._collect(pending, pred_scores, gt_targets, output, json_parts)
            mark("collect")
            self._collect(rec, pred_scores, gt_targets, output, json_parts)
            mark("collect_last")
            model.check()
            mark("check")
            if world > 1:
                # the small things go to rank 0 as objects: per-video frame counts and the pred / gt id arrays of the output file
                # (8 bytes per frame).  The [frames x classes] score and target matrices do NOT travel whole (round 3 pickled 0.8 GB of
                # them through gather_object): the metric is computed class-sharded, see _sharded_ap
                gathered = [None] * world if rank == 0 else None
                dist.gather_object((per_video, output), gathered, dst=0)
                if rank == 0:                       # back into the loader's order (a rank's dict is in the order of its per_video list)
                    entries = []
                    for pv, out in gathered:
                        entries += [(p_idx, vid, ent) for (p_idx, _n), (vid, ent) in zip(pv, out.items())]
                    output = {vid: ent for _p, vid, ent in sorted(entries, key=lambda e: e[0])}
                pred_local = torch.cat(pred_scores, 0) if pred_scores else torch.zeros((0, len(self.all_class_names)), device=device)
                gt_local = self._cat_targets(gt_targets).to(pred_local.device) if gt_targets else torch.zeros_like(pred_local)
                result = self._sharded_ap(pred_local, gt_local, world, rank)
                if rank == 0 and self.cfg["eval"] is not None:
                    os.makedirs(self.output_dir, exist_ok=True)
                    with open(os.path.join(self.output_dir, "output_miniROAD.json"), "wb") as file:
                        file.write(self._json_int_lists(output))
                t_end = time.time()
                nf = torch.tensor([int(pred_local.shape[0])], dtype=torch.int64, device=pred_local.device)
                dist.all_reduce(nf)
                num_frames = int(nf.item())
                self.last_fps = num_frames / max(t_end - t_begin, 1e-9)
                if rank == 0:
                    logger.info(f"Processed {num_frames} frames in {t_end - t_begin:.1f} seconds ({self.last_fps:.1f} FPS)")
                return result["mean_AP"]
            if finish_ap is None:
                pred_all = torch.cat(pred_scores, 0) if pred_scores else torch.zeros((0, len(self.all_class_names)))
                gt_all = self._cat_targets(gt_targets, matrix=True).to(pred_all.device) if gt_targets else torch.zeros_like(pred_all)
                if torch.device(device).type == "cuda" and self.metric == "AP":        # an empty eval set: the empty report
                    finish_ap = perframe_average_precision_device(pred_all.to(device), gt_all.to(device), self.all_class_names,
                                                                  self.data_processing, self.metric, defer=True)
            num_frames = int(gt_all.shape[0])
            if self.cfg["eval"] is not None:
                os.makedirs(self.output_dir, exist_ok=True)
                with open(os.path.join(self.output_dir, "output_miniROAD.json"), "wb") as file:
                    if json_parts is not None:          # per-video pieces (bytes and slices of the device-made text), ", " between videos
                        file.write(b"{")
                        file.writelines(json_parts)
                        file.write(b"}")
                    else:
                        file.write(self._json_int_lists(output))
            mark("json_written")
            if finish_ap is not None:
                result = finish_ap()
                mark("ap_done")
            else:       # a stand-in model on a CPU box (the gloo tests of the sharding logic), or metric 'cAP' (TVSeries' calibrated variant: host path only)
                result = perframe_average_precision(pred_all.cpu().numpy(), gt_all.cpu().numpy(), self.all_class_names, self.data_processing, self.metric)
            t_end = time.time()
            time_taken = max(t_end - t_begin, 1e-9)
            self.last_fps = num_frames / time_taken
            logger.info(f"Processed {num_frames} frames in {time_taken:.1f} seconds ({self.last_fps:.1f} FPS)")
        return result["mean_AP"]

    def _sharded_ap(self, pred, gt, world, rank):
        """per-frame AP over the frames of ALL ranks without gathering the [frames x classes] matrices anywhere: rank q owns the classes
        [C q / world, C (q + 1) / world) of every frame - one all_to_all_single per matrix brings it those columns from every rank
        (each rank sends (world - 1) / world of its matrix once, RCCL over xGMI on a GPU node) -, runs the AP kernel on them, and only
        the three per-class vectors (AP, positives, score mass) are all-gathered: 24 bytes per class.  AP per class does not depend
        on the order of the frames (exact integer ranks, ties share a threshold), so the result equals the single-process one."""
        C_ = len(self.all_class_names)
        cb = [C_ * q // world for q in range(world + 1)]
        n_local = torch.tensor([int(pred.shape[0])], dtype=torch.int64, device=pred.device)
        ns = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(ns, n_local)
        ns = [int(x.item()) for x in ns]
        mine = cb[rank + 1] - cb[rank]

        def exchange(m):
            m = m.to(torch.float32)
            send = torch.cat([m[:, cb[q]:cb[q + 1]].contiguous().reshape(-1) for q in range(world)]) if m.numel() else m.reshape(-1)
            recv = torch.empty(sum(ns) * mine, dtype=torch.float32, device=m.device)
            dist.all_to_all_single(recv, send, output_split_sizes=[n * mine for n in ns],
                                   input_split_sizes=[int(m.shape[0]) * (cb[q + 1] - cb[q]) for q in range(world)])
            return recv.reshape(sum(ns), mine) if mine else recv.reshape(sum(ns), 0)
        p_cols = exchange(pred)
        by_id = torch.tensor([int(gt.dim() == 1)], dtype=torch.int64, device=pred.device)
        dist.all_reduce(by_id, op=dist.ReduceOp.MIN)                 # class ids only if EVERY rank holds ids (a rank without videos: a matrix of no rows)
        if int(by_id.item()):
            # one-hot ground truth held as one class id per frame (_targets_to_device): every rank gets every frame's id (4 bytes per
            # frame, padded to the longest rank) instead of its columns of the target matrix; an id outside my columns = no positive there
            width_n = max(ns) if ns else 0
            buf = torch.full((max(width_n, 1),), -1, dtype=torch.int32, device=pred.device)
            buf[:gt.shape[0]] = gt.to(torch.int32)
            bufs = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(bufs, buf)
            ids = torch.cat([bufs[q][:ns[q]] for q in range(world)]) - cb[rank]
            g_cols = ids
            if not (p_cols.is_cuda and self.metric == "AP"):
                g_cols = torch.zeros((sum(ns), mine), dtype=torch.float32, device=pred.device)
                ok = (ids >= 0) & (ids < mine)
                g_cols[torch.arange(sum(ns), device=pred.device)[ok], ids[ok].long()] = 1
        else:
            g_cols = exchange(self._gt_matrix(gt))
        if mine == 0:
            raw = (np.zeros(0), np.zeros(0, np.int64), np.zeros(0))
        elif p_cols.is_cuda and self.metric == "AP":
            raw = perframe_ap_raw_device(p_cols, g_cols)
        else:
            raw = perframe_ap_raw(p_cols.cpu().numpy(), g_cols.cpu().numpy(), self.metric)
        width = max(cb[q + 1] - cb[q] for q in range(world))
        pack = torch.zeros((3, width), dtype=torch.float64, device=pred.device)
        pack[0, :mine] = torch.from_numpy(np.asarray(raw[0], dtype=np.float64))
        pack[1, :mine] = torch.from_numpy(np.asarray(raw[1], dtype=np.float64))          # counts < 2^53: exact in fp64
        pack[2, :mine] = torch.from_numpy(np.asarray(raw[2], dtype=np.float64))
        parts = [torch.zeros_like(pack) for _ in range(world)]
        dist.all_gather(parts, pack)
        ap = np.concatenate([parts[q][0, :cb[q + 1] - cb[q]].cpu().numpy() for q in range(world)])
        npos = np.concatenate([parts[q][1, :cb[q + 1] - cb[q]].cpu().numpy() for q in range(world)]).astype(np.int64)
        ssum = np.concatenate([parts[q][2, :cb[q + 1] - cb[q]].cpu().numpy() for q in range(world)])
        return report_from_raw(ap, npos, ssum, self.all_class_names)

    def aggregate_last(self, output_path=None, window_size: int = 200):
        """utils/aggregate.py:46-90 on the per-frame argmax the last eval left in HBM (`last_device_argmax`), with this config's
        number of classes; ground truth from the output JSON's "gt" lists"""
        from .aggregate import aggregate_device
        js = json.load(open(os.path.join(self.output_dir, "output_miniROAD.json")))
        return aggregate_device(self.last_device_argmax, {k: v["gt"] for k, v in js.items()}, output_path, window_size,
                                n_classes=len(self.all_class_names))

    def forward(self, model, dataloader, logger, device):
        return self.eval(model, dataloader, logger, device)
