"""Thin host wrapper around the C ABI: torch supplies device memory and streams,
the arithmetic runs in libprego_amd.so."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import PregoError, check, ptr_array

_PARAM_ORDER = [
    "layer1.0.weight", "layer1.0.bias", "layer1.1.weight", "layer1.1.bias",
    "gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0",
    "f_classification.0.weight", "f_classification.0.bias",
]
_L1_KEYS = ["gru.weight_ih_l1", "gru.weight_hh_l1", "gru.bias_ih_l1", "gru.bias_hh_l1"]       # nn.GRU(num_layers=2), rnn.py:32,38


def param_order(num_layers: int = 1):
    """the parameters a training step differentiates, in the flat gradient bucket's order: layer1 / LayerNorm | GRU layer 0 (| GRU layer
    1) | head - the order the backward finishes its three sub-buckets in, reversed"""
    if num_layers == 1:
        return list(_PARAM_ORDER)
    k = _PARAM_ORDER.index("f_classification.0.weight")
    return _PARAM_ORDER[:k] + _L1_KEYS + _PARAM_ORDER[k:]



def plan_passes(lens, max_clips: int, single: bool = False, max_slots: int = 512):
    """Split a clip list into forward() calls.  The C ABI packs up to `max_clips` clips per call into its recurrence
    slots itself (continuous batching: longest-first bin packing, see csrc/miniroad.cpp::build_plan); only calls that
    need one clip per slot (h0 / h_last) are limited to `max_slots` clips.  Returns lists of clip indices."""
    n = len(lens)
    cap = max_slots if single else max_clips
    if n <= cap:
        return [list(range(n))]
    order = sorted(range(n), key=lambda i: (-lens[i], i))
    return [order[s:s + cap] for s in range(0, n, cap)]


def _dev_readable(t: torch.Tensor) -> bool:
    """feature tensors the pack kernel can read: device memory, or PINNED host memory (hipHostMalloc: mapped into the device's address
    space at the same address) - the kernel then pulls the rows over PCIe itself, chunk by chunk, under the previous chunk's recurrence
    (zero-copy feeding of Evaluate: no staging copy, no H2D in front of the forward).  The caller keeps a pinned tensor alive until
    the stream has run the call."""
    return t.is_cuda or (t.device.type == "cpu" and t.is_pinned())


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class MiniRoadEngine:
    """One handle per (device, model, compute_dtype).  compute_dtype: 'fp16' / 'bf16' (16-bit MFMA operands, fp32 accumulation; fp16 =
    same speed, 8x less operand rounding, inference only) or 'fp32' (exact-fp32 MFMA, the parity mode)."""

    def __init__(self, d_rgb: int, d_flow: int, emb: int, hid: int, n_classes: int, device,
                 compute_dtype: str = "bf16", lib=None, num_layers: int = 1):
        # lib: tests that inject faults run a handle on libprego_amd_debug.so (_lib.load_debug()); everything else runs the product library
        self.lib = lib if lib is not None else _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise PregoError("prego_amd runs on an MI355X only (device must be cuda:N); there is no CPU path")
        self.dims = (d_rgb, d_flow, emb, hid, n_classes)
        self.compute_dtype = compute_dtype
        h = C.c_void_p()
        codes = {"bf16": _lib.PREGO_BF16, "fp16": _lib.PREGO_F16, "fp32": _lib.PREGO_F32, "fp16x2": _lib.PREGO_F16X2}
        if compute_dtype not in codes:
            raise PregoError(f"compute_dtype {compute_dtype!r}: expected one of {sorted(codes)}")
        with torch.cuda.device(self.device):
            check(self.lib.prego_miniroad_create_layers(C.byref(h), d_rgb, d_flow, emb, hid, n_classes, int(num_layers), codes[compute_dtype]))
        self.h = h
        self.num_layers = int(num_layers)
        self.max_clips = self.lib.prego_miniroad_max_clips(self.h)
        self._ws: Optional[torch.Tensor] = None
        self._res: Optional[torch.Tensor] = None          # whole-call relu(h) buffer (prego_miniroad_set_resident): torch's allocator owns it
        # fp16x2 rows are 34 KB (fp32 intermediates, split operands): 32 768 rows measured best there (262.6 against 269.4 ms per pass
        # at 49 152; scripts/probes/chunk_sweep3.sh); the 16-bit modes are flat between 32 768 and 49 152 (120.2 / 120.4 ms)
        self.rows_per_chunk = 32768 if compute_dtype == "fp16x2" else 49152
        #                                  packed rows per chunk: 192 M tiles = whole rounds of 256x256 tiles on 256 CUs for N = 2048 (6) and 3072 (9);
                                         # X + GI of a chunk (1 GB) stay close to the 256 MB Infinity Cache (sweep 32 k..256 k rows: 137..142 ms)
        self._weights_version = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.prego_miniroad_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- weights ---------------------------------------------------------------------------
    def set_weights(self, sd: dict):
        ts = []
        for k in _PARAM_ORDER:
            t = sd[k].detach()
            if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
                t = t.to(self.device, torch.float32).contiguous()
            ts.append(t)
        with torch.cuda.device(self.device):
            check(self.lib.prego_miniroad_set_weights(self.h, *[C.c_void_p(t.data_ptr()) for t in ts],
                                                      C.c_void_p(_stream_ptr(self.device))))
            for l in range(1, self.num_layers):          # stacked GRU (rnn.py:38): gru.*_l1
                tl = []
                for k in (f"gru.weight_ih_l{l}", f"gru.weight_hh_l{l}", f"gru.bias_ih_l{l}", f"gru.bias_hh_l{l}"):
                    t = sd[k].detach()
                    if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
                        t = t.to(self.device, torch.float32).contiguous()
                    tl.append(t)
                check(self.lib.prego_miniroad_set_gru_layer(self.h, l, *[C.c_void_p(t.data_ptr()) for t in tl], C.c_void_p(_stream_ptr(self.device))))
                ts += tl
        self._keep = ts   # alive until the stream has consumed them

    # -- forward ---------------------------------------------------------------------------
    def _workspace(self, n_clips, lens_arr, flags):
        need = self.lib.prego_miniroad_workspace_bytes(self.h, n_clips, lens_arr, self.rows_per_chunk, flags)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def _resident(self, n_clips, lens_arr, flags):
        """The whole-call buffer a long inference call needs for the once-per-pass classifier / the split pass (include/prego_amd.h,
        ABI 7): sized by the library's query, allocated HERE by torch's caching allocator (stream-ordered: no hipMalloc and no
        synchronisation inside prego_miniroad_forward), grown with 1/8 of slack and kept for the following calls."""
        need = self.lib.prego_miniroad_resident_bytes(self.h, n_clips, lens_arr, flags)
        if need and (self._res is None or self._res.numel() < need):
            try:
                res = torch.empty(need + need // 8, dtype=torch.uint8, device=self.device)
            except torch.OutOfMemoryError:
                return            # no room beside the caller's tensors: the call runs the chunked pass with the per-chunk classifier
            check(self.lib.prego_miniroad_set_resident(self.h, C.c_void_p(res.data_ptr()), res.numel()))
            self._res = res       # the old buffer goes back to the allocator of the same stream: work already enqueued on it finishes first

    def forward_ragged(self, rgb: Sequence[torch.Tensor], flow: Optional[Sequence[Optional[torch.Tensor]]],
                       softmax: bool = True, want_out: bool = True, want_argmax: bool = False,
                       h0: Optional[torch.Tensor] = None, want_h_last: bool = False):
        """rgb[i]: fp32 cuda (or pinned host, see _dev_readable) [T_i, d_rgb] contiguous; flow[i] likewise or None (= zeros).
        Returns (outs list of [T_i, C] or None, argmax list of int32 [T_i] or None, h_last or None)."""
        d_rgb, d_flow, emb, hid, ncls = self.dims
        if d_rgb == 0:                  # --no_rgb (rnn.py:23-29,54-57): the model's only input is the flow stream
            if flow is None or any(f is None for f in flow):
                raise PregoError("a --no_rgb model needs a flow tensor for every clip")
            rgb = None
        src_list = rgb if rgb is not None else flow
        n = len(src_list)
        if n == 0:                      # empty clip list: nothing to do (the reference's loader simply yields nothing)
            return ([] if want_out else None), ([] if want_argmax else None), (torch.empty((0, hid), device=self.device) if want_h_last else None)
        outs = [None] * n
        args = [None] * n
        # GRU state: [n, hid]; a two-layer engine takes / returns nn.GRU's [layers, n, hid]
        L = self.num_layers
        st_shape = (lambda m: (m, hid) if L == 1 else (L, m, hid))
        if h0 is not None and tuple(h0.shape) != st_shape(n):
            raise PregoError(f"h0 {tuple(h0.shape)}: expected {st_shape(n)}")
        h_last = torch.empty(st_shape(n), dtype=torch.float32, device=self.device) if want_h_last else None
        lens = [int(r.shape[0]) for r in src_list]
        single = h0 is not None or want_h_last
        for idx in plan_passes(lens, self.max_clips, single, {"fp32": 256, "fp16x2": 128}.get(self.compute_dtype, 512)):
            sub_h0 = None if h0 is None else h0[..., idx, :].to(torch.float32).contiguous()
            sub_hl = None if h_last is None else torch.empty(st_shape(len(idx)), dtype=torch.float32, device=self.device)
            sub_out, sub_arg = [None] * len(idx), [None] * len(idx)
            self._forward_pass(None if rgb is None else [rgb[i] for i in idx], None if flow is None else [flow[i] for i in idx], softmax,
                               want_out, want_argmax, sub_h0, sub_hl, sub_out, sub_arg, 0)
            for k, i in enumerate(idx):
                outs[i], args[i] = sub_out[k], sub_arg[k]
            if h_last is not None:
                h_last[..., idx, :] = sub_hl
        return (outs if want_out else None), (args if want_argmax else None), h_last

    # -- link-fed inference (Evaluate: the H2D copy of a batch under its forward) ------------------------------------------
    def plan_starts(self, lens: Sequence[int], link_row_bytes: int):
        """(start_step per clip, number of steps) of the schedule the next forward_ragged of exactly these clips will use when its
        features arrive over the link at `link_row_bytes` per frame: frame a of clip i is consumed at step start_step[i] + a"""
        n = len(lens)
        arr = (C.c_int32 * n)(*[int(l) for l in lens])
        out = (C.c_int32 * n)()
        ns = C.c_int32(0)
        check(self.lib.prego_miniroad_plan_starts(self.h, n, arr, int(link_row_bytes), out, C.byref(ns)))
        return list(out), int(ns.value)

    def set_feed_events(self, upto_steps: Sequence[int], events: Sequence["torch.cuda.Event"], link_row_bytes: int):
        """for the NEXT forward_ragged call: rows needed at steps < upto_steps[j] are valid once events[0..j] have fired"""
        n = len(events)
        up = (C.c_int32 * n)(*[int(min(u, 2**31 - 1)) for u in upto_steps])
        ev = (C.c_void_p * n)(*[e.cuda_event for e in events])
        check(self.lib.prego_miniroad_set_feed_events(self.h, n, up, ev, int(link_row_bytes)))
        self._link_fed_next = True          # a link-fed call keeps the per-chunk classifier: it needs no resident buffer

    def _forward_pass(self, rgb, flow, softmax, want_out, want_argmax, h0, h_last, outs, args, base):
        d_rgb, d_flow, emb, hid, ncls = self.dims
        n = len(rgb) if rgb is not None else len(flow)
        lens = []
        # features: fp32, or - for a bf16 / fp16 engine - tensors that already hold its operand type (PREGO_FWD_IN16: a feeder that
        # keeps 16-bit features in pinned memory ships half the bytes; every clip of the call in the same dtype)
        first = (rgb if rgb is not None else flow)[0]
        dt = first.dtype
        op_dt = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(self.compute_dtype)
        if dt != torch.float32 and dt != op_dt:
            raise PregoError(f"features are {dt}: a {self.compute_dtype} engine takes fp32" + (f" or {op_dt}" if op_dt else "") + " tensors")
        for i, r in enumerate(rgb if rgb is not None else flow):
            want = d_rgb if rgb is not None else d_flow
            if r.dtype != dt or not _dev_readable(r) or not r.is_contiguous() or r.dim() != 2 or r.shape[1] != want:
                raise PregoError(f"{'rgb' if rgb is not None else 'flow'}[{i}] must be a contiguous {dt} cuda (or pinned host) tensor [T, {want}], "
                                 f"got {tuple(r.shape)} {r.dtype} on {r.device}")
            lens.append(r.shape[0])
        if flow is not None:
            for i, f in enumerate(flow):
                if f is None:
                    continue
                if f.dtype != dt or not _dev_readable(f) or not f.is_contiguous() or tuple(f.shape) != (lens[i], d_flow):
                    raise PregoError(f"flow[{i}] must be a contiguous {dt} cuda tensor [{lens[i]}, {d_flow}]")
        lens_arr = (C.c_int32 * n)(*lens)
        flags = (_lib.FWD_SOFTMAX if softmax else 0) | (_lib.FWD_IN16 if dt != torch.float32 else 0)
        ws = self._workspace(n, lens_arr, flags)
        if h0 is None and h_last is None and not getattr(self, "_link_fed_next", False):
            self._resident(n, lens_arr, flags)
        self._link_fed_next = False
        rgb_p = None if rgb is None else ptr_array([r.data_ptr() for r in rgb])
        flow_p = None if flow is None else ptr_array([None if f is None else f.data_ptr() for f in flow])
        out_p = arg_p = None
        # one allocation per output kind, per-clip views into it (512 clips = 1 024 allocator calls otherwise: the host, not
        # the GPU, then bounds a pass of short clips)
        total = sum(lens)
        if want_out:
            flat = torch.empty((total, ncls), dtype=torch.float32, device=self.device)
            o, p0, ptrs = 0, flat.data_ptr(), []
            for i in range(n):
                outs[base + i] = flat[o:o + lens[i]]
                ptrs.append(p0 + o * ncls * 4)
                o += lens[i]
            out_p = ptr_array(ptrs)
        if want_argmax:
            flat_a = torch.empty((total,), dtype=torch.int32, device=self.device)
            o, p0, ptrs = 0, flat_a.data_ptr(), []
            for i in range(n):
                args[base + i] = flat_a[o:o + lens[i]]
                ptrs.append(p0 + o * 4)
                o += lens[i]
            arg_p = ptr_array(ptrs)
        with torch.cuda.device(self.device):
            check(self.lib.prego_miniroad_forward(
                self.h, n, lens_arr, rgb_p, flow_p, out_p, arg_p,
                C.c_void_p(h0.data_ptr()) if h0 is not None else None,
                C.c_void_p(h_last.data_ptr()) if h_last is not None else None,
                flags, C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(_stream_ptr(self.device))))

    # -- streaming (online use) ------------------------------------------------
    def step(self, rgb: Optional[torch.Tensor], flow: Optional[torch.Tensor], h: torch.Tensor, softmax: bool = True,
             out: Optional[torch.Tensor] = None, argmax: Optional[torch.Tensor] = None):
        """One new frame for each of n <= 16 independent streams: rgb [n, d_rgb] / flow [n, d_flow] (None = zero flow) fp32 cuda
        contiguous, h [n, hid] fp32 cuda = the GRU state, UPDATED IN PLACE (zeros before a stream's first frame).
        Returns (probabilities or logits [n, C], argmax int32 [n]); pass `out` / `argmax` to reuse buffers.  bf16 / fp16 engines run
        the three / four-launch fast path (prego_miniroad_step); fp32 / fp16x2 engines, hidden sizes other than 1024 and two-layer models
        (h: [layers, n, hid]) the general forward with h0 / h_last."""
        d_rgb, d_flow, emb, hid, ncls = self.dims
        src = rgb if d_rgb > 0 else flow
        if src is None:
            raise PregoError("step: a --no_rgb model needs the flow frame" if d_rgb == 0 else "step: rgb is None")
        n = src.shape[0]
        general = self.compute_dtype in ("fp32", "fp16x2") or hid != 1024 or self.num_layers != 1     # what the streaming kernels are not built for
        for t, d in ((rgb if d_rgb > 0 else None, d_rgb), (flow, d_flow)):
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.shape != (n, d)):
                raise PregoError(f"step: expected contiguous fp32 cuda [{n}, {d}], got {tuple(t.shape)} {t.dtype} on {t.device}")
        h_shape = (n, hid) if self.num_layers == 1 else (self.num_layers, n, hid)
        if not h.is_cuda or h.dtype != torch.float32 or not h.is_contiguous() or tuple(h.shape) != h_shape:
            raise PregoError(f"step: expected the GRU state as contiguous fp32 cuda {list(h_shape)}, got {tuple(h.shape)} {h.dtype} on {h.device}")
        if out is None:
            out = torch.empty((n, ncls), dtype=torch.float32, device=self.device)
        if argmax is None:
            argmax = torch.empty((n,), dtype=torch.int32, device=self.device)
        if general:
            rl = [rgb[i:i + 1] for i in range(n)] if d_rgb > 0 else None
            fl = [flow[i:i + 1] for i in range(n)] if flow is not None else None
            o, a, hl = self.forward_ragged(rl, fl, softmax=softmax, want_out=True, want_argmax=True, h0=h, want_h_last=True)
            h.copy_(hl)
            out.copy_(torch.cat(o))
            argmax.copy_(torch.cat(a))
            return out, argmax
        with torch.cuda.device(self.device):
            check(self.lib.prego_miniroad_step(
                self.h, n, C.c_void_p(rgb.data_ptr()) if (rgb is not None and d_rgb > 0) else None,
                C.c_void_p(flow.data_ptr()) if flow is not None else None, C.c_void_p(h.data_ptr()), C.c_void_p(out.data_ptr()),
                C.c_void_p(argmax.data_ptr()), _lib.FWD_SOFTMAX if softmax else 0, C.c_void_p(_stream_ptr(self.device))))
        return out, argmax

    # -- training ------------------------------------------------------------------------
    def set_dropout(self, p: float, seed: int):
        check(self.lib.prego_miniroad_set_dropout(self.h, float(p), int(seed) & 0xFFFFFFFFFFFFFFFF))

    def forward_train(self, rgb: torch.Tensor, flow: Optional[torch.Tensor]) -> torch.Tensor:
        """training-mode forward of a uniform batch [B,T,D]: raw logits [B,T,C]; keeps activations for backward()."""
        d_rgb, d_flow, emb, hid, ncls = self.dims
        if d_rgb == 0:
            rgb = None
        src = rgb if rgb is not None else flow
        B, T = src.shape[0], src.shape[1]
        if B > self.max_clips:
            raise PregoError(f"training batch {B} > {self.max_clips} clips per call")
        rgb = None if rgb is None else rgb.contiguous()
        flow = None if flow is None else flow.contiguous()
        lens_arr = (C.c_int32 * B)(*([T] * B))
        flags = _lib.FWD_KEEP
        need = self.lib.prego_miniroad_workspace_bytes(self.h, B, lens_arr, B * T, flags)
        if getattr(self, "_ws_train", None) is None or self._ws_train.numel() < need:
            self._ws_train = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty((B, T, ncls), dtype=torch.float32, device=self.device)
        esz = 4
        rgb_p = None if rgb is None else ptr_array([rgb.data_ptr() + b * T * d_rgb * esz for b in range(B)])
        flow_p = None if flow is None else ptr_array([flow.data_ptr() + b * T * d_flow * esz for b in range(B)])
        out_p = ptr_array([out.data_ptr() + b * T * ncls * esz for b in range(B)])
        with torch.cuda.device(self.device):
            check(self.lib.prego_miniroad_forward(self.h, B, lens_arr, rgb_p, flow_p, out_p, None, None, None, flags,
                                                  C.c_void_p(self._ws_train.data_ptr()), self._ws_train.numel(),
                                                  C.c_void_p(_stream_ptr(self.device))))
        self._train_ctx = (B, T, lens_arr, rgb, flow)
        return out

    _CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int)

    def set_bucket_hook(self, fn):
        """fn(engine, i) is called from INSIDE backward(), on the calling thread, right after the launches that make sub-bucket i of the flat
        gradient final have been enqueued and its event recorded (i = 0 head, 1 GRU; bucket 2 is final when backward returns): the
        data-parallel trainer enqueues that sub-bucket's all-reduce there.  None removes the hook."""
        self._bucket_hook = fn
        if fn is None:
            self._cb = None
            check(self.lib.prego_miniroad_backward_callback(self.h, None, None))
            return

        def tramp(_user, i):
            try:
                fn(self, int(i))
            except BaseException as e:          # never unwind through the C frames: surface it after the call
                self._cb_error = e
        self._cb = self._CB(tramp)
        check(self.lib.prego_miniroad_backward_callback(self.h, self._cb, None))

    def backward(self, dlogits: torch.Tensor) -> dict:
        """gradients of the ten parameters (reference state_dict names) for the last forward_train()."""
        d_rgb, d_flow, emb, hid, ncls = self.dims
        B, T, lens_arr, _, _ = self._train_ctx
        dlogits = dlogits.to(torch.float32).contiguous()
        need = self.lib.prego_miniroad_backward_workspace_bytes(self.h, B, lens_arr)
        if getattr(self, "_ws_bwd", None) is None or self._ws_bwd.numel() < need:
            self._ws_bwd = torch.empty(need, dtype=torch.uint8, device=self.device)
        shapes = {"layer1.0.weight": (emb, d_rgb + d_flow), "layer1.0.bias": (emb,), "layer1.1.weight": (emb,),
                  "layer1.1.bias": (emb,), "gru.weight_ih_l0": (3 * hid, emb), "gru.weight_hh_l0": (3 * hid, hid),
                  "gru.bias_ih_l0": (3 * hid,), "gru.bias_hh_l0": (3 * hid,), "f_classification.0.weight": (ncls, hid),
                  "f_classification.0.bias": (ncls,)}
        # one flat fp32 bucket (17.9 M elements = 71.7 MB), the ten gradients are views into it: data-parallel training
        # all-reduces the bucket in place (no gather / scatter copies around the collective)
        order = param_order(self.num_layers)
        if self.num_layers == 2:
            shapes.update({"gru.weight_ih_l1": (3 * hid, hid), "gru.weight_hh_l1": (3 * hid, hid), "gru.bias_ih_l1": (3 * hid,), "gru.bias_hh_l1": (3 * hid,)})
        sizes = [int(torch.Size(shapes[k]).numel()) for k in order]
        offs, total = [], 0
        guard_off = None
        for k, n in zip(order, sizes):
            if k == "gru.weight_ih_l0":             # end of the layer1 / LayerNorm bucket (the LAST one the backward finishes): one slot of
                guard_off, total = total, total + 64    # 64 floats whose first element carries this rank's timeout flag through the all-reduce
            offs.append(total)
            total += (n + 63) // 64 * 64            # 256-byte aligned views
        self._grad_flat = torch.empty(total, dtype=torch.float32, device=self.device)   # every gradient tensor is overwritten by backward
        grads = {k: self._grad_flat[o:o + n].view(shapes[k]) for k, o, n in zip(order, offs, sizes)}
        # buckets of the flat tensor in the order the backward finishes them (csrc/miniroad.cpp: head, GRU, then LayerNorm / layer1):
        # the first two are announced by events recorded inside the backward, the last one is final when backward returns
        o_ih, o_fc = offs[order.index("gru.weight_ih_l0")], offs[order.index("f_classification.0.weight")]
        self._grad_bounds = [(o_fc, total), (o_ih, o_fc), (0, o_ih)]       # bucket 2 ends with the guard slot
        self._guard_off = guard_off
        self._grad_offsets = {k: (o, n) for k, o, n in zip(order, offs, sizes)}     # where each tensor lives in the flat bucket
        self._grad_events = None
        self._early_done = set()
        self._cb_error = None
        dp = torch.distributed.is_available() and torch.distributed.is_initialized()
        if dp and (torch.distributed.get_world_size() > 1 or os.environ.get("PREGO_DP_FORCE_COLLECTIVE") == "1"):
            if getattr(self, "_bwd_events", None) is None:
                evs = [torch.cuda.Event(), torch.cuda.Event()]
                with torch.cuda.device(self.device):
                    for e in evs:
                        e.record()                  # creates the hipEvent_t behind the torch object
                    check(self.lib.prego_miniroad_backward_events(self.h, C.c_void_p(evs[0].cuda_event), C.c_void_p(evs[1].cuda_event)))
                self._bwd_events = evs
            self._grad_events = [self._bwd_events[0], self._bwd_events[1], None]
        dl_p = ptr_array([dlogits.data_ptr() + b * T * ncls * 4 for b in range(B)])
        with torch.cuda.device(self.device):
            if self.num_layers == 2:        # the second layer's gradients: handed over beside the call (ABI 7)
                check(self.lib.prego_miniroad_set_gru_layer_grads(self.h, 1, *[C.c_void_p(grads[k].data_ptr()) for k in _L1_KEYS]))
            check(self.lib.prego_miniroad_backward(
                self.h, B, lens_arr, dl_p, *[C.c_void_p(grads[k].data_ptr()) for k in _PARAM_ORDER],
                C.c_void_p(self._ws_train.data_ptr()), self._ws_train.numel(),
                C.c_void_p(self._ws_bwd.data_ptr()), self._ws_bwd.numel(), C.c_void_p(_stream_ptr(self.device))))
            if self._grad_events is not None:
                # data parallel: this rank's timeout flag rides in the last sub-bucket (summed over the ranks), and the fused AdamW step
                # of EVERY rank is a no-op while the sum is non-zero (prego_miniroad_guard_publish / _set_peer_guard; advisor, round 5: a
                # rank that gave up used to feed garbage into the other ranks' steps for up to CHECK_EVERY steps)
                gp = self._grad_flat.data_ptr() + 4 * guard_off
                check(self.lib.prego_miniroad_guard_publish(self.h, C.c_void_p(gp), C.c_void_p(_stream_ptr(self.device))))
                check(self.lib.prego_miniroad_set_peer_guard(self.h, C.c_void_p(gp)))
            else:
                check(self.lib.prego_miniroad_set_peer_guard(self.h, None))
        if self._cb_error is not None:
            e, self._cb_error = self._cb_error, None
            raise e
        if getattr(self, "_bucket_hook", None) is not None and self._grad_events is None:
            self._early_done = set()            # hook armed without a process group: nothing was reduced
        return grads

    def check(self):
        """synchronise and surface a recurrence timeout"""
        with torch.cuda.device(self.device):
            check(self.lib.prego_miniroad_check(self.h, C.c_void_p(_stream_ptr(self.device))))

    def pass_info(self) -> dict:
        """what the last forward() ran: mode 0 = chunked pass, R > 0 = split pass with the recurrence on R XCDs; steps, slots"""
        m, st, sl = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.lib.prego_miniroad_pass_info(self.h, C.byref(m), C.byref(st), C.byref(sl)))
        return dict(mode=m.value, steps=st.value, slots=sl.value)

    # -- kernel timing (bench.py roofline leg) -----------------------------------------------
    def timing_enable(self, on: bool = True):
        check(self.lib.prego_miniroad_timing_enable(self.h, 1 if on else 0))

    def timing_read(self) -> dict:
        d = [C.c_double() for _ in range(5)]
        n = [C.c_int64() for _ in range(3)]
        check(self.lib.prego_miniroad_timing_read(self.h, C.byref(d[0]), C.byref(n[0]), C.byref(d[1]), C.byref(d[2]),
                                                  C.byref(n[1]), C.byref(d[3]), C.byref(n[2]), C.byref(d[4])))
        return dict(gemm_ms=d[0].value, gemm_launches=n[0].value, gemm_flop=d[1].value, gru_ms=d[2].value,
                    gru_launches=n[1].value, pack_ms=d[3].value, pack_launches=n[2].value, pack_bytes=d[4].value)


def oad_loss(logits: torch.Tensor, target: torch.Tensor, want_grad: bool = True, grad_scale: float = 1.0, reduction: str = "mean"):
    """OadLoss on the device (criterions/loss.py:15-34) for uniform [B,T,C] tensors; reduction 'mean' or 'sum' (loss.py:30-33).
    Returns (loss scalar tensor, dlogits [B,T,C] or None)."""
    if reduction not in ("mean", "sum"):
        raise PregoError(f"oad_loss: reduction {reduction!r} (the reference knows 'mean' and 'sum', loss.py:30-33)")
    lib = _lib.load()
    if target.shape[1] != logits.shape[1]:
        # `Transformer` emits ONE logit row per window ([B,1,C], ViT.py:140) while the target keeps every frame ([B,T,C]):
        # loss.py:18-19 takes logits[:, -1] and target[:, -1], so only the last rows meet
        if logits.shape[1] != 1:
            raise PregoError(f"oad_loss: logits {tuple(logits.shape)} vs target {tuple(target.shape)}")
        target = target[:, -1:, :]
    B, T, Cn = logits.shape
    logits = logits.to(torch.float32).contiguous()
    target = target.to(torch.float32).contiguous()
    dev = logits.device
    if dev.type != "cuda":
        raise PregoError("oad_loss runs on the GPU only")
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dl = torch.empty_like(logits) if want_grad else None
    lens_arr = (C.c_int32 * B)(*([T] * B))
    lp = ptr_array([logits.data_ptr() + b * T * Cn * 4 for b in range(B)])
    tp = ptr_array([target.data_ptr() + b * T * Cn * 4 for b in range(B)])
    dp = None if dl is None else ptr_array([dl.data_ptr() + b * T * Cn * 4 for b in range(B)])
    with torch.cuda.device(dev):
        check(lib.prego_oad_loss_reduce(B, lens_arr, lp, tp, Cn, 1 if reduction == "sum" else 0, C.c_void_p(loss.data_ptr()), dp, float(grad_scale),
                                        C.c_void_p(_stream_ptr(dev))))
    return loss, dl
