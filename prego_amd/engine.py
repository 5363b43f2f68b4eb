"""Thin host wrapper around the C ABI: torch supplies device memory and streams,
the arithmetic runs in libprego_amd.so."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import PregoError, check, ptr_array

_PARAM_ORDER = [
    "layer1.0.weight", "layer1.0.bias", "layer1.1.weight", "layer1.1.bias",
    "gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0",
    "f_classification.0.weight", "f_classification.0.bias",
]


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class MiniRoadEngine:
    """One handle per (device, model).  compute_dtype: 'bf16' (product path) or 'fp32' (parity mode)."""

    def __init__(self, d_rgb: int, d_flow: int, emb: int, hid: int, n_classes: int, device,
                 compute_dtype: str = "bf16"):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise PregoError("prego_amd runs on an MI355X only (device must be cuda:N); there is no CPU path")
        self.dims = (d_rgb, d_flow, emb, hid, n_classes)
        self.compute_dtype = compute_dtype
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.prego_miniroad_create(C.byref(h), d_rgb, d_flow, emb, hid, n_classes,
                                                 _lib.PREGO_BF16 if compute_dtype == "bf16" else _lib.PREGO_F32))
        self.h = h
        self.max_clips = self.lib.prego_miniroad_max_clips(self.h)
        self._ws: Optional[torch.Tensor] = None
        self.rows_per_chunk = 65536
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
        self._keep = ts   # alive until the stream has consumed them

    # -- forward ---------------------------------------------------------------------------
    def _workspace(self, n_clips, lens_arr, flags):
        need = self.lib.prego_miniroad_workspace_bytes(self.h, n_clips, lens_arr, self.rows_per_chunk, flags)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def forward_ragged(self, rgb: Sequence[torch.Tensor], flow: Optional[Sequence[Optional[torch.Tensor]]],
                       softmax: bool = True, want_out: bool = True, want_argmax: bool = False,
                       h0: Optional[torch.Tensor] = None, want_h_last: bool = False):
        """rgb[i]: fp32 cuda [T_i, d_rgb] contiguous; flow[i] likewise or None (= zeros).
        Returns (outs list of [T_i, C] or None, argmax list of int32 [T_i] or None, h_last or None)."""
        d_rgb, d_flow, emb, hid, ncls = self.dims
        n = len(rgb)
        outs = [None] * n
        args = [None] * n
        h_last = torch.empty((n, hid), dtype=torch.float32, device=self.device) if want_h_last else None
        for s in range(0, n, self.max_clips):
            e = min(n, s + self.max_clips)
            self._forward_pass(rgb[s:e], None if flow is None else flow[s:e], softmax, want_out, want_argmax,
                               None if h0 is None else h0[s:e], None if h_last is None else h_last[s:e],
                               outs, args, s)
        return (outs if want_out else None), (args if want_argmax else None), h_last

    def _forward_pass(self, rgb, flow, softmax, want_out, want_argmax, h0, h_last, outs, args, base):
        d_rgb, d_flow, emb, hid, ncls = self.dims
        n = len(rgb)
        lens = []
        for i, r in enumerate(rgb):
            if r.dtype != torch.float32 or not r.is_cuda or not r.is_contiguous() or r.dim() != 2 or r.shape[1] != d_rgb:
                raise PregoError(f"rgb[{i}] must be a contiguous fp32 cuda tensor [T, {d_rgb}], got {tuple(r.shape)} {r.dtype}")
            lens.append(r.shape[0])
        if flow is not None:
            for i, f in enumerate(flow):
                if f is None:
                    continue
                if f.dtype != torch.float32 or not f.is_cuda or not f.is_contiguous() or tuple(f.shape) != (lens[i], d_flow):
                    raise PregoError(f"flow[{i}] must be a contiguous fp32 cuda tensor [{lens[i]}, {d_flow}]")
        lens_arr = (C.c_int32 * n)(*lens)
        flags = _lib.FWD_SOFTMAX if softmax else 0
        ws = self._workspace(n, lens_arr, flags)
        rgb_p = ptr_array([r.data_ptr() for r in rgb])
        flow_p = None if flow is None else ptr_array([None if f is None else f.data_ptr() for f in flow])
        out_p = arg_p = None
        if want_out:
            for i in range(n):
                outs[base + i] = torch.empty((lens[i], ncls), dtype=torch.float32, device=self.device)
            out_p = ptr_array([outs[base + i].data_ptr() for i in range(n)])
        if want_argmax:
            for i in range(n):
                args[base + i] = torch.empty((lens[i],), dtype=torch.int32, device=self.device)
            arg_p = ptr_array([args[base + i].data_ptr() for i in range(n)])
        with torch.cuda.device(self.device):
            check(self.lib.prego_miniroad_forward(
                self.h, n, lens_arr, rgb_p, flow_p, out_p, arg_p,
                C.c_void_p(h0.data_ptr()) if h0 is not None else None,
                C.c_void_p(h_last.data_ptr()) if h_last is not None else None,
                flags, C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(_stream_ptr(self.device))))

    def check(self):
        """synchronise and surface a recurrence timeout"""
        with torch.cuda.device(self.device):
            check(self.lib.prego_miniroad_check(self.h, C.c_void_p(_stream_ptr(self.device))))

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
