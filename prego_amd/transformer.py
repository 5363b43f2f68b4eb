"""`Transformer` (ViTEnc) behind the reference's plug-in API (model/transformer_models/ViT.py:25-143) and the causal
`AttentionLayer(FullAttention)` op of attn.py as a function.  Same constructor keys (`patch_dim`, `num_heads`,
`attn_dropout_rate` on top of the yaml), same `forward(rgb, flow) -> {'logits': [B,1,C]}` (raw logits in both modes),
same state_dict keys/shapes, so reference checkpoints load.  bf16 MFMA operands, fp32 residual
stream / LayerNorm / softmax.  Training (trainer/train.py:20-24) runs through `prego_vit_forward_train` / `prego_vit_backward`
behind a torch.autograd.Function, with cfg['dropout'] and cfg['attn_dropout_rate'] as stateless hash masks at the reference's six
nn.Dropout sites (positional, attention probabilities, proj_drop, after the attention block, twice inside the FFN)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._lib import PregoError, check, ptr_array
from .config import FEATURE_SIZES
from .engine import _stream_ptr
from .registry import META_ARCHITECTURES


def _tensor_order(num_layers):
    names = ["linear_encoding.weight", "linear_encoding.bias", "cls_token", "position_encoding.pe.weight"]
    for l in range(num_layers):
        a, f = 2 * l, 2 * l + 1
        names += [f"encoder.net.{a}.fn.norm.weight", f"encoder.net.{a}.fn.norm.bias", f"encoder.net.{a}.fn.fn.qkv.weight",
                  f"encoder.net.{a}.fn.fn.proj.weight", f"encoder.net.{a}.fn.fn.proj.bias",
                  f"encoder.net.{f}.fn.norm.weight", f"encoder.net.{f}.fn.norm.bias",
                  f"encoder.net.{f}.fn.fn.net.0.weight", f"encoder.net.{f}.fn.fn.net.0.bias",
                  f"encoder.net.{f}.fn.fn.net.3.weight", f"encoder.net.{f}.fn.fn.net.3.bias"]
    return names + ["pre_head_ln.weight", "pre_head_ln.bias", "mlp_head.weight", "mlp_head.bias"]


class _Norm(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = inner


class _Wrap(nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.fn = inner


class _Attn(nn.Module):                       # Attention.py:7-19 (parameter container)
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)


class _FF(nn.Module):                         # Transformer.py:35-44 (parameter container; net.0 and net.3 are the Linears)
    def __init__(self, dim, hidden):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Identity(), nn.Linear(hidden, dim), nn.Identity())


class _PE(nn.Module):                         # PositionalEncoding.py:25-34
    def __init__(self, n, dim):
        super().__init__()
        self.pe = nn.Embedding(n, dim)
        self.register_buffer("position_ids", torch.arange(n).expand((1, -1)))


@META_ARCHITECTURES.register("Transformer")
class ViTEnc(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.img_dim = cfg["window_size"]
        self.out_dim = cfg["num_classes"]
        self.embedding_dim = cfg["embedding_dim"]
        self.patch_dim = cfg["patch_dim"]
        self.num_heads = cfg["num_heads"]
        self.num_layers = cfg["num_layers"]
        self.hidden_dim = cfg["hidden_dim"]
        if self.patch_dim != 1:
            raise PregoError("patch_dim != 1 is not supported (the conv-patch branch is dead code in the reference, ViT.py:92-112)")
        self.use_flow = not cfg["no_flow"]
        self.use_rgb = not cfg["no_rgb"]
        self.d_rgb = FEATURE_SIZES[cfg["rgb_type"]] if self.use_rgb else 0
        self.d_flow = FEATURE_SIZES[cfg["flow_type"]] if self.use_flow else 0
        self.num_channels = self.d_rgb + self.d_flow
        self.seq_length = self.img_dim // self.patch_dim + 1
        E = self.embedding_dim
        self.cls_token = nn.Parameter(torch.zeros(1, 1, E))                                   # ViT.py:55
        self.linear_encoding = nn.Linear(self.num_channels, E)                                  # ViT.py:58
        self.position_encoding = _PE(self.seq_length, E)                                        # ViT.py:60-62
        layers = []
        for _ in range(self.num_layers):                                                        # Transformer.py:60-77
            layers += [_Wrap(_Norm(E, _Attn(E))), _Wrap(_Norm(E, _FF(E, self.hidden_dim)))]
        self.encoder = nn.Module()
        self.encoder.net = nn.Sequential(*layers)
        self.pre_head_ln = nn.LayerNorm(E)                                                      # ViT.py:79
        self.mlp_head = nn.Linear(E, self.out_dim)                                              # ViT.py:90
        self.dropout_rate = float(cfg.get("dropout", 0.0))           # ViT.py:38
        self.attn_dropout_rate = float(cfg.get("attn_dropout_rate", 0.0))   # ViT.py:47
        self.causal = bool(cfg.get("causal_attention", False))     # extension, see DESIGN.md
        # MFMA operand type of the INFERENCE entry points (forward in eval mode, forward_frames): 'fp16' (default: same rate as bf16,
        # 8x less operand rounding), 'bf16', or 'fp32' (parity mode: fp32 operands on the exact-fp32 MFMA GEMM and an fp32
        # attention kernel; window forward only, no per-frame runner).  Training always runs on the bf16 handle.
        self.compute_dtype = cfg.get("compute_dtype", "fp16")
        if self.compute_dtype not in ("fp16", "bf16", "fp32"):
            raise PregoError(f"ViTEnc compute_dtype {self.compute_dtype!r}: 'fp16', 'bf16' or 'fp32'")
        self._h = None               # bf16 handle: training, and inference when compute_dtype == 'bf16'
        self._ver = None
        self._h16 = None             # fp16 / fp32 handle (inference only), own weight copies
        self._ver16 = None
        self._ws = None

    def _handle(self):
        dev = self.mlp_head.weight.device
        if dev.type != "cuda":
            raise PregoError("prego_amd runs on an MI355X only (device must be cuda:N); there is no CPU path")
        lib = _lib.load()
        if self._h is None:
            h = C.c_void_p()
            with torch.cuda.device(dev):
                check(lib.prego_vit_create(C.byref(h), self.d_rgb, self.d_flow, self.embedding_dim, self.hidden_dim,
                                           self.num_heads, self.num_layers, self.img_dim, self.out_dim))
            self._h = h
        sd = dict(self.named_parameters())
        ver = tuple((p.data_ptr(), p._version) for p in sd.values())
        if ver != self._ver:
            ts = [sd[k].detach().float().contiguous() for k in _tensor_order(self.num_layers)]
            with torch.cuda.device(dev):
                check(lib.prego_vit_set_weights(self._h, ptr_array([t.data_ptr() for t in ts]), len(ts),
                                                C.c_void_p(_stream_ptr(dev))))
            self._keep, self._ver = ts, ver
        return lib, dev

    def _eval_handle(self):
        """(lib, device, handle) of the inference entry points: the fp16- / fp32-operand handle when compute_dtype says so"""
        if self.compute_dtype == "bf16":
            lib, dev = self._handle()
            return lib, dev, self._h
        dev = self.mlp_head.weight.device
        if dev.type != "cuda":
            raise PregoError("prego_amd runs on an MI355X only (device must be cuda:N); there is no CPU path")
        lib = _lib.load()
        if self._h16 is None:
            h = C.c_void_p()
            with torch.cuda.device(dev):
                check(lib.prego_vit_create(C.byref(h), self.d_rgb, self.d_flow, self.embedding_dim, self.hidden_dim,
                                           self.num_heads, self.num_layers, self.img_dim, self.out_dim))
                check(lib.prego_vit_set_compute_dtype(h, _lib.PREGO_F32 if self.compute_dtype == "fp32" else _lib.PREGO_F16))
            self._h16 = h
        sd = dict(self.named_parameters())
        ver = tuple((p.data_ptr(), p._version) for p in sd.values())
        if ver != self._ver16:
            ts = [sd[k].detach().float().contiguous() for k in _tensor_order(self.num_layers)]
            with torch.cuda.device(dev):
                check(lib.prego_vit_set_weights(self._h16, ptr_array([t.data_ptr() for t in ts]), len(ts), C.c_void_p(_stream_ptr(dev))))
            self._keep16, self._ver16 = ts, ver
        return lib, dev, self._h16

    def _inputs(self, sequence_input_rgb, sequence_input_flow):
        rgb = sequence_input_rgb.float().contiguous() if self.use_rgb else None
        flow = sequence_input_flow.float().contiguous() if self.use_flow else None
        B, T = (rgb if rgb is not None else flow).shape[:2]
        if T != self.img_dim:
            raise PregoError(f"ViTEnc needs T == window_size ({self.img_dim}), got {T} (learned positional table, PositionalEncoding.py:25-41)")
        return rgb, flow, B

    def forward(self, sequence_input_rgb, sequence_input_flow):
        if self.training and torch.is_grad_enabled():
            names = _tensor_order(self.num_layers)
            sd = dict(self.named_parameters())
            return {"logits": _ViTTrainFn.apply(self, sequence_input_rgb, sequence_input_flow, *[sd[k] for k in names]).unsqueeze(1)}
        lib, dev, hnd = self._eval_handle()
        rgb, flow, B = self._inputs(sequence_input_rgb, sequence_input_flow)
        need = lib.prego_vit_workspace_bytes(hnd, B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty((B, self.out_dim), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(lib.prego_vit_forward(hnd, B, None if rgb is None else C.c_void_p(rgb.data_ptr()), None if flow is None else C.c_void_p(flow.data_ptr()),
                                        C.c_void_p(out.data_ptr()), (1 if self.causal else 0) | (2 if getattr(self, "debug_all_rows", False) else 0),
                                        C.c_void_p(self._ws.data_ptr()),
                                        self._ws.numel(), C.c_void_p(_stream_ptr(dev))))
        return {"logits": out.unsqueeze(1)}

    # -- per-frame inference over whole videos (the eval loop, trainer/eval.py:36-56) ------------------------------------------
    # windows that go through the encoder together (one ViTEnc forward of that batch size).  8 192-frame video, same device:
    # 128: 221 k frames/s, 256: 275 k, 512: 329 k, 1 024: 347 k, 2 048: 366 k (scripts/probes/vit_wb_sweep.py); 1 024 windows = 2.9 GB
    # of workspace
    windows_per_batch = 1024
    max_clips = 64                   # Evaluate batches this many videos per call (each runs on its own: no cross-video batching)

    @torch.no_grad()
    def forward_frames(self, rgb, flow=None, want_argmax=True):
        """One whole video: rgb [T, d_rgb] / flow [T, d_flow] fp32 cuda (flow None = zeros) -> (logits [T, C], argmax int32 [T]).
        logits[t] = the reference forward on the `window_size` frames ending at t, zero feature rows in front of the video - the
        windows the training loader cuts (dataset.py:53-55,96-103) at stride 1; linear_encoding runs once per frame
        (prego_vit_forward_frames).  ViTEnc's output is raw logits in both modes (ViT.py:138-141)."""
        lib, dev, hnd = self._eval_handle()
        rgb = rgb.float().contiguous() if self.use_rgb else None
        flow = flow.float().contiguous() if (self.use_flow and flow is not None) else None
        src = rgb if rgb is not None else flow
        if src is None:
            raise PregoError("forward_frames: a --no_rgb model needs the flow tensor")
        T = int(src.shape[0])
        wb = min(self.windows_per_batch, T)
        need = lib.prego_vit_frames_workspace_bytes(hnd, T, wb)
        if getattr(self, "_ws_frames", None) is None or self._ws_frames.numel() < need:
            self._ws_frames = None
            self._ws_frames = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty((T, self.out_dim), dtype=torch.float32, device=dev)
        arg = torch.empty((T,), dtype=torch.int32, device=dev) if want_argmax else None
        with torch.cuda.device(dev):
            check(lib.prego_vit_forward_frames(hnd, T, None if rgb is None else C.c_void_p(rgb.data_ptr()),
                                               None if flow is None else C.c_void_p(flow.data_ptr()), C.c_void_p(out.data_ptr()),
                                               None if arg is None else C.c_void_p(arg.data_ptr()), wb, 1 if self.causal else 0,
                                               C.c_void_p(self._ws_frames.data_ptr()), self._ws_frames.numel(), C.c_void_p(_stream_ptr(dev))))
        return out, arg

    @torch.no_grad()
    def forward_clips(self, rgb_list, flow_list=None, want_probs=True, want_argmax=True):
        """the batched-eval interface `Evaluate` drives (same shape as MROAD.forward_clips): per video the per-frame score matrix
        [T, C] (raw logits here) and int32 argmax [T]"""
        outs, args = [], []
        n = len(rgb_list) if rgb_list is not None else len(flow_list)
        for i in range(n):
            o, a = self.forward_frames(None if rgb_list is None else rgb_list[i], None if flow_list is None else flow_list[i], want_argmax)
            outs.append(o)
            args.append(a)
        return (outs if want_probs else None), (args if want_argmax else None), None

    def check(self):
        """ViTEnc kernels have no bounded spins: nothing to poll (MROAD.check surfaces a recurrence timeout)"""
        torch.cuda.synchronize(self.mlp_head.weight.device)

    def __del__(self):
        try:
            for h in (self._h, getattr(self, "_h16", None)):
                if h is not None:
                    _lib.load().prego_vit_destroy(h)
        except Exception:
            pass


class _ViTTrainFn(torch.autograd.Function):
    """trainer/train.py:20-24 for model: 'Transformer': autograd only carries the tensors between the two C-ABI calls"""

    @staticmethod
    def forward(ctx, model, rgb_in, flow_in, *params):
        lib, dev = model._handle()
        rgb, flow, B = model._inputs(rgb_in, flow_in)
        need = lib.prego_vit_train_workspace_bytes(model._h, B)
        if getattr(model, "_ws_train", None) is None or model._ws_train.numel() < need:
            model._ws_train = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty((B, model.out_dim), dtype=torch.float32, device=dev)
        any_drop = model.dropout_rate > 0 or model.attn_dropout_rate > 0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if any_drop else 0      # torch's RNG drives the mask seed
        seed = getattr(model, "debug_dropout_seed", seed)
        with torch.cuda.device(dev):
            check(lib.prego_vit_set_dropout(model._h, float(model.dropout_rate), float(model.attn_dropout_rate), seed & 0xFFFFFFFFFFFFFFFF))
            check(lib.prego_vit_forward_train(model._h, B, None if rgb is None else C.c_void_p(rgb.data_ptr()), None if flow is None else C.c_void_p(flow.data_ptr()),
                                              C.c_void_p(out.data_ptr()), 1 if model.causal else 0,
                                              C.c_void_p(model._ws_train.data_ptr()), model._ws_train.numel(), C.c_void_p(_stream_ptr(dev))))
        # activations and mask seeds live in the model's one training workspace / handle: generation-checked in backward
        model._train_gen = getattr(model, "_train_gen", 0) + 1
        ctx.model, ctx.B, ctx.keep, ctx.gen = model, B, (rgb, flow), model._train_gen
        ctx.shapes = [tuple(p.shape) for p in params]
        return out

    @staticmethod
    def backward(ctx, dout):
        model, B = ctx.model, ctx.B
        if getattr(model, "_train_gen", 0) != ctx.gen:
            raise PregoError("ViTEnc backward: another training forward ran on this model since the forward of this graph; its kept "
                             "activations and dropout seeds were overwritten (run backward before the next forward)")
        lib = _lib.load()
        dev = dout.device
        dout = dout.float().contiguous()
        grads = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in ctx.shapes]
        with torch.cuda.device(dev):
            check(lib.prego_vit_backward(model._h, B, C.c_void_p(dout.data_ptr()), ptr_array([g.data_ptr() for g in grads]), len(grads),
                                         1 if model.causal else 0, C.c_void_p(model._ws_train.data_ptr()), model._ws_train.numel(),
                                         C.c_void_p(_stream_ptr(dev))))
        return (None, None, None) + tuple(grads)


class _AttnLayerTrainFn(torch.autograd.Function):
    """AttentionLayer.forward under autograd (attn.py:151-170): the two C-ABI calls around torch's graph"""

    @staticmethod
    def forward(ctx, layer, x_in, *params):
        lib, dev = layer.lib, layer.device
        h = layer._train_handle(params)
        # FullAttention's nn.Dropout(attention_dropout) on A (attn.py:39,54): active in training mode; the mask seed comes from torch's RNG
        # (one draw per forward, like nn.Dropout consumes the generator), the backward of THIS forward regenerates the same mask
        p_drop = layer.attention_dropout if layer.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p_drop > 0 else 0
        check(lib.prego_attention_layer_set_dropout(h, float(p_drop), seed))
        B, L, D = x_in.shape
        x = x_in.detach().float().contiguous()
        need = lib.prego_attention_layer_train_workspace_bytes(h, B, L)
        if layer._ws_train is None or layer._ws_train.numel() < need:
            layer._ws_train = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty((B, L, D), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(lib.prego_attention_layer_forward_train(h, B, L, 1 if layer.mask_flag else 0, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()),
                                                          C.c_void_p(layer._ws_train.data_ptr()), layer._ws_train.numel(), C.c_void_p(_stream_ptr(dev))))
        layer._train_gen += 1      # q, k, v, the attention output and the row log-sum-exp stay in the layer's ONE training workspace
        ctx.layer, ctx.dims, ctx.gen, ctx.need_dx = layer, (B, L, D), layer._train_gen, bool(x_in.requires_grad)
        ctx.drop = (float(p_drop), seed)
        ctx.shapes = [tuple(p.shape) for p in params]
        return out

    @staticmethod
    def backward(ctx, dout):
        layer, (B, L, D) = ctx.layer, ctx.dims
        if layer._train_gen != ctx.gen:
            raise PregoError("AttentionLayer backward: another training forward ran on this layer since the forward of this graph; its "
                             "kept activations were overwritten (run backward before the next forward)")
        lib, dev = layer.lib, layer.device
        dout = dout.float().contiguous()
        grads = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in ctx.shapes]
        dx = torch.empty((B, L, D), dtype=torch.float32, device=dev) if ctx.need_dx else None
        with torch.cuda.device(dev):
            check(lib.prego_attention_layer_set_dropout(layer._ht, *ctx.drop))          # the mask of the forward this graph belongs to
            check(lib.prego_attention_layer_backward(layer._ht, B, L, 1 if layer.mask_flag else 0, C.c_void_p(dout.data_ptr()),
                                                     None if dx is None else C.c_void_p(dx.data_ptr()), ptr_array([g.data_ptr() for g in grads]),
                                                     len(grads), C.c_void_p(layer._ws_train.data_ptr()), layer._ws_train.numel(),
                                                     C.c_void_p(_stream_ptr(dev))))
        return (None, dx) + tuple(grads)


class AttentionLayer:
    """AttentionLayer(FullAttention(mask_flag)) of attn.py:139-170 as an object that owns converted weights: the four projection
    matrices are ingested (fp32 -> bf16) once, every call only moves activations.

    Under autograd (grad mode on and x or a projection parameter requiring grad) the call keeps q, k, v, the attention output and
    the row log-sum-exp and backward() returns the gradients of x and of the eight parameters; that path computes with bf16
    operands on its own handle (as ViTEnc's training does), re-ingesting the parameters whenever an optimizer changed them."""

    def __init__(self, wq, bq, wk, bk, wv, bv, wo, bo, n_heads: int, mask_flag: bool = True, compute_dtype: str = "fp16",
                 attention_dropout: float = 0.0):
        self.lib = _lib.load()
        # FullAttention(attention_dropout=...) (attn.py:36-39; the reference's default is 0.1): acts on the attention probabilities of the
        # AUTOGRAD path while the layer is in training mode (train() / eval(), as nn.Module); inference calls never drop
        if not (0.0 <= float(attention_dropout) < 1.0):
            raise PregoError(f"attention_dropout {attention_dropout}: expected 0 <= p < 1")
        self.attention_dropout = float(attention_dropout)
        self.training = True
        if compute_dtype not in ("fp16", "bf16", "fp32"):
            raise PregoError(f"AttentionLayer compute_dtype {compute_dtype!r}: 'fp16', 'bf16' or 'fp32'")
        self.compute_dtype = compute_dtype
        self.device = wq.device
        if self.device.type != "cuda":
            raise PregoError("prego_amd runs on an MI355X only (device must be cuda:N); there is no CPU path")
        self.d_model, self.n_heads, self.mask_flag = int(wq.shape[0]), int(n_heads), bool(mask_flag)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.prego_attention_layer_create(C.byref(h), self.d_model, self.n_heads))
            check(self.lib.prego_attention_layer_set_compute_dtype(h, {"fp16": _lib.PREGO_F16, "bf16": _lib.PREGO_BF16, "fp32": _lib.PREGO_F32}[compute_dtype]))
            ts = [t.detach().float().contiguous() for t in (wq, bq, wk, bk, wv, bv, wo, bo)]
            self.h = h
            check(self.lib.prego_attention_layer_set_weights(self.h, *[C.c_void_p(t.data_ptr()) for t in ts], C.c_void_p(_stream_ptr(self.device))))
        self._keep = ts
        self._ws = None
        self.params = (wq, bq, wk, bk, wv, bv, wo, bo)
        self._ht, self._ht_key, self._ws_train, self._train_gen = None, None, None, 0

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def _train_handle(self, params):
        """the bf16 handle of the autograd path, holding the CURRENT values of the parameters"""
        key = tuple((t.data_ptr(), t._version) for t in params)
        with torch.cuda.device(self.device):
            if self._ht is None:
                h = C.c_void_p()
                check(self.lib.prego_attention_layer_create(C.byref(h), self.d_model, self.n_heads))
                self._ht = h
            if self._ht_key != key:
                ts = [t.detach().float().contiguous() for t in params]
                check(self.lib.prego_attention_layer_set_weights(self._ht, *[C.c_void_p(t.data_ptr()) for t in ts], C.c_void_p(_stream_ptr(self.device))))
                self._keep_train, self._ht_key = ts, key
        return self._ht

    def __call__(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.params)):
            # the autograd path runs on bf16 operands (the training kernels take bf16 handles; an fp16 layer trains on bf16 like MROAD
            # and ViTEnc do).  A layer that was ASKED for the fp32 parity mode must not silently answer with bf16-operand numbers
            # because grad mode happens to be on: say so.  (Evaluate under torch.no_grad(), or detach the inputs.)
            if self.compute_dtype == "fp32":
                raise PregoError("AttentionLayer(compute_dtype='fp32') called with grad enabled on tensors that require grad: the autograd "
                                 "path is bf16-operand only; wrap the call in torch.no_grad() for the fp32 parity mode")
            return _AttnLayerTrainFn.apply(self, x, *self.params)
        B, L, D = x.shape
        x = x.detach().float().contiguous()
        need = self.lib.prego_attention_layer_handle_workspace_bytes(self.h, B, L)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty((B, L, D), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.prego_attention_layer_handle_forward(self.h, B, L, 1 if self.mask_flag else 0, C.c_void_p(x.data_ptr()),
                                                                C.c_void_p(out.data_ptr()), C.c_void_p(self._ws.data_ptr()), self._ws.numel(),
                                                                C.c_void_p(_stream_ptr(self.device))))
        return out

    def __del__(self):
        try:
            for name in ("h", "_ht"):
                if getattr(self, name, None):
                    self.lib.prego_attention_layer_destroy(getattr(self, name))
                    setattr(self, name, None)
        except Exception:
            pass


_LAYER_CACHE = {}


def attention_layer(x, wq, bq, wk, bk, wv, bv, wo, bo, n_heads: int, mask_flag: bool = True, compute_dtype: str = "fp16"):
    """AttentionLayer(FullAttention(mask_flag)) forward (attn.py:139-170): x [B,L,D] fp32 cuda -> [B,L,D].  Functional form: the
    converted weights are cached on (tensor identity, version), so repeated calls with the same parameters ingest them once."""
    ws = (wq, bq, wk, bk, wv, bv, wo, bo)
    key = tuple((t.data_ptr(), t._version) for t in ws) + (int(n_heads), bool(mask_flag), compute_dtype)
    layer = _LAYER_CACHE.get("layer") if _LAYER_CACHE.get("key") == key else None
    if layer is None:
        layer = AttentionLayer(*ws, n_heads=n_heads, mask_flag=mask_flag, compute_dtype=compute_dtype)
        _LAYER_CACHE.clear()
        _LAYER_CACHE.update(key=key, layer=layer)
    return layer(x)
