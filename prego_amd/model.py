"""`MiniROAD` behind the reference's plug-in API (step_recognition/model/rnn/rnn.py:18-71).

Same constructor (`MROAD(cfg)`), same `forward(rgb_input, flow_input) -> {'logits': ...}`
(probabilities in eval mode, raw logits in training mode, rnn.py:66-70), same
state_dict keys/shapes/dtypes (`gru.*`, `layer1.*`, `f_classification.*`), so reference
checkpoints load here and ours load there.  The torch sub-modules are parameter
containers only (constructed in the reference's order, so a given torch seed gives
the reference's initial weights); every FLOP runs in libprego_amd.so.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .config import FEATURE_SIZES
from .engine import MiniRoadEngine
from ._lib import PregoError
from .registry import META_ARCHITECTURES


@META_ARCHITECTURES.register("MiniROAD")
class MROAD(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.use_flow = not cfg["no_flow"]
        self.use_rgb = not cfg["no_rgb"]
        self.d_rgb = FEATURE_SIZES[cfg["rgb_type"]] if self.use_rgb else 0
        self.d_flow = FEATURE_SIZES[cfg["flow_type"]] if self.use_flow else 0
        self.input_dim = self.d_rgb + self.d_flow
        self.hidden_dim = cfg["hidden_dim"]
        self.num_layers = cfg["num_layers"]
        self.out_dim = cfg["num_classes"]
        self.window_size = cfg["window_size"]
        self.embedding_dim = cfg["embedding_dim"]
        if self.num_layers not in (1, 2):
            raise PregoError(f"prego_amd MiniROAD runs nn.GRU with num_layers 1 or 2 (cfg['num_layers'] = {self.num_layers})")
        # parameter containers, reference construction order (rnn.py:38-47)
        self.gru = nn.GRU(self.embedding_dim, self.hidden_dim, self.num_layers, batch_first=True)
        self.layer1 = nn.Sequential(
            nn.Linear(self.input_dim, self.embedding_dim),
            nn.LayerNorm(self.embedding_dim),
            nn.ReLU(),
            nn.Dropout(p=cfg["dropout"]),
        )
        self.f_classification = nn.Sequential(nn.Linear(self.hidden_dim, self.out_dim))
        # build-specific knobs (not reference keys)
        self.compute_dtype = cfg.get("compute_dtype", "fp16")          # 'fp16' | 'bf16' | 'fp32' | 'fp16x2' (split operands: fp32-class results)
        self.assume_zero_flow = bool(cfg.get("assume_zero_flow", False))  # dataset.py:69 zeroes the flow half
        self.grad_compress = cfg.get("grad_compress")                    # None | 'bf16': data-parallel gradient all-reduce on bf16 (half the bytes)
        self._engines = {}            # (device, operand dtype) -> [MiniRoadEngine, parameter versions its copies belong to]

    # -- engine plumbing -------------------------------------------------------------------
    def _engine_dtype(self, train: bool) -> str:
        # fp16 operands are an inference mode (same speed as bf16, 8x less operand rounding); the training kernels (kept
        # activations, BPTT, wgrads, fused AdamW copies) take bf16 / fp32 handles
        if train and self.compute_dtype == "fp16x2":
            return "fp32"           # the split-operand mode is inference only; its training counterpart is the exact-fp32 engine
        return "bf16" if (train and self.compute_dtype == "fp16") else self.compute_dtype

    @property
    def _engine(self):
        """the training-side engine if one exists (trainer: gradient bucket, timeout check), else the eval engine"""
        dev = self.layer1[0].weight.device
        for train in (True, False):
            ent = self._engines.get((dev, self._engine_dtype(train)))
            if ent is not None:
                return ent[0]
        return None

    def engine(self, train: bool = False) -> MiniRoadEngine:
        dev = self.layer1[0].weight.device
        key = (dev, self._engine_dtype(train))
        ent = self._engines.get(key)
        if ent is None:
            ent = [MiniRoadEngine(self.d_rgb, self.d_flow, self.embedding_dim, self.hidden_dim, self.out_dim, dev, key[1],
                                  num_layers=self.num_layers), None]
            self._engines[key] = ent
        vers = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if vers != ent[1]:
            ent[0].set_weights(dict(self.named_parameters()))
            ent[1] = vers
        return ent[0]

    def _mark_ingested(self, train: bool = True):
        """the engine's operand copies were refreshed in place (fused AdamW): record the current parameter versions as theirs"""
        ent = self._engines.get((self.layer1[0].weight.device, self._engine_dtype(train)))
        if ent is not None:
            ent[1] = tuple((p.data_ptr(), p._version) for p in self.parameters())

    def forward(self, rgb_input, flow_input):
        if self.training:
            from .autograd import miniroad_train_forward
            return {"logits": miniroad_train_forward(self, rgb_input, flow_input)}
        eng = self.engine()
        src = rgb_input if self.use_rgb else flow_input
        B, T = src.shape[0], src.shape[1]
        rgb = [rgb_input[b].contiguous() for b in range(B)] if self.use_rgb else None
        if self.use_flow and (not self.assume_zero_flow or not self.use_rgb):
            flow = [flow_input[b].contiguous() for b in range(B)]       # --no_rgb: flow is the model's only input (rnn.py:54-57)
        else:
            flow = None
        outs, _, _ = eng.forward_ragged(rgb, flow, softmax=True)
        return {"logits": torch.stack(outs, 0)}

    @torch.no_grad()
    def forward_clips(self, rgb_list, flow_list=None, want_probs=True, want_argmax=True):
        """Ragged batched inference (the data-parallel hot path): many whole videos per call."""
        eng = self.engine()
        return eng.forward_ragged(rgb_list, flow_list, softmax=True, want_out=want_probs, want_argmax=want_argmax)

    link_fed_eval = True           # Evaluate may feed forward_clips while it runs (engine().plan_starts / set_feed_events)

    @property
    def max_clips(self) -> int:
        """videos `Evaluate` may hand to one forward_clips call"""
        return self.engine().max_clips

    def check(self):
        """surface a recurrence spin timeout of the eval engine (PregoError, PREGO_ETIMEOUT); synchronises the stream"""
        self.engine().check()

    @torch.no_grad()
    def step(self, rgb, flow, h):
        """Online inference (not exposed by the reference, whose eval loop runs whole videos): one new frame per stream.
        rgb [n, d_rgb] / flow [n, d_flow] (None = zeros), h [n, hidden_dim] = the GRU state, updated in place.  Returns
        (probabilities [n, C], argmax int32 [n]) - the eval branch of MROAD.forward (rnn.py:66-70) at T = 1 with h0 = h."""
        return self.engine().step(rgb, flow, h, softmax=True)
