"""CPU baseline port: the same MROAD forward restated on PyTorch's stock CPU ops.

TEST/BENCH INFRASTRUCTURE ONLY (never imported by prego_amd/).  The reference's
arithmetic *is* PyTorch's CPU kernels (nn.Linear / nn.LayerNorm / nn.GRU, called at
model/rnn/rnn.py:38-47,58-69), so the fair "reference on the host cores" timing is a
restatement that calls the same ATen ops in the same order; this file is that
restatement ("kind": "port" in bench.py's cpu_baseline).  It is pinned against the
golden vectors in tests/test_oracle_golden.py::test_torch_port_matches_golden.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.nn.functional as F


class TorchPort:
    def __init__(self, sd: dict, hidden: int):
        t = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in sd.items()}
        self.p = t
        e = t["gru.weight_ih_l0"].shape[1]
        self.gru = torch.nn.GRU(e, hidden, 1, batch_first=True)             # rnn.py:38
        with torch.no_grad():
            self.gru.weight_ih_l0.copy_(t["gru.weight_ih_l0"]); self.gru.weight_hh_l0.copy_(t["gru.weight_hh_l0"])
            self.gru.bias_ih_l0.copy_(t["gru.bias_ih_l0"]); self.gru.bias_hh_l0.copy_(t["gru.bias_hh_l0"])
        self.gru.eval()
        self.hidden = hidden

    @torch.no_grad()
    def forward(self, rgb: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
        """eval-mode MROAD.forward (rnn.py:51-71): returns softmax probabilities [B,T,C]."""
        p = self.p
        x = torch.cat((rgb, flow), 2)                                             # rnn.py:53
        x = F.linear(x, p["layer1.0.weight"], p["layer1.0.bias"])                 # rnn.py:40
        x = F.relu(F.layer_norm(x, (x.shape[-1],), p["layer1.1.weight"], p["layer1.1.bias"], 1e-5))  # :41-42
        h0 = torch.zeros(1, x.shape[0], self.hidden)                              # rnn.py:49,60
        ht, _ = self.gru(x, h0)                                                   # rnn.py:61
        logits = F.linear(F.relu(ht), p["f_classification.0.weight"], p["f_classification.0.bias"])  # :62-64
        return F.softmax(logits, dim=-1)                                          # rnn.py:69


def time_reference_faithful(port: TorchPort, clips, budget_s: float = 20.0):
    """Reference-faithful batching (trainer/eval.py:36-45, dataset.py:120-123): one whole video per
    forward call, batch 1.  Runs clips until `budget_s` of CPU time is spent.  Returns (frames, seconds)."""
    frames, t_total = 0, 0.0
    for rgb, flow in clips:
        t0 = time.perf_counter()
        port.forward(rgb[None], flow[None])
        t_total += time.perf_counter() - t0
        frames += rgb.shape[0]
        if t_total > budget_s:
            break
    return frames, t_total
