"""ViTEnc forward, 256 windows of 128 frames, 10 iterations: for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.transformer  # noqa: F401
cfg = assembly101_cfg(model="Transformer", window_size=128, num_heads=8, patch_dim=1, attn_dropout_rate=0.0, dropout=0.0)
m = build_model(cfg, "cuda:0")
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.vit_state_dict(cfg, 20).items()})
m.eval()
rgb = torch.randn(256, 128, 2048, device="cuda").clamp_(min=0); flow = torch.randn(256, 128, 2048, device="cuda").clamp_(min=0)
with torch.no_grad():
    for _ in range(10):
        out = m(rgb, flow)
torch.cuda.synchronize()
