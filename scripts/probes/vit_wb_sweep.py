import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.transformer  # noqa
cfg = assembly101_cfg(model="Transformer", window_size=128, patch_dim=1, num_heads=8, attn_dropout_rate=0.0, dropout=0.0)
m = build_model(cfg, "cuda:0")
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.vit_state_dict(cfg, 20).items()}); m.eval()
T = 8192
rgb = torch.randn(T, 2048, device="cuda").clamp_(min=0); flow = torch.randn(T, 2048, device="cuda").clamp_(min=0)
for wb in (128, 256, 512, 1024, 2048):
    m.windows_per_batch = wb
    for _ in range(2): m.forward_frames(rgb, flow)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): m.forward_frames(rgb, flow)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"windows_per_batch {wb}: {T / dt / 1e3:.1f} k frames/s ({dt * 1e3:.1f} ms per {T}-frame video)")
