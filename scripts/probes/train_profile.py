"""one training configuration (B=16, T=128) run 20 steps: for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model, build_criterion
import prego_amd.model, prego_amd.loss  # noqa: F401
cfg = assembly101_cfg(compute_dtype="bf16")
sd = W.miniroad_state_dict(cfg, 20)
m = build_model(cfg, "cuda:0"); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
crit = build_criterion(cfg)
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.05)
rgb = torch.randn(16, 128, 2048, device="cuda").clamp_(min=0); flow = torch.randn(16, 128, 2048, device="cuda").clamp_(min=0)
tgt = torch.zeros(16, 128, 86, device="cuda"); tgt[:, :, 3] = 1
m.train()
for _ in range(20):
    loss = crit(m(rgb, flow), tgt); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
torch.cuda.synchronize()
