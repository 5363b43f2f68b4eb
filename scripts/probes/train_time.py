"""MiniROAD training step 16 x 128 (fwd + OadLoss + BPTT + fused AdamW): ms per step, best of batches"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_criterion, build_model
import prego_amd.loss, prego_amd.model  # noqa
from prego_amd.optim import FusedAdamW
dev = "cuda:0"
cfg = assembly101_cfg(compute_dtype="bf16")
m = build_model(cfg, dev); m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
crit = build_criterion(cfg, dev)
opt = FusedAdamW([{"params": list(m.parameters())}], lr=1e-4, weight_decay=0.05, model=m)
rgb = torch.randn(16, 128, 2048, device=dev).clamp_(min=0); flow = torch.randn(16, 128, 2048, device=dev).clamp_(min=0)
tgt = torch.zeros(16, 128, 86, device=dev); tgt[:, :, 3] = 1
def step():
    m.train(); loss = crit(m(rgb, flow), tgt); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(5): step()
best = 1e9
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
m.engine(train=True).check()
print(f"train step: {best*1e3:.4f} ms")
