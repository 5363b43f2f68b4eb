"""MiniROAD training step 16 x 128 (fwd + OadLoss + BPTT + fused AdamW), GPU-bound timing as bench.py's secondary.train_step_ms
(no host sync inside the loop).  usage: python scripts/probes/train_step_time.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.optim import FusedAdamW
from prego_amd.registry import build_criterion, build_model
import prego_amd.loss, prego_amd.model  # noqa: F401

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = "cuda:0"
cfg = assembly101_cfg(compute_dtype="bf16")
m = build_model(cfg, dev)
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
crit = build_criterion(cfg, dev)
opt = FusedAdamW([{"params": list(m.parameters()), "initial_lr": 1e-4}], lr=1e-4, weight_decay=0.05, model=m)
B, T = 16, 128
rgb = torch.randn(B, T, 2048, device=dev).clamp_(min=0)
flow = torch.randn(B, T, 2048, device=dev).clamp_(min=0)
tgt = torch.zeros(B, T, 86, device=dev); tgt[:, :, 3] = 1


def step():
    m.train()
    loss = crit(m(rgb, flow), tgt)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(10):
    step()
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
m.engine(train=True).check()
print(f"train step 16 x 128: {best:.3f} ms")
