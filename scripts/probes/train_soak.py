"""soak of the training step (persistent forward + persistent BPTT kernels): N steps, loss must stay finite and no timeout"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model, build_criterion
import prego_amd.model, prego_amd.loss  # noqa: F401
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cfg = assembly101_cfg(compute_dtype="bf16")
m = build_model(cfg, "cuda:0"); m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
crit = build_criterion(cfg); opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.05)
g = torch.Generator(device="cuda").manual_seed(0)
m.train(); t0 = time.perf_counter(); first = last = None
for i in range(n):
    B = 16 if i % 3 else 40                                   # alternate one-tile and multi-group batches
    rgb = torch.randn((B, 128, 2048), device="cuda", generator=g).clamp_(min=0); flow = torch.randn((B, 128, 2048), device="cuda", generator=g).clamp_(min=0)
    tgt = torch.zeros(B, 128, 86, device="cuda"); tgt[:, :, i % 86] = 1
    loss = crit(m(rgb, flow), tgt); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    if i % 50 == 0:
        m.engine().check(); v = float(loss.detach()); first = v if first is None else first; last = v
        assert v == v and abs(v) < 1e4, v
m.engine().check(); torch.cuda.synchronize()
print(f"train soak: {n} steps ok in {time.perf_counter()-t0:.1f} s, loss {first:.4f} -> {last:.4f}")
