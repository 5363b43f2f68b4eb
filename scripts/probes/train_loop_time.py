"""train_one_epoch over a pinned synthetic loader (16 x 128 windows, rgb + flow): ms per step with the batches arriving over the link,
prefetched on a side stream (default) against the reference's blocking copies"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_criterion, build_model, build_trainer
import prego_amd.loss, prego_amd.model, prego_amd.trainer as TR  # noqa
from prego_amd.optim import FusedAdamW
dev = "cuda:0"
zero = "--zero-flow" in sys.argv
cfg = assembly101_cfg(compute_dtype="bf16", assume_zero_flow=zero)
m = build_model(cfg, dev); m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
crit = build_criterion(cfg, dev)
opt = FusedAdamW([{"params": list(m.parameters())}], lr=1e-4, weight_decay=0.05, model=m)
tr = build_trainer(cfg)
g = torch.Generator().manual_seed(1)
batches = []
for i in range(24):
    rgb = torch.randn(16, 128, 2048, generator=g).clamp_(min=0).pin_memory()
    flow = torch.zeros(16, 128, 2048).pin_memory() if zero else torch.randn(16, 128, 2048, generator=g).clamp_(min=0).pin_memory()
    tgt = torch.zeros(16, 128, 86); tgt[:, :, i % 86] = 1
    batches.append((rgb, flow, tgt.pin_memory(), ["v"] * 16, torch.zeros(16), torch.full((16,), 128)))
for pre in (True, False, True, False):
    TR.PREFETCH = pre
    tr(batches[:4], m, crit, opt, None, 0, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for e in range(3): tr(batches, m, crit, opt, None, e, dev)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (3 * len(batches))
    print(f"train_one_epoch, {'zero flow' if zero else 'rgb + flow'}, prefetch {pre}: {dt * 1e3:.3f} ms per step = {16 * 128 / dt / 1e6:.2f} M frames/s")
m.engine(train=True).check()
