"""host time of every segment of a training step (no synchronisation inside the loop): is the step GPU-bound or host-bound?"""
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
seg = {k: 0.0 for k in ("train()", "forward", "loss", "zero_grad", "backward", "step")}
def step(acc):
    t0 = time.perf_counter(); m.train(); t1 = time.perf_counter()
    out = m(rgb, flow); t2 = time.perf_counter()
    loss = crit(out, tgt); t3 = time.perf_counter()
    opt.zero_grad(set_to_none=True); t4 = time.perf_counter()
    loss.backward(); t5 = time.perf_counter()
    opt.step(); t6 = time.perf_counter()
    if acc:
        for k, d in zip(seg, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
            seg[k] += d
for _ in range(10): step(False)
torch.cuda.synchronize(); n = 200; t0 = time.perf_counter()
for _ in range(n): step(True)
host = time.perf_counter() - t0
torch.cuda.synchronize(); wall = time.perf_counter() - t0
print(f"per step: host enqueue {host / n * 1e3:.3f} ms, wall {wall / n * 1e3:.3f} ms;  " + "  ".join(f"{k} {v / n * 1e6:.0f} us" for k, v in seg.items()))
# where the backward's host time goes: the C call itself against the Python / autograd around it
eng = m.engine(train=True)
lib = eng.lib
orig = lib.prego_miniroad_backward
acc = {"c": 0.0, "n": 0}
def timed(*a):
    t = time.perf_counter(); r = orig(*a); acc["c"] += time.perf_counter() - t; acc["n"] += 1; return r
lib.prego_miniroad_backward = timed
orig_f = lib.prego_miniroad_forward
accf = {"c": 0.0}
def timed_f(*a):
    t = time.perf_counter(); r = orig_f(*a); accf["c"] += time.perf_counter() - t; return r
lib.prego_miniroad_forward = timed_f
torch.cuda.synchronize()
for _ in range(n): step(False)
torch.cuda.synchronize()
print(f"C call prego_miniroad_backward: {acc['c'] / acc['n'] * 1e6:.0f} us per step; prego_miniroad_forward: {accf['c'] / n * 1e6:.0f} us")
