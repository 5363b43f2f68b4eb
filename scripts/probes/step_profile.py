"""Runs ON THE GPU BOX under rocprofv3: kernel times of the streaming fast path (prego_miniroad_step), 1 and 16 streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.model  # noqa: F401
cfg = assembly101_cfg()
m = build_model(cfg, "cuda:0")
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
m.eval()
eng = m.engine()
for n in (1, 16):
    x = torch.randn(n, 2048, device="cuda").clamp_(min=0)
    f = torch.randn(n, 2048, device="cuda").clamp_(min=0)
    h = torch.zeros(n, 1024, device="cuda")
    out, arg = torch.empty(n, 86, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda")
    for _ in range(20):
        eng.step(x, f, h, out=out, argmax=arg)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(300):
        eng.step(x, f, h, out=out, argmax=arg)
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print(f"n={n}: host issue {t_issue / 300 * 1e6:.1f} us per step, end to end {t_all / 300 * 1e6:.1f} us per step")
