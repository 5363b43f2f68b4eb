"""does an HBM-bound, low-register kernel (a big device copy) overlap with the persistent recurrence?  (stream A: a pass that
is almost only recurrence = 128 clips x 4000 frames with zero-flow; stream B: copies)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.model  # noqa: F401
dev = torch.device("cuda:0")
cfg = assembly101_cfg(compute_dtype="bf16")
sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
model = build_model(cfg, dev); model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); model.eval()
eng = model.engine()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rgb = [torch.randn((T, 2048), device=dev).clamp_(min=0) for _ in range(128)]
src = torch.empty(1 << 30, dtype=torch.float32, device=dev).normal_()      # 4 GiB
dst = torch.empty_like(src)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def fwd():
    with torch.cuda.stream(sA):
        eng.forward_ragged(rgb, None, softmax=True, want_out=True, want_argmax=True)
def copies(n):
    with torch.cuda.stream(sB):
        for _ in range(n):
            dst.copy_(src, non_blocking=True)
def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
fwd(); copies(2); torch.cuda.synchronize(); eng.check()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for rnd in range(2):
    eng.timing_enable(True); a = wall(fwd); kt = eng.timing_read(); eng.timing_enable(False)
    b = wall(lambda: copies(n))
    eng.timing_enable(True); c = wall(lambda: (fwd(), copies(n))); kt2 = eng.timing_read(); eng.timing_enable(False)
    eng.check()
    print(f"pass alone {a:.1f} ms (recurrence {kt['gru_ms']:.1f}); {n} x 4 GiB copies alone {b:.1f} ms ({n*8.59/b:.2f} TB/s r+w); together {c:.1f} ms "
          f"(recurrence {kt2['gru_ms']:.1f}); sum {a+b:.1f}")
