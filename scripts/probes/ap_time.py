"""time prego_perframe_ap on an eval-set-sized matrix and check it against the host path on a slice"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from prego_amd.metrics import perframe_average_precision_device, perframe_average_precision
n, C = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2306143, 86
g = torch.Generator(device="cuda"); g.manual_seed(3)
pr = torch.softmax(torch.randn(n, C, device="cuda", generator=g) * 3, -1)
if "--distinct" not in sys.argv:
    pr = (pr * 4096).round() / 4096            # plenty of ties
lab = torch.randint(0, C, (n,), device="cuda", generator=g)
gt = torch.zeros(n, C, device="cuda"); gt[torch.arange(n, device="cuda"), lab] = 1
names = [f"c{i}" for i in range(C)]
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = perframe_average_precision_device(pr, gt, names)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"AP of {n} x {C}: {dt*1e3:.1f} ms, mAP {r['mean_AP']:.6f}")
m = 200000
h = perframe_average_precision(pr[:m].cpu().numpy(), gt[:m].cpu().numpy(), names)
d = perframe_average_precision_device(pr[:m].contiguous(), gt[:m].contiguous(), names)
print("slice check: host", h["mean_AP"], "device", d["mean_AP"], "max per-class diff", max(abs(h["per_class_AP"][k] - d["per_class_AP"][k]) for k in h["per_class_AP"]))
