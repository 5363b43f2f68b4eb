"""Fixed cost of one recurrence launch: prego_debug_recurrence_only over 128 equal slots (8 groups x 16 columns) for several step
counts, alone on the device; least-squares intercept = what a launch costs before / after its steps.
    python scripts/probes/rec_launch_cost.py"""
import os
os.environ.setdefault("PREGO_AMD_DEBUG_LIB", "1")      # the prego_debug_* hooks live in libprego_amd_debug.so (include/prego_amd_debug.h)
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from prego_amd import _lib
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.model  # noqa: F401

lib = _lib.load()
dev = torch.device("cuda:0")
dt = sys.argv[1] if len(sys.argv) > 1 else "fp16"
cfg = assembly101_cfg(compute_dtype=dt)
m = build_model(cfg, dev)
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
m.eval()
eng = m.engine()
H = 1024
s = torch.cuda.current_stream()
res = []
for cols in (16, 4, 1):
    n_slots = 8 * cols
    xs, ys = [], []
    for steps in (64, 128, 256, 384, 768, 1536):
        rows = n_slots * steps
        gi = (torch.randn(rows, 3 * H, device=dev) * 0.5).to(torch.float16 if dt == "fp16" else torch.bfloat16)
        hr = torch.empty(rows, H, dtype=gi.dtype, device=dev)
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            _lib.check(lib.prego_debug_recurrence_only(eng.h, n_slots, steps, 8, C.c_void_p(gi.data_ptr()), C.c_void_p(hr.data_ptr()), C.c_void_p(s.cuda_stream)))
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        eng.check()
        xs.append(steps); ys.append(best * 1e3)
    A = np.stack([np.array(xs, float), np.ones(len(xs))], 1)
    (slope, icpt), *_ = np.linalg.lstsq(A, np.array(ys), rcond=None)
    res.append({"columns_per_group": cols, "us_per_step": slope, "us_per_launch": icpt, "points": dict(zip(xs, [round(y, 1) for y in ys]))})
    print(json.dumps(res[-1]))
json.dump(res, open("gpurun_out/rec_launch_cost.json", "w"), indent=1)
