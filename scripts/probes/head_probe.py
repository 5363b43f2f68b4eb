"""The head kernel alone (prego_debug_head_only): time against the number of rows - what is fixed per launch, what per row.
    python scripts/probes/head_probe.py"""
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
cfg = assembly101_cfg()
m = build_model(cfg, "cuda:0")
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
m.eval()
eng = m.engine()
s = torch.cuda.current_stream()
use_map = "nomap" not in sys.argv
xs, ys = [], []
for steps in (8, 64, 128, 256, 384, 768):
    n_slots = 128
    rows = n_slots * steps
    hr = torch.rand(rows, 1024, device="cuda").to(torch.float16)
    out = torch.empty(rows, 86, device="cuda")
    arg = torch.empty(rows, dtype=torch.int32, device="cuda")
    r = torch.arange(rows, device="cuda", dtype=torch.int32)
    rowmap = torch.stack([r % n_slots, r // n_slots], 1).contiguous()        # (clip, frame) of every packed row
    best = 1e9
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        _lib.check(lib.prego_debug_head_only(eng.h, n_slots, steps, C.c_void_p(hr.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(arg.data_ptr()),
                                             C.c_void_p(rowmap.data_ptr()) if use_map else None, C.c_void_p(s.cuda_stream)))
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3)
    xs.append(rows); ys.append(best)
    print(f"{rows} rows: {best:.1f} us ({rows * 2048 / best / 1e6:.2f} TB/s of relu(h) rows)")
# the two destination paths (row map / plan tables) must agree bit for bit
o2 = torch.empty_like(out); a2 = torch.empty_like(arg)
_lib.check(lib.prego_debug_head_only(eng.h, n_slots, steps, C.c_void_p(hr.data_ptr()), C.c_void_p(o2.data_ptr()), C.c_void_p(a2.data_ptr()), None,
                                     C.c_void_p(s.cuda_stream)))
_lib.check(lib.prego_debug_head_only(eng.h, n_slots, steps, C.c_void_p(hr.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(arg.data_ptr()),
                                     C.c_void_p(rowmap.data_ptr()), C.c_void_p(s.cuda_stream)))
torch.cuda.synchronize()
print("row map vs plan tables: identical outputs", bool(torch.equal(out, o2) and torch.equal(arg, a2)),
      "| probabilities sum to 1:", bool(torch.allclose(out.sum(1), torch.ones_like(out[:, 0]), atol=1e-4)))
A = np.stack([np.array(xs, float), np.ones(len(xs))], 1)
(slope, icpt), *_ = np.linalg.lstsq(A, np.array(ys), rcond=None)
print(json.dumps({"us_per_1000_rows": slope * 1000, "us_per_launch": icpt}))
