"""per-phase cycle breakdown of the recurrence kernel (debug build knobs: PREGO_GRU_STAMPS=1)"""
import os
os.environ.setdefault("PREGO_AMD_DEBUG_LIB", "1")      # the prego_debug_* hooks live in libprego_amd_debug.so (include/prego_amd_debug.h)
import os, sys, ctypes as C
os.environ["PREGO_GRU_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.model

dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg = assembly101_cfg(compute_dtype=dtype)
sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
m = build_model(cfg, "cuda:0"); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.eval()
eng = m.engine()
T = 2000
for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,128,182,400").split(",")]:
    rgb = [torch.randn((T, 2048), device="cuda").clamp_(min=0) for _ in range(n)]
    eng.forward_ragged(rgb, None); eng.check()
    out = (C.c_uint64 * 8)(); eng.lib.prego_miniroad_debug_stamps(eng.h, out)
    torch.cuda.synchronize(); ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    eng.timing_enable(True); eng.forward_ragged(rgb, None); kt = eng.timing_read(); eng.timing_enable(False)
    eng.lib.prego_miniroad_debug_stamps(eng.h, out)
    steps = out[6]
    names = ["rest of gather + mfma", "step top -> first segment valid", "reduce+barrier", "gates+publish", "outputs"]
    tot = sum(out[i] for i in range(5))
    print(f"clips={n} {dtype}: kernel {kt['gru_ms']*1e3/steps:.2f} us/step (events); wave0 cycles/step total {tot/steps:.0f}: " +
          ", ".join(f"{names[i]} {out[i]/steps:.0f}" for i in range(5)) + f"; retry rounds/step {out[5]/steps:.2f}")
