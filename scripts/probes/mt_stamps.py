"""(needs a diagnostic build: HIPCC flags += -DGRU_MT_STAMPS; the stamp branches cost the kernel ~8 %)
per-phase cycles of a tile-slot of the pipelined multi-tile recurrence kernel (PREGO_GRU_STAMPS=1): 512 clips x 512 frames in
256 / 512 slots (2 / 4 tiles per group); workgroup 0, wave 0"""
import os
os.environ.setdefault("PREGO_AMD_DEBUG_LIB", "1")      # the prego_debug_* hooks live in libprego_amd_debug.so (include/prego_amd_debug.h)
import os, sys, ctypes as C
os.environ["PREGO_GRU_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.model  # noqa: F401

cfg = assembly101_cfg(compute_dtype="fp16")
m = build_model(cfg, "cuda:0")
m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20, head_gain=8.0).items()})
m.eval()
eng = m.engine()
rgb = [torch.randn((512, 2048), device="cuda").clamp_(min=0) for _ in range(512)]
out = (C.c_uint64 * 8)()
eng.forward_ragged(rgb, None); eng.check()
eng.lib.prego_miniroad_debug_stamps(eng.h, out)
eng.forward_ragged(rgb, None); eng.check()
eng.lib.prego_miniroad_debug_stamps(eng.h, out)
names = ["gather wait", "LDS fetch + MFMA (+ tag check + next DMA)", "REDONE TILES (count)", "gi DMA issue", "reduce write + barrier", "gates", "stores"]
n = max(1, out[7])
print(f"PREGO_PLAN_SLOTS={os.environ.get('PREGO_PLAN_SLOTS')}: tile-slots {n}, cycles per slot {sum(out[i] for i in range(7) if i != 2) / n:.0f}: " +
      ", ".join(f"{names[i]} {out[i] / n:.0f}" if i != 2 else f"{names[i]} {out[i]}" for i in range(7)))
