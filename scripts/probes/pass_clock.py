#!/usr/bin/env python3
"""Where does a split pass lose time against its two launches run alone?  Per-phase shader cycles of the recurrence (workgroup 0, wave 0),
its real time per step and the shader clock of its XCD - in the pass and replayed alone (prego_debug_split_fault mode 4) - and, with
PASS_CLOCK_FF=1, the feed-forward launch's job sums and clock in the pass and alone (mode 3).  Debug library; bench workload.
    python3 scripts/probes/pass_clock.py            # recurrence stamps
    PASS_CLOCK_FF=1 python3 scripts/probes/pass_clock.py"""
import ctypes as C
import os
import sys
import time

FF = os.environ.get("PASS_CLOCK_FF") == "1"
os.environ["PREGO_AMD_DEBUG_LIB"] = "1"
os.environ.setdefault("PREGO_SPLIT_PASS", "3")
os.environ["PREGO_SPLIT_STATS" if FF else "PREGO_GRU_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from prego_amd import _lib, weights as W  # noqa: E402
from prego_amd.config import assembly101_cfg  # noqa: E402
from prego_amd.registry import build_model  # noqa: E402
from prego_amd.workloads import assembly101_eval_lengths  # noqa: E402
import prego_amd.model  # noqa: F401,E402

dev = torch.device("cuda", 0)
cfg = assembly101_cfg(compute_dtype="fp16")
model = build_model(cfg, dev)
model.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20, head_gain=8.0).items()})
model.eval()
eng = model.engine()
lib = _lib.load()
lens = assembly101_eval_lengths(seed=20)
gen = torch.Generator(device=dev)
gen.manual_seed(1234)
rgb = [torch.randn((T, 2048), device=dev, generator=gen).clamp_(min=0) for T in lens]
flow = [torch.randn((T, 2048), device=dev, generator=gen).clamp_(min=0) for T in lens]


def call(mode):
    if mode:
        assert lib.prego_debug_split_fault(eng.h, mode) == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True)
    torch.cuda.synchronize()
    eng.check()
    return (time.perf_counter() - t0) * 1e3, eng.pass_info()


def stamps():
    out = (C.c_uint64 * 8)()
    lib.prego_miniroad_debug_stamps(eng.h, out)
    return [int(x) for x in out]


def report(tag, out, reps):
    if FF:
        v = [x / 1e5 / reps for x in out]
        print(f"{tag}: feed-forward workgroup-ms per pass: pack {v[0]:.1f} layer1 {v[1]:.1f} ln {v[2]:.1f} w_ih {v[3]:.1f} waits {v[4]:.1f} tickets {v[5]:.1f} "
              f"lifetime {v[7]:.1f}; shader clock of its XCDs {out[6] / max(1, out[7]) * 100:.0f} MHz", flush=True)
    else:
        steps = max(1, out[6])
        names = ["rest of gather + mfma", "top -> first segment valid", "reduce + barrier", "gates + publish", "outputs"]
        tot = sum(out[:5])
        print(f"{tag}: recurrence wave 0 of workgroup 0, {steps} steps: {tot / steps:.0f} cycles per step (" +
              ", ".join(f"{names[i]} {out[i] / steps:.0f}" for i in range(5)) + f"), retry rounds per step {out[5] / steps:.2f}, "
              f"{out[7] / 100 / steps:.3f} us per step -> shader clock {tot / max(1, out[7]) * 100:.0f} MHz", flush=True)


ms, info = call(0)
print(f"chunked pass (placement): {ms:.1f} ms {info}", flush=True)
stamps()
call(0)
stamps()
t = [call(0) for _ in range(3)]
print("split passes:", [f"{x[0]:.1f} ms" for x in t], t[-1][1], flush=True)
report("in the pass", stamps(), 3)
alone = 3 if FF else 4
t = [call(alone) for _ in range(2)]
print(f"{'feed-forward' if FF else 'recurrence'} launch alone:", [f"{x[0]:.1f} ms" for x in t], flush=True)
report("alone", stamps(), 2)
