#!/usr/bin/env python3
"""Power or fabric?  (round-5 verdict, item 2b.)  The recurrence launch of a split pass, replayed ALONE on XCDs 0-2 (prego_debug_split_fault
mode 4), beside a SYNTHETIC neighbour on XCDs 3-7 (prego_debug_hog, its own stream): nothing / MFMAs only (no memory traffic) / memory
streaming only (no matrix work) / both - against the real pass.  Per case: the launch's time, us per step, per-phase cycles and the
s_memtime / s_memrealtime ratio of its XCD (wave 0 of workgroup 0).  Debug library, bench workload."""
import ctypes as C
import os
import sys
import time

os.environ["PREGO_AMD_DEBUG_LIB"] = "1"
os.environ.setdefault("PREGO_SPLIT_PASS", "3")
os.environ["PREGO_GRU_STAMPS"] = "1"
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
hog_stream = torch.cuda.Stream(dev, priority=0)
rbuf = torch.ones(1 << 30, dtype=torch.uint8, device=dev)          # 1 GiB each: far past the 256 MB Infinity Cache
wbuf = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
sink = torch.zeros(4, device=dev)


def stamps():
    out = (C.c_uint64 * 8)()
    lib.prego_miniroad_debug_stamps(eng.h, out)
    return [int(x) for x in out]


def call(mode, hog=0):
    if mode:
        assert lib.prego_debug_split_fault(eng.h, mode) == 0
    torch.cuda.synchronize()
    if hog:
        with torch.cuda.stream(hog_stream):
            assert lib.prego_debug_hog(hog, 3, 130, C.c_void_p(rbuf.data_ptr()), C.c_void_p(wbuf.data_ptr()), rbuf.numel(), C.c_void_p(sink.data_ptr()),
                                       C.c_void_p(hog_stream.cuda_stream)) == 0
        time.sleep(0.005)                      # the neighbour is resident before the recurrence is launched
    t0 = time.perf_counter()
    eng.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True)
    eng.check()
    dt = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    return dt, eng.pass_info()


def report(tag, out, ms):
    steps = max(1, out[6])
    names = ["gather rest + mfma", "top -> first segment", "reduce + barrier", "gates + publish", "outputs"]
    tot = sum(out[:5])
    print(f"{tag:34s}: {ms:6.1f} ms per call, {out[7] / 100 / steps:.3f} us per step, {tot / steps:.0f} cycles per step (" +
          ", ".join(f"{names[i]} {out[i] / steps:.0f}" for i in range(5)) + f"), s_memtime / s_memrealtime {tot / max(1, out[7]) * 100:.0f} MHz", flush=True)


call(0)
stamps()
call(0)
stamps()
ms = min(call(0)[0] for _ in range(3))
report("the real pass", stamps(), ms)
for name, hog in (("alone", 0), ("beside MFMAs only (XCDs 3-7)", 1), ("beside memory streaming only", 2), ("beside both", 3), ("alone again", 0)):
    ms = min(call(4, hog)[0] for _ in range(2))
    report("recurrence " + name, stamps(), ms)
