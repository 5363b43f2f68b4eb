"""one long clip (recurrence on ONE group = 32 CUs for ~40 ms) on stream A, feed-forward-sized GEMMs on stream B"""
import os
os.environ.setdefault("PREGO_AMD_DEBUG_LIB", "1")      # the prego_debug_* hooks live in libprego_amd_debug.so (include/prego_amd_debug.h)
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W, _lib
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.model  # noqa: F401
dev = torch.device("cuda:0")
cfg = assembly101_cfg(compute_dtype="bf16")
sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
model = build_model(cfg, dev); model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); model.eval()
eng = model.engine(); lib = _lib.load()
ncl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 9
ng = int(sys.argv[3]) if len(sys.argv) > 3 else 40
T = 20000
rgb = [torch.randn((T, 2048), device=dev).clamp_(min=0) for _ in range(ncl)]
M, N, K = 65536, 2048, 4096
A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16); B = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
bias = torch.randn(N, device=dev); Cm = torch.empty(M, N, device=dev)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def fwd():
    with torch.cuda.stream(sA):
        eng.forward_ragged(rgb, None, softmax=True, want_out=True, want_argmax=True)
def gemms(n):
    for _ in range(n):
        lib.prego_debug_gemm_bf16(variant, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(bias.data_ptr()),
                                  C.c_void_p(Cm.data_ptr()), M, N, K, C.c_void_p(sB.cuda_stream))
def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
fwd(); gemms(3); torch.cuda.synchronize(); eng.check()
for rnd in range(2):
    eng.timing_enable(True); a = wall(fwd); k1 = eng.timing_read()
    b = wall(lambda: gemms(ng)); eng.timing_read()
    c = wall(lambda: (fwd(), gemms(ng))); k2 = eng.timing_read(); eng.timing_enable(False); eng.check()
    print(f"{ncl} clip(s) x {T}: pass alone {a:.1f} ms, {ng} gemms (variant {variant}) alone {b:.1f} ms, together {c:.1f} ms; recurrence {k1['gru_ms']:.1f} -> {k2['gru_ms']:.1f} ms")
