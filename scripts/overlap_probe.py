"""How much of the recurrence phase's idle CU time can other kernels use?  Runs the BASELINE pass on one stream and a
loop of feed-forward-sized GEMMs on a second stream, alone and together (same process, interleaved).
usage: python scripts/overlap_probe.py [n_gemms]"""
import os
os.environ.setdefault("PREGO_AMD_DEBUG_LIB", "1")      # the prego_debug_* hooks live in libprego_amd_debug.so (include/prego_amd_debug.h)
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prego_amd import weights as W, _lib
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
from prego_amd.workloads import assembly101_eval_lengths
import prego_amd.model  # noqa: F401

n_gemms = int(sys.argv[1]) if len(sys.argv) > 1 else 40
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 9
dev = torch.device("cuda:0")
cfg = assembly101_cfg(compute_dtype="bf16")
sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
model = build_model(cfg, dev); model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); model.eval()
eng = model.engine()
lens = assembly101_eval_lengths(seed=20)
gen = torch.Generator(device=dev); gen.manual_seed(1)
rgb = [torch.randn((T, 2048), device=dev, generator=gen).clamp_(min=0) for T in lens]
flow = [torch.randn((T, 2048), device=dev, generator=gen).clamp_(min=0) for T in lens]
lib = _lib.load()
M, N, K = 65536, 2048, 4096
A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
B = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
bias = torch.randn(N, device=dev)
Cm = torch.empty(M, N, device=dev)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def fwd():
    with torch.cuda.stream(sA):
        eng.forward_ragged(rgb, flow, softmax=True, want_out=True, want_argmax=True)


def gemms(n):
    p = C.c_void_p(sB.cuda_stream)
    for _ in range(n):
        lib.prego_debug_gemm_bf16(variant, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(bias.data_ptr()),
                                  C.c_void_p(Cm.data_ptr()), M, N, K, p)


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(2):
    fwd(); gemms(4)
torch.cuda.synchronize(); eng.check()
for rnd in range(3):
    a = wall(fwd)
    b = wall(lambda: gemms(n_gemms))
    both = wall(lambda: (gemms(n_gemms), fwd()))
    both2 = wall(lambda: (fwd(), gemms(n_gemms)))
    eng.check()
    print(f"round {rnd}: pass alone {a:.1f} ms; {n_gemms} GEMMs alone {b:.1f} ms ({b/n_gemms:.3f} each); together "
          f"{both:.1f} ms (gemms enqueued first) / {both2:.1f} ms (pass first); sum {a+b:.1f}", flush=True)
