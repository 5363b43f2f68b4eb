"""do two torch streams run kernels concurrently on this box?  (spin kernel on one, GEMMs on the other)"""
import os
os.environ.setdefault("PREGO_AMD_DEBUG_LIB", "1")      # the prego_debug_* hooks live in libprego_amd_debug.so (include/prego_amd_debug.h)
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
M, N, K = 65536, 2048, 4096
A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16); B = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
bias = torch.randn(N, device=dev); Cm = torch.empty(M, N, device=dev)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
print("stream handles", hex(sA.cuda_stream), hex(sB.cuda_stream))
def gemms(n, s):
    for _ in range(n):
        lib.prego_debug_gemm_bf16(9, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(bias.data_ptr()),
                                  C.c_void_p(Cm.data_ptr()), M, N, K, C.c_void_p(s.cuda_stream))
def spin(s, cycles):
    with torch.cuda.stream(s):
        torch.cuda._sleep(cycles)
def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
gemms(3, sB); spin(sA, 1000)
for rnd in range(2):
    a = wall(lambda: spin(sA, 40_000_000)); b = wall(lambda: gemms(20, sB))
    c = wall(lambda: (spin(sA, 40_000_000), gemms(20, sB)))
    d = wall(lambda: (gemms(10, sA), gemms(10, sB)))
    print(f"spin alone {a:.1f} ms, 20 gemms alone {b:.1f} ms, together {c:.1f} ms; 10+10 gemms on two streams {d:.1f} ms")
