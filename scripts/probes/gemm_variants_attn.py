#!/usr/bin/env python3
"""bf16 GEMM kernel variants (prego_debug_gemm_bf16: 0 = 128x128 two-stage, 1 = 256x128 three-stage, 9 = 256x256 two-stage, 12 = ping-pong)
at the causal layer's projection shapes (fp32 C)."""
import ctypes as C, os, sys, time
os.environ["PREGO_AMD_DEBUG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import _lib
lib = _lib.load()
dev = "cuda:0"
for (M, N, K) in ((16384, 6144, 2048), (16384, 2048, 2048)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev); Cm = torch.empty(M, N, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream); p = lambda t: C.c_void_p(t.data_ptr())
    for v in (0, 1, 9, 12, 30):
        try:
            fn = lambda: lib.prego_debug_gemm_bf16(v, p(A), p(B), p(bias), p(Cm), M, N, K, s)
            if fn() != 0:
                print(f"variant {v}: refused"); continue
            for _ in range(3): fn()
            best = 1e9
            for rep in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): fn()
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
            print(f"M {M} N {N} K {K} variant {v}: {best*1e6:.1f} us = {2.0*M*N*K/best/1e12:.0f} TFLOP/s", flush=True)
        except Exception as e:
            print(v, e)
