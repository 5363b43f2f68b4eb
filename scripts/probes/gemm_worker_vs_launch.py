#!/usr/bin/env python3
"""The ping-pong GEMM as one workgroup per tile (the production launch) against the same K loop as a PERSISTENT tile-queue worker
(prego_debug_gemm_worker, one workgroup per CU) at the causal layer's projection shapes: is the per-tile prologue / epilogue what those
K = 2 048 GEMMs lose?  Debug library; fp32 C, bf16 operands."""
import ctypes as C
import os
import sys
import time
os.environ["PREGO_AMD_DEBUG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import _lib
lib = _lib.load()
dev = "cuda:0"
for (M, N, K) in ((16384, 6144, 2048), (16384, 2048, 2048), (16384, 2048, 4096), (49152, 3072, 2048)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    Cm = torch.empty(M, N, device=dev)
    Cw = torch.empty(M, N, device=dev)
    ctr = torch.zeros(1, dtype=torch.int32, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())

    def launch():
        assert lib.prego_debug_gemm_bf16(12, p(A), p(B), p(bias), p(Cm), M, N, K, s) == 0

    def worker():
        ctr.zero_()
        assert lib.prego_debug_gemm_worker(p(A), p(B), p(bias), p(Cw), M, N, K, 0, p(ctr), 256, s) == 0
    for fn in (launch, worker):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    res = {}
    for name, fn in (("launch", launch), ("worker", worker)):
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
        res[name] = best
    same = bool(torch.equal(Cm, Cw))
    fl = 2.0 * M * N * K
    print(f"M {M} N {N} K {K}: one workgroup per tile {res['launch']*1e6:.1f} us = {fl/res['launch']/1e12:.0f} TFLOP/s; persistent worker {res['worker']*1e6:.1f} us = "
          f"{fl/res['worker']/1e12:.0f} TFLOP/s (incl. its counter memset); identical: {same}", flush=True)
