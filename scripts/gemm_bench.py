"""bf16 NT GEMM microbenchmark: variants interleaved in ONE process (cdna_hip_programming.md rule 24), random data,
checked against torch.  usage: python scripts/gemm_bench.py [variants e.g. 9,13] [M] [N] [K]
variants: 0 = 128x128 two-stage, 1 = 256x128 three-stage counted-vmcnt, 9 = 256x256 two-stage (previous production), 12 = ping-pong (production)"""
import os
os.environ.setdefault("PREGO_AMD_DEBUG_LIB", "1")      # the prego_debug_* hooks live in libprego_amd_debug.so (include/prego_amd_debug.h)
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prego_amd import _lib

lib = _lib.load()
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
K = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
torch.manual_seed(0)
A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
B = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
Cs = {v: torch.empty(M, N, device="cuda") for v in variants}
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ref = None
for v in variants:
    rc = lib.prego_debug_gemm_bf16(v, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(bias.data_ptr()),
                                   C.c_void_p(Cs[v].data_ptr()), M, N, K, s)
    assert rc == 0, lib.prego_last_error()
    torch.cuda.synchronize()
    if ref is None:
        rows = torch.randint(0, M, (512,), device="cuda")
        ref = (A[rows].float() @ B.float().T + bias)
    err = (Cs[v][rows] - ref).abs().max().item()
    print(f"variant {v}: max abs err vs fp32 reference on 512 rows {err:.3e}")
    assert err < 5e-2 or v in (21, 22, 23), err        # 21-23: timing-only builds of the four-wave kernel (wrong results by construction)
times = {v: [] for v in variants}
for rnd in range(12):
    for v in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.prego_debug_gemm_bf16(v, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(bias.data_ptr()),
                                      C.c_void_p(Cs[v].data_ptr()), M, N, K, s)
        e1.record(); torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 5)
fl = 2.0 * M * N * K
for v in variants:
    t = sorted(times[v][2:])
    print(f"variant {v}: M={M} N={N} K={K} median {t[len(t)//2]:.3f} ms = {fl/t[len(t)//2]/1e9:.0f} TFLOP/s, min {t[0]:.3f} ms = {fl/t[0]/1e9:.0f} TFLOP/s")
