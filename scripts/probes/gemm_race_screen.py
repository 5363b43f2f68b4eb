"""Race screen for the ping-pong GEMM (a new synchronisation structure must be screened over many runs at several sizes):
every run must be bit-identical to the plain 128x128 kernel on fresh random operands.
usage: python scripts/probes/gemm_race_screen.py [runs_per_shape]"""
import os
os.environ.setdefault("PREGO_AMD_DEBUG_LIB", "1")      # the prego_debug_* hooks live in libprego_amd_debug.so (include/prego_amd_debug.h)
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import _lib
lib = _lib.load()
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
load = len(sys.argv) > 2 and sys.argv[2] == "load"       # a second stream streams 1 GiB copies meanwhile (changes the DMA timing)
if load:
    sB = torch.cuda.Stream(); src = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_(); dst = torch.empty_like(src)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def gemm(v, A, B, bias, M, N, K):
    out = torch.full((M, N), float("nan"), device="cuda")
    rc = lib.prego_debug_gemm_bf16(v, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(out.data_ptr()), M, N, K, s)
    assert rc == 0
    return out
bad = 0
for (M, N, K) in [(4096, 256, 128), (4096, 512, 192), (8192, 2048, 2048), (49152, 2048, 4096), (49152, 3072, 2048), (12345, 768, 1024), (65536, 256, 8192)]:
    for r in range(runs):
        g = torch.Generator(device="cuda").manual_seed(1000 * r + M % 997)
        A = (torch.rand((M, K), device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
        B = (torch.rand((N, K), device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
        bias = torch.randn((N,), device="cuda", generator=g)
        if load:
            with torch.cuda.stream(sB):
                for _ in range(4): dst.copy_(src, non_blocking=True)
        ref = gemm(0, A, B, bias, M, N, K)
        got = gemm(12, A, B, bias, M, N, K)
        torch.cuda.synchronize()
        if not torch.equal(ref, got):
            bad += 1
            d = (ref - got).abs()
            print(f"MISMATCH M={M} N={N} K={K} run {r}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}")
    print(f"M={M} N={N} K={K}: {runs} runs screened", flush=True)
print("race screen:", "CLEAN" if bad == 0 else f"{bad} BAD RUNS")
