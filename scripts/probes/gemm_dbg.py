import os
os.environ.setdefault("PREGO_AMD_DEBUG_LIB", "1")      # the prego_debug_* hooks live in libprego_amd_debug.so (include/prego_amd_debug.h)
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import _lib
lib = _lib.load()
v = int(sys.argv[1]); M, N, K = [int(x) for x in sys.argv[2:5]]
torch.manual_seed(0)
A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16); B = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
bias = torch.zeros(N, device="cuda"); Cm = torch.full((M, N), -77.0, device="cuda")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    lib.prego_debug_gemm_bf16(v, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(Cm.data_ptr()), M, N, K, s)
    torch.cuda.synchronize()
    ref = A.float() @ B.float().T
    err = (Cm - ref).abs()
    print(f"rep {rep}: max err {err.max().item():.3e}; bad elements {(err > 0.05).sum().item()} of {M*N}")
    bad = (err > 0.05)
    if bad.any():
        bt = bad.view(M // 64, 64, N // 32, 32).any(3).any(1)   # 64x32 blocks
        print("bad 64x32 blocks (rows=M/64, cols=N/32):"); print(bt.int().cpu().numpy()[:8, :16])
        # K-partial check: is the result a partial sum?
        i, j = [int(x[0]) for x in torch.nonzero(bad)[0:1].T]
        parts = [(A[i, k*64:(k+1)*64].float() @ B[j, k*64:(k+1)*64].float()).item() for k in range(K // 64)]
        print("first bad element", i, j, "got", Cm[i, j].item(), "ref", ref[i, j].item(), "k-tile partials", [round(p, 3) for p in parts[:8]])
