"""reference point only (not used by the product): what torch.matmul (hipBLASLt / rocBLAS) reaches on the projection shapes"""
import torch, time
for (M, N, K) in [(65536, 2048, 4096), (65536, 3072, 2048), (49152, 2048, 4096), (49152, 3072, 2048), (4096, 4096, 4096), (8192, 8192, 8192)]:
    A = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16); B = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    for _ in range(3): C = A @ B.T
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): C = A @ B.T
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 5)
    ts.sort(); t = ts[len(ts) // 2]
    print(f"torch bf16 matmul (bf16 out) M={M} N={N} K={K}: {t:.3f} ms = {2.0*M*N*K/t/1e9:.0f} TFLOP/s")
