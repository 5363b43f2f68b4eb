"""GPU tests of the projection GEMM kernels through the C ABI's measurement hook (prego_debug_gemm_bf16): the production
ping-pong kernel against the plain 128x128 kernel (bit-exact: same MFMA instruction, same
ascending-k accumulation order) and against an fp32 torch matmul of the same bf16 operands (tolerance), on ragged M, the
minimum K, and a grid larger than the chip."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _gemm(lib, variant, A, B, bias, M, N, K):
    out = torch.full((M, N), float("nan"), device="cuda")
    rc = lib.prego_debug_gemm_bf16(variant, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(bias.data_ptr()),
                                   C.c_void_p(out.data_ptr()), M, N, K, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.prego_last_error()
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (4113, 512, 1024), (70001, 256, 192), (65536, 2048, 256)])
def test_pingpong_gemm_matches_plain_kernel_and_fp32(M, N, K):
    from prego_amd import _lib
    lib = _lib.load_debug()          # kernel-level hook prego_debug_gemm_bf16: only in libprego_amd_debug.so
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.rand((M, K), device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    B = (torch.rand((N, K), device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    bias = torch.randn((N,), device="cuda", generator=g)
    plain = _gemm(lib, 0, A, B, bias, M, N, K)            # 128x128 two-stage kernel
    for variant in (12,):                                 # ping-pong (production)
        got = _gemm(lib, variant, A, B, bias, M, N, K)
        assert not torch.isnan(got).any(), f"variant {variant}: unwritten output elements"
        assert torch.equal(got, plain), f"variant {variant}: differs from the plain kernel, max {float((got - plain).abs().max()):.3e}"
    rows = torch.randint(0, M, (256,), device="cuda", generator=g)
    ref = A[rows].float() @ B.float().T + bias
    assert float((plain[rows] - ref).abs().max()) < 2e-3 * (K ** 0.5)      # fp32 accumulation of K products in [-1, 1]
