# diagnostic build of the ping-pong GEMM with cycle stamps around the C stores (never the product library)
set -e
cd prego_amd/lib && mkdir -p alt
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -DPP_DIAG -c ../csrc/gemm_pp.hip -o alt/gemm_pp_diag.o
cat > alt/diag_main.hip <<'EOC'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
int launch_gemm_bf16_pingpong_mode(int mode, const void* A, int lda, const void* B, int ldb, const float* bias, void* C, int ldc, int M, int N, int K, bool out_bf16, hipStream_t s);
void pp_diag_print();
int main(int argc, char** argv) {
  int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
  unsigned short *A, *B; float *bias, *C;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&bias, N * 4); hipMalloc(&C, (size_t)M * N * 4);
  std::vector<unsigned short> h((size_t)M * K);
  srand(1); for (auto& x : h) x = (unsigned short)(0x3c00 + (rand() & 0x83ff));     // random bf16 around +-[0.008, 2)
  hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4);
  for (int mode = 1; mode <= 2; ++mode) {
    for (int r = 0; r < 3; ++r) launch_gemm_bf16_pingpong_mode(mode, A, K, B, K, bias, C, N, M, N, K, false, 0);
    pp_diag_print();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int r = 0; r < 10; ++r) launch_gemm_bf16_pingpong_mode(mode, A, K, B, K, bias, C, N, M, N, K, false, 0); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms per GEMM = %.0f TFLOP/s\n", mode, ms / 10, 2.0 * M * N * K / (ms / 10) / 1e9);
    pp_diag_print();
  }
  return 0;
}
EOC
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -DPP_DIAG -c alt/diag_main.hip -o alt/diag_main.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 alt/diag_main.o alt/gemm_pp_diag.o -o alt/pp_diag
