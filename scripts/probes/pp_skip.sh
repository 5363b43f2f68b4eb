# timing-only builds of the ping-pong GEMM without its fragment reads / without its LDS-DMA (never the product library):
# where the K loop's time goes.  Runs ON THE GPU BOX:  bash scripts/probes/pp_skip.sh
set -e
cd prego_amd/lib && mkdir -p alt
cat > alt/skip_main.hip <<'EOC'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
int launch_gemm_bf16_pingpong_mode(int mode, const void* A, int lda, const void* B, int ldb, const float* bias, void* C, int ldc, int M, int N, int K, bool out_bf16, hipStream_t s, bool f16);
int main(int argc, char** argv) {
  int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
  unsigned short *A, *B; float *bias; void *C;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&bias, N * 4); hipMalloc(&C, (size_t)M * N * 4);
  std::vector<unsigned short> h((size_t)M * K);
  srand(1); for (auto& x : h) x = (unsigned short)(0x3c00 + (rand() & 0x83ff));
  hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4);
  for (int r = 0; r < 3; ++r) launch_gemm_bf16_pingpong_mode(0, A, K, B, K, bias, C, N, M, N, K, true, 0, false);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0); for (int r = 0; r < 10; ++r) launch_gemm_bf16_pingpong_mode(0, A, K, B, K, bias, C, N, M, N, K, true, 0, false); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%s M=%d N=%d K=%d: %.3f ms per GEMM = %.0f TFLOP/s\n", argv[4], M, N, K, best / 10, 2.0 * M * N * K / (best / 10) / 1e9);
  return 0;
}
EOC
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -c alt/skip_main.hip -o alt/skip_main.o
VARS=${VARS:-"base reads dma both"}
for v in $VARS; do
  case $v in base) D="";; reads) D="-DPP_SKIP_READS";; dma) D="-DPP_SKIP_DMA";; both) D="-DPP_SKIP_READS -DPP_SKIP_DMA";; *) D="-D$v";; esac
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-result -Wno-inline-asm $D -c ../csrc/gemm_pp.hip -o alt/gemm_pp_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 alt/skip_main.o alt/gemm_pp_$v.o -o alt/pp_skip_$v
done
for rnd in 1 2 3; do for v in $VARS; do ./alt/pp_skip_$v 49152 2048 4096 $v; ./alt/pp_skip_$v 49152 3072 2048 $v; done; done
