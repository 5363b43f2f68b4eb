// probe: block -> XCC id map for a 256-block launch at one workgroup per CU (forced by 128 KB LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 1) void k(int* xcc, unsigned* cnt, unsigned* total) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = 1;
  if (threadIdx.x == 0) {
    int id = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
    xcc[blockIdx.x] = id;
    unsigned t = __hip_atomic_fetch_add(&cnt[id], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    xcc[256 + blockIdx.x] = (int)t;
    __hip_atomic_fetch_add(total, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(4);
    xcc[512 + blockIdx.x] = (int)spins;
  }
}
int main() {
  int* d; unsigned* c;
  hipMalloc(&d, 768 * 4); hipMalloc(&c, 64 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(c, 0, 64 * 4);
    k<<<256, 256, 131072>>>(d, c, c + 32);
    hipError_t e = hipDeviceSynchronize();
    std::vector<int> h(768); std::vector<unsigned> hc(64);
    hipMemcpy(h.data(), d, 768 * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, 64 * 4, hipMemcpyDeviceToHost);
    printf("rep %d err %d counts:", rep, (int)e); for (int i = 0; i < 8; ++i) printf(" %u", hc[i]); printf(" total %u\n", hc[32]);
    printf(" xcc of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %d", h[i]); printf("\n");
    int same = 0; for (int i = 0; i < 256; ++i) same += (h[i] == h[i % 8]); printf(" blocks with xcc == xcc[b%%8]: %d/256, max spins %d\n", same, *std::max_element(h.begin() + 512, h.end()));
  }
  return 0;
}
