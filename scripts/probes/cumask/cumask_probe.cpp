// Probe (runs ON THE GPU BOX): which XCDs / CUs a CU-masked stream's workgroups land on.
//   hipcc --offload-arch=gfx950 -O2 cumask_probe.cpp -o cumask_probe && ./cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
__global__ void where_kernel(unsigned* out, int spin) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF;        // HW_REG_XCC_ID[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);                // HW_REG_HW_ID (full): cu_id [11:8], sh_id [12], se_id [15:13]
    out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hw;
  }
  // hold the CU for a while so that workgroups spread over all enabled CUs
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
}
static void run(const char* name, const std::vector<uint32_t>& mask, int nblocks) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
  unsigned* d; hipMalloc(&d, nblocks * 8);
  hipMemset(d, 0xff, nblocks * 8);
  // 64 KB of LDS per workgroup: at most 2 per CU
  hipLaunchKernelGGL(where_kernel, dim3(nblocks), dim3(256), 65536, s, d, 2000000);
  e = hipStreamSynchronize(s);
  std::vector<unsigned> h(nblocks * 2);
  hipMemcpy(h.data(), d, nblocks * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::map<unsigned, int>> per;
  for (int i = 0; i < nblocks; ++i) per[h[2 * i]][(h[2 * i + 1] >> 8) & 0xFF]++;
  printf("%s (%d blocks): %s\n", name, nblocks, hipGetErrorString(e));
  for (auto& x : per) { printf("  xcc %u: %zu distinct (se,sh,cu) ids, blocks:", x.first, x.second.size()); int t = 0; for (auto& c : x.second) t += c.second; printf(" %d\n", t); }
  hipFree(d); hipStreamDestroy(s);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d\n", p.multiProcessorCount);
  std::vector<uint32_t> all(8, 0xFFFFFFFFu);
  run("all 256 bits", all, 256);
  std::vector<uint32_t> lo(8, 0u); lo[0] = 0xFFFFFFFFu; lo[1] = 0xFFFFFFFFu; lo[2] = 0xFFFFFFFFu;
  run("bits 0..95", lo, 96);
  std::vector<uint32_t> il(8, 0u);
  for (int b = 0; b < 256; ++b) if ((b % 8) < 3) il[b / 32] |= 1u << (b % 32);
  run("bits with (b % 8) < 3", il, 96);
  std::vector<uint32_t> hi(8, 0u);
  for (int b = 0; b < 256; ++b) if ((b % 8) >= 3) hi[b / 32] |= 1u << (b % 32);
  run("bits with (b % 8) >= 3", hi, 160);
  std::vector<uint32_t> one(8, 0u);
  for (int b = 0; b < 256; ++b) if ((b % 8) == 5) one[b / 32] |= 1u << (b % 32);
  run("bits with (b % 8) == 5", one, 32);
  return 0;
}
