// DEBUG LIBRARY ONLY (prego_amd/build.py: DEBUG_ONLY_SOURCES; include/prego_amd_debug.h: prego_debug_hog).
// A synthetic neighbour for the recurrence launch of a split pass: what does the recurrence lose beside a launch that
// only COMPUTES on the other XCDs (power, clocks), and what beside one that only MOVES MEMORY (fabric, HBM)?  Workgroups on XCDs below
// xcd_lo leave at once; the others run for `ticks` of s_memrealtime (100 MHz):
//   kind 1: back-to-back MFMAs on registers (v_mfma_f32_16x16x32_f16, 8 independent accumulators per wave), no memory traffic
//   kind 2: streaming reads of `buf` (16 bytes per lane and load, non-temporal, 8 in flight) + a write of every 8th block, no matrix work
//   kind 3: both, alternating
#include "common.h"
#include "kernels.h"

__global__ __launch_bounds__(256) void debug_hog_kernel(int xcd_lo, int kind, unsigned long long ticks, const u32x4* __restrict__ buf,
                                                        u32x4* __restrict__ wbuf, size_t n16, float* __restrict__ sink) {
  const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
  if (xcc < xcd_lo) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = (__bf16)0; b[k] = (__bf16)0; }
  unsigned keep = 0u;
  for (;;) {
    if (kind & 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = op16<f16_t>::mfma(a, b, acc[k]);
    }
    if (kind & 2) {
      u32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = __builtin_nontemporal_load(buf + (i % n16));
        i += stride;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) keep ^= v[k][0] ^ v[k][3];
      __builtin_nontemporal_store(v[0], wbuf + ((i / 8) % n16));
    }
    if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) break;
  }
  if (keep == 0x12345678u || acc[0][0] == 12345.f) sink[0] = acc[1][1] + (float)keep;      // keeps the work alive
}

void launch_debug_hog(int xcd_lo, int kind, int ms, const void* buf, void* wbuf, size_t bytes, float* sink, hipStream_t s) {
  debug_hog_kernel<<<2048, 256, 0, s>>>(xcd_lo, kind, (unsigned long long)ms * 100000ull, (const u32x4*)buf, (u32x4*)wbuf, bytes / 16, sink);
}
