// Classification head + eval epilogue of MiniROAD, fused:
//   logits = relu(h) W_c^T + b_c            f_classification, model/rnn/rnn.py:62-64
//   probs  = softmax(logits, -1)            eval branch,      model/rnn/rnn.py:66-70
//   pred   = argmax(probs, axis=1)          trainer/eval.py:53
// and the scatter from packed time-major rows back to the caller's per-clip [T, C] arrays.
// C <= 128.  One wave owns HM x 16 packed rows and all C (padded to 16*NTC) columns: every W_c fragment it pulls
// from L2 feeds HM row tiles (with one row tile per wave the kernel moved 6x more W_c bytes than relu(h) bytes); the
// row softmax is a 16-lane shuffle reduce over the MFMA accumulator layout (col = lane&15, row = (lane>>4)*4 + reg).
// HBM-bound on reading relu(h) (2 KB/row in bf16).
#include "common.h"
#include "kernels.h"

#define HM 4      // 16-row tiles per wave
template <typename WT, int NTC>
__global__ __launch_bounds__(256) void head_softmax_kernel(
    const WT* __restrict__ Hrelu,      // [nrows][HID] chunk-relative packed rows
    const WT* __restrict__ Wc,         // [16*NTC][HID] zero padded rows
    const float* __restrict__ bc,      // [16*NTC] zero padded
    SlotPlan plan, int row0, int nrows, int HID, int C, int apply_softmax,
    float* const* __restrict__ out_ptrs,     // per clip [T][C] fp32 (probs or logits), entries nullable
    int* const* __restrict__ argmax_ptrs) {  // per clip [T] int32, nullable array / entries
  constexpr bool BF = (sizeof(WT) == 2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int rbase = (blockIdx.x * 4 + wave) * 16 * HM;
  if (rbase >= nrows) return;
  int arow[HM];
#pragma unroll
  for (int m = 0; m < HM; ++m) { arow[m] = rbase + m * 16 + l15; if (arow[m] > nrows - 1) arow[m] = nrows - 1; }

  // ---- where do this wave's 64 rows go?  ONE plan lookup per row, all 64 in parallel (lane L looks up row rbase + L), issued
  // before the K loop: a binary search per row inside the epilogue (16 in a row per lane, ~15 dependent L2 loads each) was
  // 2/3 of this kernel's time.  The wave's first row is located on the scalar path, the lanes search the <= 64 steps behind it.
  __shared__ unsigned long long s_out[4][64], s_arg[4][64];
  {
    typedef const __attribute__((address_space(4))) int* cint_p;
    cint_p ro_c = (cint_p)plan.rowoff;
    int lo = 0, hi = plan.s_max;                 // wave-uniform: rowoff[lo] <= first row < rowoff[hi]
    const int first = __builtin_amdgcn_readfirstlane(row0 + rbase);
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (ro_c[mid] <= first) lo = mid; else hi = mid;
    }
    int myrow = rbase + lane; if (myrow > nrows - 1) myrow = nrows - 1;
    myrow += row0;
    int a = lo, b = lo + 64 < plan.s_max ? lo + 64 : plan.s_max;     // 64 rows span at most 64 steps
    if (plan.rowoff[b] <= myrow) { a = b; b = plan.s_max; }          // (only when a step has < 1 row: never; kept for safety)
    while (b - a > 1) {
      const int mid = (a + b) >> 1;
      if (plan.rowoff[mid] <= myrow) a = mid; else b = mid;
    }
    const int slot = myrow - plan.rowoff[a];
    int k = plan.seg_off[slot];
    const int kend = plan.seg_off[slot + 1];
    while (k + 1 < kend && plan.seg_start[k + 1] <= a) ++k;
    const int clip = plan.seg_clip[k], t = a - plan.seg_start[k];
    float* op = out_ptrs ? out_ptrs[clip] : nullptr;
    int* ap = argmax_ptrs ? argmax_ptrs[clip] : nullptr;
    s_out[wave][lane] = op ? (unsigned long long)(op + (size_t)t * C) : 0ull;
    s_arg[wave][lane] = ap ? (unsigned long long)(ap + t) : 0ull;
  }

  f32x4 acc[HM][NTC];
#pragma unroll
  for (int m = 0; m < HM; ++m)
#pragma unroll
    for (int j = 0; j < NTC; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if constexpr (BF) {
    const bf16_t* bp = (const bf16_t*)Wc + (size_t)l15 * HID + 8 * l4;
    for (int k = 0; k < HID; k += 32) {
      bf16x8 af[HM];
#pragma unroll
      for (int m = 0; m < HM; ++m) af[m] = *(const bf16x8*)((const bf16_t*)Hrelu + (size_t)arow[m] * HID + 8 * l4 + k);
#pragma unroll
      for (int j = 0; j < NTC; ++j) {
        const bf16x8 bfr = *(const bf16x8*)(bp + (size_t)j * 16 * HID + k);
#pragma unroll
        for (int m = 0; m < HM; ++m) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bfr, acc[m][j], 0, 0, 0);
      }
    }
  } else {
    const float* bp = (const float*)Wc + (size_t)l15 * HID + 4 * l4;
    for (int k = 0; k < HID; k += 16) {
      float4 af[HM];
#pragma unroll
      for (int m = 0; m < HM; ++m) af[m] = *(const float4*)((const float*)Hrelu + (size_t)arow[m] * HID + 4 * l4 + k);
#pragma unroll
      for (int j = 0; j < NTC; ++j) {
        const float4 bv = *(const float4*)(bp + (size_t)j * 16 * HID + k);
#pragma unroll
        for (int m = 0; m < HM; ++m) {
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].x, bv.x, acc[m][j], 0, 0, 0);
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].y, bv.y, acc[m][j], 0, 0, 0);
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].z, bv.z, acc[m][j], 0, 0, 0);
          acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].w, bv.w, acc[m][j], 0, 0, 0);
        }
      }
    }
  }

  // lane holds, for rows r = m*16 + l4*4 + e (e = 0..3), columns c = j*16 + l15
#pragma unroll
  for (int m = 0; m < HM; ++m)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v[NTC];
    float mx = -INFINITY;
    int mi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NTC; ++j) {
      const int c = j * 16 + l15;
      v[j] = acc[m][j][e] + bc[c];
      if (c < C && v[j] > mx) { mx = v[j]; mi = c; }    // ascending c: first max wins inside the lane
    }
    // reduce (max, lowest index) over the 16 lanes that share this row
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const float omx = __shfl_xor(mx, o, 64);
      const int omi = __shfl_xor(mi, o, 64);
      if (omx > mx || (omx == mx && omi < mi)) { mx = omx; mi = omi; }
    }
    float s = 0.f;
    if (apply_softmax) {
#pragma unroll
      for (int j = 0; j < NTC; ++j) {
        const int c = j * 16 + l15;
        v[j] = (c < C) ? __expf(v[j] - mx) : 0.f;
        s += v[j];
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
      const float inv = 1.0f / s;
#pragma unroll
      for (int j = 0; j < NTC; ++j) v[j] *= inv;
    }
    const int rr = m * 16 + l4 * 4 + e;            // row inside the wave's tile: its destinations were looked up by lane rr
    if (rbase + rr < nrows) {
      float* op = (float*)s_out[wave][rr];
      if (op) {
#pragma unroll
        for (int j = 0; j < NTC; ++j) {
          const int c = j * 16 + l15;
          if (c < C) op[c] = v[j];
        }
      }
      if (l15 == 0) {
        int* aptr = (int*)s_arg[wave][rr];
        if (aptr) *aptr = mi;
      }
    }
  }
}

int launch_head_softmax(bool bf16, const void* Hrelu, const void* Wc, const float* bc, const SlotPlan& plan,
                        int row0, int nrows, int hid, int C, int apply_softmax,
                        float* const* out_ptrs, int* const* argmax_ptrs, hipStream_t s) {
  if (nrows <= 0) return 0;
  const int ntc = (C + 15) / 16;
  const int grid = (nrows + 64 * HM - 1) / (64 * HM);
#define HL(WT, N)                                                                                             \
  head_softmax_kernel<WT, N><<<grid, 256, 0, s>>>((const WT*)Hrelu, (const WT*)Wc, bc, plan, row0, nrows,     \
                                                  hid, C, apply_softmax, out_ptrs, argmax_ptrs)
#define HD(N)                          \
  case N:                              \
    if (bf16) HL(bf16_t, N);           \
    else HL(float, N);                 \
    break
  switch (ntc) {
    HD(1); HD(2); HD(3); HD(4); HD(5); HD(6); HD(7); HD(8);
    default: return -1;
  }
#undef HD
#undef HL
  return 0;
}
