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
#include <cstdlib>

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
        for (int m = 0; m < HM; ++m) acc[m][j] = op16<WT>::mfma(af[m], bfr, acc[m][j]);
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

// ------------------------------------------------------------------------------------------------------
// v2 (bf16, C <= 96): W_c never leaves the register file and the frames ride on the MFMA lane.
//   logits^T[class, frame] = W_c[class, k] . relu(h)^T[k, frame]:  A = W_c fragments (register resident: wave q keeps the K-quarter
//   [256 q, 256 q + 256) of all NTC class tiles = 32 NTC VGPRs), B = relu(h) rows read straight from HBM (a lane's 16 bytes are
//   8 consecutive k of ONE frame: no LDS, no re-layout).  The round-1 kernel re-read W_c from L2 for every 64 rows (3 KB of L2
//   traffic per row beside the 2 KB of HBM) in a K loop of 32 dependent round trips.
// A workgroup is persistent over blocks of 32 frames (two 16-frame tiles); per block a wave loads its 16 fragments, runs 16 NTC
// MFMAs, and the four K-quarter partials meet in LDS (double buffered: one barrier per block); the NEXT block's fragments are
// already in flight during all of that.  Waves 0 and 1 then own one frame tile each: with the classes in the accumulator
// registers and the frame on the lane, softmax / argmax are in-register over 4 NTC values + two shuffles, and the
// probabilities leave as 16-byte stores of four consecutive classes.  The partials are added in K order (bit-reproducible).
// ------------------------------------------------------------------------------------------------------
template <int NTC, typename OT = bf16_t>
__global__ __launch_bounds__(256, 1) void head_softmax_v2_kernel(
    const bf16_t* __restrict__ Hrelu, const bf16_t* __restrict__ Wc, const float* __restrict__ bc, SlotPlan plan, int row0,
    int nrows, int C, int apply_softmax, float* const* __restrict__ out_ptrs, int* const* __restrict__ argmax_ptrs,
    const int2* __restrict__ rowmap /*nullable: (clip, frame) of every row of this chunk, written by the pack kernel*/) {
  constexpr int HID = 1024, KS = 8;                  // k-steps of 32 per K-quarter
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* part = (f32x4*)smem;                        // [2 parities][4 src waves][2 frame tiles][NTC][64 lanes]
  constexpr int PSTRIDE = 4 * 2 * NTC * 64;
  const int lane = threadIdx.x & 63, q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  // resident W_c fragments: A operand, row = class 16 j + l15, k = 256 q + 32 ks + 8 g
  bf16x8 wc[NTC][KS];
#pragma unroll
  for (int j = 0; j < NTC; ++j)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wc[j][ks] = *(const bf16x8*)(Wc + (size_t)(j * 16 + l15) * HID + q * 256 + ks * 32 + 8 * g);
  float bias[NTC][4];                                // class 16 j + 4 g + e
#pragma unroll
  for (int j = 0; j < NTC; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[j][e] = bc[j * 16 + 4 * g + e];

  const int nblk = (nrows + 31) / 32;
  // plan tables as plain pointers (the SlotPlan struct passed by reference would live in scratch)
  const int* __restrict__ p_rowoff = plan.rowoff; const int* __restrict__ p_seg_off = plan.seg_off;
  const int* __restrict__ p_seg_clip = plan.seg_clip; const int* __restrict__ p_seg_start = plan.seg_start;
  const int* __restrict__ p_blk = plan.blk_step;
  const int s_max = plan.s_max;
  bf16x8 cur[2][KS], nxt[2][KS];
#define HEAD_LOAD(DST, BLK)                                                                                             \
  do {                                                                                                                  \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                                     \
      int r_ = (BLK) * 32 + t * 16 + l15; if (r_ > nrows - 1) r_ = nrows - 1;                                           \
      _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                               \
        const u32x4 v_ = __builtin_nontemporal_load((const u32x4*)(Hrelu + (size_t)r_ * HID + q * 256 + ks * 32 + 8 * g)); \
        DST[t][ks] = __builtin_bit_cast(bf16x8, v_);                                                                    \
      }                                                                                                                 \
    }                                                                                                                   \
  } while (0)
  // Vector-memory operations retire IN ORDER, so what an iteration requests and uses itself must be requested BEFORE the next
  // block's rows, and nothing may touch the prefetched rows before the iteration is over.  (The first form of this loop requested the
  // row map entry and the two pointer-table entries of a frame behind the prefetch and let hipcc spread the nxt -> cur copies among the
  // MFMAs, each behind the wait for its load: every iteration waited for the rows it had just requested, 6.5 us per 32 frames.)
  // Now: the row map entry of the NEXT block and the table entries of THIS block go out first (the entry of this block was requested an
  // iteration ago), then the next block's rows, unconditionally (block index clamped: one path, static wait counts); the table entries
  // are first used in the epilogue (16 younger requests stay in flight) and the copies sit behind a scheduling barrier at the very end.
  const int rq = (q & 1) * 16 + l15;                          // my frame inside a block (waves 2, 3 mirror 0, 1: dummy requests)
  auto map_row = [&](int b) { int r = b * 32 + rq; return r < nrows ? r : nrows - 1; };
  int blk = blockIdx.x;
  int2 ct = make_int2(0, 0), ctn = make_int2(0, 0);
  if (blk < nblk) {
    if (rowmap != nullptr) ct = rowmap[map_row(blk)];
    HEAD_LOAD(cur, blk);
  }
  int parity = 0;
  for (; blk < nblk; blk += gridDim.x) {
    const int bn = blk + gridDim.x;
    const int bnc = bn < nblk ? bn : blk;                                     // a block that exists (the last iteration re-requests its own)
    float* op_raw = nullptr; int* ap_raw = nullptr;
    if (rowmap != nullptr) {                                                  // kernel-uniform
      op_raw = out_ptrs ? out_ptrs[ct.x] : nullptr;
      ap_raw = argmax_ptrs ? argmax_ptrs[ct.x] : nullptr;
      ctn = rowmap[map_row(bnc)];
    }
    __builtin_amdgcn_sched_barrier(0);
    HEAD_LOAD(nxt, bnc);                                                      // in flight under this block
    __builtin_amdgcn_sched_barrier(0);
    // destination of my frame (waves 0 / 1 own frame tile q; every lane looks its frame l15 up: 4x redundant, off the MFMA path)
    float* op = nullptr; int* ap = nullptr;
    const int myrow = blk * 32 + q * 16 + l15;
    if (q < 2 && myrow < nrows && rowmap == nullptr) {
      const int row = row0 + myrow;
      // largest step with rowoff[step] <= row: start at the tabulated step of row 32 * (row / 32) and walk forward (a step holds
      // >= 1 row, typically 128: zero or one hop) - a binary search over the 34 k steps was ~16 dependent L2 round trips per block
      int lo = p_blk[row >> 5];
      while (lo + 1 < s_max && p_rowoff[lo + 1] <= row) ++lo;
      const int slot = row - p_rowoff[lo];
      int k = p_seg_off[slot];
      const int kend = p_seg_off[slot + 1];
      while (k + 1 < kend && p_seg_start[k + 1] <= lo) ++k;
      const int clip = p_seg_clip[k], t = lo - p_seg_start[k];
      op = out_ptrs ? out_ptrs[clip] : nullptr;
      if (op) op += (size_t)t * C;
      ap = argmax_ptrs ? argmax_ptrs[clip] : nullptr;
      if (ap) ap += t;
    }
    f32x4 acc[2][NTC];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < NTC; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < NTC; ++j) acc[t][j] = op16<OT>::mfma(wc[j][ks], cur[t][ks], acc[t][j]);
    // K-quarter partials -> LDS (a wave keeps the partial of the frame tile it finishes itself)
    f32x4* pw = part + parity * PSTRIDE;
    if (q != 0) {
#pragma unroll
      for (int j = 0; j < NTC; ++j) pw[((q * 2 + 0) * NTC + j) * 64 + lane] = acc[0][j];
    }
    if (q != 1) {
#pragma unroll
      for (int j = 0; j < NTC; ++j) pw[((q * 2 + 1) * NTC + j) * 64 + lane] = acc[1][j];
    }
    __syncthreads();
    if (q < 2 && myrow < nrows && rowmap != nullptr) {                        // first use of the table entries: behind the MFMAs
      op = op_raw;
      if (op) op += (size_t)ct.y * C;
      ap = ap_raw;
      if (ap) ap += ct.y;
    }
    if (q < 2) {
      float v[NTC][4];
#pragma unroll
      for (int j = 0; j < NTC; ++j) {
        const f32x4 mine = q == 0 ? acc[0][j] : acc[1][j];                    // value select: no runtime register indexing
        f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int src = 0; src < 4; ++src) {                                   // K order 0..3, own partial from registers
          const f32x4 p = (src == q) ? mine : pw[((src * 2 + q) * NTC + j) * 64 + lane];
          sum[0] += p[0]; sum[1] += p[1]; sum[2] += p[2]; sum[3] += p[3];
        }
        v[j][0] = sum[0] + bias[j][0]; v[j][1] = sum[1] + bias[j][1]; v[j][2] = sum[2] + bias[j][2]; v[j][3] = sum[3] + bias[j][3];
      }
      // softmax / argmax over the classes of my frame: in-lane over (j, e), then across the four k-groups g
      float mx = -INFINITY; int mi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < NTC; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = j * 16 + 4 * g + e;
          if (c < C && v[j][e] > mx) { mx = v[j][e]; mi = c; }                 // ascending c inside the lane: first max wins
        }
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        const float omx = __shfl_xor(mx, o, 64);
        const int omi = __shfl_xor(mi, o, 64);
        if (omx > mx || (omx == mx && omi < mi)) { mx = omx; mi = omi; }
      }
      if (apply_softmax) {
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < NTC; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = j * 16 + 4 * g + e;
            v[j][e] = c < C ? __expf(v[j][e] - mx) : 0.f;
            sm += v[j][e];
          }
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        const float inv = 1.0f / sm;
#pragma unroll
        for (int j = 0; j < NTC; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] *= inv;
      }
      if (op) {
#pragma unroll
        for (int j = 0; j < NTC; ++j) {
          const int c = j * 16 + 4 * g;
          if (c + 3 < C && ((C & 3) == 0)) *(float4*)(op + c) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);   // rows are 16-byte aligned iff C % 4 == 0
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c + e < C) op[c + e] = v[j][e];
          }
        }
      }
      if (ap && g == 0) *ap = mi;
    }
    parity ^= 1;
    __builtin_amdgcn_sched_barrier(0);                                        // the copies wait for the prefetch: last thing of the iteration
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) cur[t][ks] = nxt[t][ks];
    ct = ctn;
  }
#undef HEAD_LOAD
}

int launch_head_softmax(bool bf16, const void* Hrelu, const void* Wc, const float* bc, const SlotPlan& plan,
                        int row0, int nrows, int hid, int C, int apply_softmax,
                        float* const* out_ptrs, int* const* argmax_ptrs, hipStream_t s, const void* rowmap, bool f16) {
  if (nrows <= 0) return 0;
  const int ntc = (C + 15) / 16;
  static const bool no_v2 = prego_tune_env("PREGO_HEAD_V1") != nullptr;                // A/B knob
  if (bf16 && hid == 1024 && ntc <= 6 && !no_v2) {     // the register-resident form, for EVERY size: a row's result must not depend on the batch
    const int nblk = (nrows + 31) / 32;
    const int grid2 = nblk < 256 ? nblk : 256;
#define HV2(N)                                                                                                      \
  case N: {                                                                                                         \
    const size_t lds = (size_t)2 * 4 * 2 * N * 64 * 16;                                                             \
    static DeviceOnce once;                                                                                         \
    once.run([&] { (void)hipFuncSetAttribute((const void*)head_softmax_v2_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
                   (void)hipFuncSetAttribute((const void*)head_softmax_v2_kernel<N, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }); \
    if (f16) head_softmax_v2_kernel<N, f16_t><<<grid2, 256, lds, s>>>((const bf16_t*)Hrelu, (const bf16_t*)Wc, bc, plan, row0, nrows, C,  \
                                                                     apply_softmax, out_ptrs, argmax_ptrs, (const int2*)rowmap);    \
    else head_softmax_v2_kernel<N><<<grid2, 256, lds, s>>>((const bf16_t*)Hrelu, (const bf16_t*)Wc, bc, plan, row0, nrows, C,  \
                                                      apply_softmax, out_ptrs, argmax_ptrs, (const int2*)rowmap);    \
    return 0;                                                                                                       \
  }
    switch (ntc) { HV2(1) HV2(2) HV2(3) HV2(4) HV2(5) HV2(6) default: break; }
#undef HV2
  }
  const int grid = (nrows + 64 * HM - 1) / (64 * HM);
#define HL(WT, N)                                                                                             \
  head_softmax_kernel<WT, N><<<grid, 256, 0, s>>>((const WT*)Hrelu, (const WT*)Wc, bc, plan, row0, nrows,     \
                                                  hid, C, apply_softmax, out_ptrs, argmax_ptrs)
#define HD(N)                          \
  case N:                              \
    if (bf16 && f16) HL(f16_t, N);     \
    else if (bf16) HL(bf16_t, N);      \
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
