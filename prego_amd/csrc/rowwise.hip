// Row-wise, HBM-bound kernels of the MiniROAD path:
//   pack_rows      feature streaming: per-clip fp32 [T,D] rgb/flow  ->  packed time-major X[row, Din]
//                  (replaces torch.cat at model/rnn/rnn.py:53 and the batch_first layout; never
//                  materialises the concat in fp32)
//   ln_relu_rows   LayerNorm(E) + ReLU (model/rnn/rnn.py:41-42) on the layer1 GEMM output
// Both are pure streaming: 16 B per lane, one pass, no LDS.
#include "common.h"
#include "kernels.h"

// One 256-thread block per packed row.  D_rgb, D_flow multiples of 8.
// OutT = bf16_t (MFMA operand) or float (fp32 parity mode).
// IN16: the per-clip feature arrays already hold the operand type (PREGO_FWD_IN16: a feeder that keeps 16-bit features in pinned
// host memory ships half the bytes over PCIe; the rounding fp32 -> bf16 / fp16 then happened on the host, same RNE): plain copy.
template <typename OutT, bool IN16 = false>
__global__ __launch_bounds__(256) void pack_rows_kernel(
    const float* const* __restrict__ rgb_ptrs, const float* const* __restrict__ flow_ptrs, SlotPlan plan,
    int row0, int nrows, int d_rgb, int d_flow, OutT* __restrict__ X, int2* __restrict__ rowmap /*nullable: [nrows] (clip, frame)*/) {
  const int din = d_rgb + d_flow;
  for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
    const int row = row0 + r;
    int clip, t;
    plan_clip_of_row(plan, row, clip, t);
    // the row -> (clip, frame) map this kernel had to compute anyway, kept for the head kernel's scatter (8 B per row instead of
    // a chain of dependent table look-ups per row there)
    if (rowmap != nullptr && threadIdx.x == 0) rowmap[r] = make_int2(clip, t);
    const float* rgb = rgb_ptrs ? rgb_ptrs[clip] : nullptr;
    const float* flow = flow_ptrs ? flow_ptrs[clip] : nullptr;   // nullptr = all-zero flow half
    OutT* dst = X + (size_t)r * din;
    if constexpr (IN16) {
      const bf16_t* rgb16 = (const bf16_t*)rgb;
      const bf16_t* flow16 = (const bf16_t*)flow;
      for (int c = threadIdx.x * 8; c < din; c += 256 * 8) {
        const bf16_t* src = (c < d_rgb) ? (rgb16 ? rgb16 + (size_t)t * d_rgb + c : nullptr)
                                        : (flow16 ? flow16 + (size_t)t * d_flow + (c - d_rgb) : nullptr);
        const u32x4 v = src ? __builtin_nontemporal_load((const u32x4*)src) : (u32x4){0u, 0u, 0u, 0u};
        *(u32x4*)(dst + c) = v;
      }
      continue;
    }
    for (int c = threadIdx.x * 8; c < din; c += 256 * 8) {
      float4 a, b;
      const float* src = (c < d_rgb) ? (rgb ? rgb + (size_t)t * d_rgb + c : nullptr)
                                     : (flow ? flow + (size_t)t * d_flow + (c - d_rgb) : nullptr);
      if (src) {
        a = nt_load4(src);
        b = nt_load4(src + 4);
      } else {
        a = make_float4(0, 0, 0, 0);
        b = a;
      }
      if constexpr (is_x2<OutT>::value) {        // split row [din hi | din lo] of raw fp16 (common.h)
        uint4 oh, ol;
        x2_split2(a.x, a.y, oh.x, ol.x); x2_split2(a.z, a.w, oh.y, ol.y);
        x2_split2(b.x, b.y, oh.z, ol.z); x2_split2(b.z, b.w, oh.w, ol.w);
        bf16_t* d16 = (bf16_t*)dst;              // dst = X + r * din of 4-byte elements = row start
        *(uint4*)(d16 + c) = oh;
        *(uint4*)(d16 + din + c) = ol;
      } else if constexpr (sizeof(OutT) == 2) {
        uint4 o;
        o.x = op16<OutT>::pack2_sat(a.x, a.y); o.y = op16<OutT>::pack2_sat(a.z, a.w);
        o.z = op16<OutT>::pack2_sat(b.x, b.y); o.w = op16<OutT>::pack2_sat(b.z, b.w);
        *(uint4*)(dst + c) = o;
      } else {
        *(float4*)(dst + c) = a;
        *((float4*)(dst + c) + 1) = b;
      }
    }
  }
}

// One wave per row, 4 rows per block.  E multiple of 512 and <= 4096: each lane owns 8 consecutive columns per
// 512-column slab (two 16-byte loads, ONE 16-byte bf16 store: 8-byte stores run at 0.5-0.7x the 16-byte rate).
// Two-pass statistics in registers (mean, then centred sum of squares) = what torch's CPU LayerNorm computes up to
// summation order; biased variance, eps inside the sqrt.
template <typename OutT, int MAXV, typename InT = float>
__global__ __launch_bounds__(256) void ln_relu_rows_kernel(
    const InT* __restrict__ Y, const float* __restrict__ gamma, const float* __restrict__ beta,
    int nrows, int E, float eps, OutT* __restrict__ out, float* __restrict__ stats /*nullable [nrows][2]*/,
    float drop_p, unsigned long long seed, int row0_abs, int relu, GruArm arm) {
  if (arm.hx != nullptr) {            // re-arm the recurrence's exchange buffers and rendezvous words on the side (kernels.h: GruArm)
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = i0; k < arm.words_per_buf / 4; k += stride) {
      ((uint4*)arm.hx)[k] = make_uint4(arm.pattern, arm.pattern, arm.pattern, arm.pattern);
      ((uint4*)(arm.hx + arm.words_per_buf))[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (arm.sync != nullptr && i0 < 16) arm.sync[i0] = 0u;
  }
  // training-mode Dropout(p) after the ReLU (rnn.py:43): stateless mask = hash(seed, absolute element index),
  // kept values scaled by 1/(1-p); the backward kernel regenerates the same mask
  const unsigned thresh = drop_p > 0.f ? (unsigned)(drop_p * 4294967296.0) : 0u;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nv = E / 512;                      // 8-column groups per lane
  for (int r = blockIdx.x * 4 + wave; r < nrows; r += gridDim.x * 4) {
    const InT* y = Y + (size_t)r * E;
    float v[MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (i < nv) {
        float4 a, b;
        if constexpr (sizeof(InT) == 2) {          // bf16 rows (inference path with bf16 intermediates): 8 elements = one 16-byte load
          const u32x4 w = __builtin_nontemporal_load((const u32x4*)(y + (i * 64 + lane) * 8));
          a = make_float4(op16<InT>::lo(w[0]), op16<InT>::hi(w[0]), op16<InT>::lo(w[1]), op16<InT>::hi(w[1]));
          b = make_float4(op16<InT>::lo(w[2]), op16<InT>::hi(w[2]), op16<InT>::lo(w[3]), op16<InT>::hi(w[3]));
        } else {
          a = nt_load4((const float*)y + (i * 64 + lane) * 8);
          b = nt_load4((const float*)y + (i * 64 + lane) * 8 + 4);
        }
        v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w; v[i][4] = b.x; v[i][5] = b.y; v[i][6] = b.z; v[i][7] = b.w;
        s += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
      }
    const float mu = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (i < nv) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          // the square is rounded before it is added (no fused multiply-add): the split pass's LayerNorm job (ff_pass.hip: ffp_ln) must
          // reproduce this row bit for bit, and which of the two forms -ffp-contract=fast picks depends on the code around it
#pragma clang fp contract(off)
          const float d = v[i][k] - mu;
          const float dd = d * d;
          q = q + dd;
        }
      }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + eps);
    if (stats && lane == 0) { stats[2 * r] = mu; stats[2 * r + 1] = rstd; }
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (i < nv) {
        const int c = (i * 64 + lane) * 8;
        const float4 g0 = *(const float4*)(gamma + c), g1 = *(const float4*)(gamma + c + 4);
        const float4 b0 = *(const float4*)(beta + c), b1 = *(const float4*)(beta + c + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          o[k] = (v[i][k] - mu) * rstd * gg[k] + bb[k];
          if (relu) o[k] = fmaxf(o[k], 0.f);
          if (thresh) o[k] = dropout_keep_(seed, (size_t)(row0_abs + r) * E + c + k, thresh) ? o[k] * keep_scale : 0.f;
        }
        if constexpr (is_x2<OutT>::value) {      // split row [E hi | E lo]: the W_ih GEMM's A operand
          uint4 wh, wl;
          x2_split2(o[0], o[1], wh.x, wl.x); x2_split2(o[2], o[3], wh.y, wl.y);
          x2_split2(o[4], o[5], wh.z, wl.z); x2_split2(o[6], o[7], wh.w, wl.w);
          bf16_t* o16 = (bf16_t*)out + (size_t)r * 2 * E;
          *(uint4*)(o16 + c) = wh;
          *(uint4*)(o16 + E + c) = wl;
        } else if constexpr (sizeof(OutT) == 2) {
          uint4 w;
          w.x = op16<OutT>::pack2_sat(o[0], o[1]); w.y = op16<OutT>::pack2_sat(o[2], o[3]); w.z = op16<OutT>::pack2_sat(o[4], o[5]); w.w = op16<OutT>::pack2_sat(o[6], o[7]);
          *(uint4*)(out + (size_t)r * E + c) = w;
        } else {
          *(float4*)(out + (size_t)r * E + c) = make_float4(o[0], o[1], o[2], o[3]);
          *(float4*)(out + (size_t)r * E + c + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
      }
  }
}

// fp32 -> bf16 weight conversion (set_weights)
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = f2bf(src[i]);
}

// dst[r][0:cols_dst] = bf16/f32(src[r][0:cols_src]) zero-padded to rows_dst x cols_dst
template <typename OutT>
__global__ void pad_convert_kernel(const float* __restrict__ src, int rows_src, int cols_src, int ld_src,
                                   OutT* __restrict__ dst, int rows_dst, int cols_dst) {
  size_t n = (size_t)rows_dst * cols_dst;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int r = (int)(i / cols_dst), c = (int)(i % cols_dst);
    float v = (r < rows_src && c < cols_src) ? src[(size_t)r * ld_src + c] : 0.f;
    if constexpr (sizeof(OutT) == 2) ((bf16_t*)dst)[i] = op16<OutT>::cvt_sat(v); else dst[i] = v;
  }
}

// ---- split-operand weights (fp16x2, common.h): scale = the power of two that puts max|W| in [8192, 16384), then
// row r of dst = [cols hi | cols lo] of W * scale as raw fp16.  scale_out[0] = scale, [1] = 1 / scale (both exact).
__global__ void x2_weight_scale_kernel(const float* __restrict__ src, size_t n, float* __restrict__ scale_out) {
  __shared__ float red[1024];
  float m = 0.f;
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(src[i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float mx = red[0];
    int e = 0;
    float sc = 1.f;
    if (mx > 0.f && mx < 3.0e38f) {
      frexpf(mx, &e);                            // mx = f * 2^e, f in [0.5, 1)
      sc = ldexpf(1.f, 14 - e);                  // mx * sc in [8192, 16384)
    }
    scale_out[0] = sc;
    scale_out[1] = 1.f / sc;
  }
}
__global__ void x2_weight_split_kernel(const float* __restrict__ src, int rows, int cols, const float* __restrict__ scale,
                                       bf16_t* __restrict__ dst) {
  const float sc = scale[0];
  const size_t n = (size_t)rows * cols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / cols, c = i % cols;
    const float v = src[i] * sc;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    dst[r * 2 * cols + c] = __builtin_bit_cast(bf16_t, hi);
    dst[r * 2 * cols + cols + c] = __builtin_bit_cast(bf16_t, lo);
  }
}
void launch_x2_weight_split(const float* src, int rows, int cols, void* dst, float* scale2, hipStream_t s) {
  x2_weight_scale_kernel<<<1, 1024, 0, s>>>(src, (size_t)rows * cols, scale2);
  const size_t n = (size_t)rows * cols;
  int grid = (int)((n + 255) / 256);
  if (grid > 8192) grid = 8192;
  x2_weight_split_kernel<<<grid, 256, 0, s>>>(src, rows, cols, scale2, (bf16_t*)dst);
}

// ------------------------------------------------------------------------------------------
// host launchers (called from miniroad.cpp)
// ------------------------------------------------------------------------------------------
// grid_limit > 0: at most that many workgroups (each walks rows at a stride): a THROTTLED stream for the copy that runs beside
// the latency-bound recurrence (fewer loads in flight per CU = less queueing in front of the recurrence's gather)
void launch_pack_rows(bool bf16, const float* const* rgb_ptrs, const float* const* flow_ptrs, const SlotPlan& plan,
                      int row0, int nrows, int d_rgb, int d_flow, void* X, hipStream_t s, int grid_limit, void* rowmap, bool f16, bool in16) {
  if (nrows <= 0) return;
  int grid = nrows < 65536 ? nrows : 65536;
  if (grid_limit > 0 && grid > grid_limit) grid = grid_limit;
  if (bf16 && in16)        // 16-bit features in the operand type already: the same copy kernel for bf16 and fp16
    pack_rows_kernel<bf16_t, true><<<grid, 256, 0, s>>>(rgb_ptrs, flow_ptrs, plan, row0, nrows, d_rgb, d_flow, (bf16_t*)X, (int2*)rowmap);
  else if (bf16 && f16)
    pack_rows_kernel<f16_t><<<grid, 256, 0, s>>>(rgb_ptrs, flow_ptrs, plan, row0, nrows, d_rgb, d_flow, (f16_t*)X, (int2*)rowmap);
  else if (bf16)
    pack_rows_kernel<bf16_t><<<grid, 256, 0, s>>>(rgb_ptrs, flow_ptrs, plan, row0, nrows, d_rgb, d_flow, (bf16_t*)X, (int2*)rowmap);
  else
    pack_rows_kernel<float><<<grid, 256, 0, s>>>(rgb_ptrs, flow_ptrs, plan, row0, nrows, d_rgb, d_flow, (float*)X, (int2*)rowmap);
}

// split-operand (fp16x2) feature rows: fp32 features -> [din hi | din lo] fp16 rows
void launch_pack_rows_x2(const float* const* rgb_ptrs, const float* const* flow_ptrs, const SlotPlan& plan, int row0, int nrows,
                         int d_rgb, int d_flow, void* X, hipStream_t s, int grid_limit, void* rowmap) {
  if (nrows <= 0) return;
  int grid = nrows < 65536 ? nrows : 65536;
  if (grid_limit > 0 && grid > grid_limit) grid = grid_limit;
  pack_rows_kernel<x2_t><<<grid, 256, 0, s>>>(rgb_ptrs, flow_ptrs, plan, row0, nrows, d_rgb, d_flow, (x2_t*)X, (int2*)rowmap);
}
// LayerNorm + ReLU of fp32 rows -> split rows [E hi | E lo] (inference only: no dropout, no statistics kept)
void launch_ln_relu_x2(const float* Y, const float* gamma, const float* beta, int nrows, int E, float eps, void* out, hipStream_t s,
                       const GruArm* armp) {
  if (nrows <= 0) return;
  const GruArm arm = armp ? *armp : GruArm{nullptr, 0ull, 0u, nullptr};
  int grid = (nrows + 3) / 4;
  if (grid > 16384) grid = 16384;
  if (E <= 2048) ln_relu_rows_kernel<x2_t, 4><<<grid, 256, 0, s>>>(Y, gamma, beta, nrows, E, eps, (x2_t*)out, nullptr, 0.f, 0ull, 0, 1, arm);
  else ln_relu_rows_kernel<x2_t, 8><<<grid, 256, 0, s>>>(Y, gamma, beta, nrows, E, eps, (x2_t*)out, nullptr, 0.f, 0ull, 0, 1, arm);
}

void launch_ln_relu(bool bf16, const void* Yv, const float* gamma, const float* beta, int nrows, int E, float eps, void* out,
                    float* stats, float drop_p, unsigned long long seed, int row0_abs, hipStream_t s, int relu, bool in_bf16, bool f16,
                    const GruArm* armp) {
  if (nrows <= 0) return;
  const GruArm arm = armp ? *armp : GruArm{nullptr, 0ull, 0u, nullptr};
  int grid = (nrows + 3) / 4;
  if (grid > 16384) grid = 16384;
  const float* Y = (const float*)Yv;
  if (f16) {            // fp16 operand mode (inference): fp16 or fp32 rows in, fp16 out
    if (in_bf16) {
      if (E <= 2048) ln_relu_rows_kernel<f16_t, 4, f16_t><<<grid, 256, 0, s>>>((const f16_t*)Yv, gamma, beta, nrows, E, eps, (f16_t*)out, stats, drop_p, seed, row0_abs, relu, arm);
      else ln_relu_rows_kernel<f16_t, 8, f16_t><<<grid, 256, 0, s>>>((const f16_t*)Yv, gamma, beta, nrows, E, eps, (f16_t*)out, stats, drop_p, seed, row0_abs, relu, arm);
    } else {
      if (E <= 2048) ln_relu_rows_kernel<f16_t, 4><<<grid, 256, 0, s>>>(Y, gamma, beta, nrows, E, eps, (f16_t*)out, stats, drop_p, seed, row0_abs, relu, arm);
      else ln_relu_rows_kernel<f16_t, 8><<<grid, 256, 0, s>>>(Y, gamma, beta, nrows, E, eps, (f16_t*)out, stats, drop_p, seed, row0_abs, relu, arm);
    }
    return;
  }
  if (in_bf16) {        // bf16 in, bf16 out (inference path)
    const bf16_t* Yb = (const bf16_t*)Yv;
    if (E <= 2048) ln_relu_rows_kernel<bf16_t, 4, bf16_t><<<grid, 256, 0, s>>>(Yb, gamma, beta, nrows, E, eps, (bf16_t*)out, stats, drop_p, seed, row0_abs, relu, arm);
    else ln_relu_rows_kernel<bf16_t, 8, bf16_t><<<grid, 256, 0, s>>>(Yb, gamma, beta, nrows, E, eps, (bf16_t*)out, stats, drop_p, seed, row0_abs, relu, arm);
    return;
  }
  if (E <= 2048) {
    if (bf16) ln_relu_rows_kernel<bf16_t, 4><<<grid, 256, 0, s>>>(Y, gamma, beta, nrows, E, eps, (bf16_t*)out, stats, drop_p, seed, row0_abs, relu, arm);
    else ln_relu_rows_kernel<float, 4><<<grid, 256, 0, s>>>(Y, gamma, beta, nrows, E, eps, (float*)out, stats, drop_p, seed, row0_abs, relu, arm);
  } else {
    if (bf16) ln_relu_rows_kernel<bf16_t, 8><<<grid, 256, 0, s>>>(Y, gamma, beta, nrows, E, eps, (bf16_t*)out, stats, drop_p, seed, row0_abs, relu, arm);
    else ln_relu_rows_kernel<float, 8><<<grid, 256, 0, s>>>(Y, gamma, beta, nrows, E, eps, (float*)out, stats, drop_p, seed, row0_abs, relu, arm);
  }
}

void launch_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t s) {
  int grid = (int)((n + 255) / 256);
  if (grid > 8192) grid = 8192;
  f32_to_bf16_kernel<<<grid, 256, 0, s>>>(src, (bf16_t*)dst, n);
}

void launch_pad_convert(bool bf16, const float* src, int rows_src, int cols_src, int ld_src, void* dst, int rows_dst,
                        int cols_dst, hipStream_t s, bool f16) {
  size_t n = (size_t)rows_dst * cols_dst;
  int grid = (int)((n + 255) / 256);
  if (grid > 8192) grid = 8192;
  if (bf16 && f16) pad_convert_kernel<f16_t><<<grid, 256, 0, s>>>(src, rows_src, cols_src, ld_src, (f16_t*)dst, rows_dst, cols_dst);
  else if (bf16) pad_convert_kernel<bf16_t><<<grid, 256, 0, s>>>(src, rows_src, cols_src, ld_src, (bf16_t*)dst, rows_dst, cols_dst);
  else pad_convert_kernel<float><<<grid, 256, 0, s>>>(src, rows_src, cols_src, ld_src, (float*)dst, rows_dst, cols_dst);
}

// h-state rows between the caller's clip order and the sorted (by length) order used internally
__global__ void permute_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                    const int* __restrict__ sorted_clip, int n, int width, int to_sorted) {
  const int i = blockIdx.x;              // sorted position
  if (i >= n) return;
  const int c = sorted_clip[i];
  const float* s = to_sorted ? src + (size_t)c * width : src + (size_t)i * width;
  float* d = to_sorted ? dst + (size_t)i * width : dst + (size_t)c * width;
  for (int k = threadIdx.x; k < width; k += blockDim.x) d[k] = s[k];
}
void launch_permute_rows(const float* src, float* dst, const int* sorted_clip, int n, int width, int to_sorted,
                         hipStream_t s) {
  if (n > 0) permute_rows_kernel<<<n, 256, 0, s>>>(src, dst, sorted_clip, n, width, to_sorted);
}

// out[i] = a[i] + (i < n_add ? b[i] : 0)   (folds b_hh's r,z rows into the W_ih GEMM bias)
__global__ void add_vec_kernel(const float* a, const float* b, float* out, int n, int n_add) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + (i < n_add ? b[i] : 0.f);
}
void launch_add_vec(const float* a, const float* b, float* out, int n, int n_add, hipStream_t s) {
  add_vec_kernel<<<(n + 255) / 256, 256, 0, s>>>(a, b, out, n, n_add);
}

// ---- split pass: the GI row order the recurrence wants (miniroad.cpp: w_ih_perm) ------------------------------------------------------------
// nn.GRU's W_ih rows are [r: H | z: H | n: H]; a recurrence lane owns two neighbouring units and needs their r, z and n entries of a GI row
// every step - three 4-byte loads 2 KB apart.  With the rows of W_ih (and the bias) permuted ONCE to (unit pair, gate, unit % 2) order the
// projection writes those six values next to each other and the lane takes them with one 12-byte load.  Same dot products, same K order:
// every GI value is bit-identical, only its column moved.
__global__ __launch_bounds__(256) void permute_gi_rows_kernel(const unsigned short* __restrict__ w, const float* __restrict__ bias,
                                                              unsigned short* __restrict__ wp, float* __restrict__ bp, int H, int E) {
  const int n = blockIdx.x;                         // destination row
  const int pair = n / 6, rem = n - pair * 6, gate = rem >> 1, u = pair * 2 + (rem & 1);
  const int src = gate * H + u;
  const uint4* a = (const uint4*)(w + (size_t)src * E);
  uint4* b = (uint4*)(wp + (size_t)n * E);
  for (int i = threadIdx.x; i < E / 8; i += 256) b[i] = a[i];
  if (threadIdx.x == 0) bp[n] = bias[src];
}
void launch_permute_gi_rows(const void* w, const float* bias, void* w_perm, float* bias_perm, int H, int E, hipStream_t s) {
  permute_gi_rows_kernel<<<3 * H, 256, 0, s>>>((const unsigned short*)w, bias, (unsigned short*)w_perm, bias_perm, H, E);
}
