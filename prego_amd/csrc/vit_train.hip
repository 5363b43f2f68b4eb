// Training glue of the "Transformer" (ViTEnc) path: the element-wise and small-reduction halves of the backward pass of
// model/transformer_models/ViT.py:117-143 and Transformer.py:35-47 (the matrix products run on the NT MFMA GEMMs, the
// attention backward in attention_bwd.hip, LayerNorm backward in train.hip).
//   gelu_bwd          du = df * gelu'(u), exact-erf GELU (nn.GELU default, Transformer.py:40)
//   vit_head_bwd      mlp_head Linear + pre_head_ln on TOKEN 0 only (ViT.py:134-138): dx[b,0,:], d(ln), d(head)
//   vit_tokens_bwd    learned positional table, cls token appended at the END (ViT.py:126-129), rows of the encoding GEMM
// Every reduction over the batch runs in a fixed order (bit-reproducible gradients).
#include "common.h"
#include "kernels.h"

__global__ void gelu_bwd_kernel(const float* __restrict__ df, const float* __restrict__ u, size_t n, float* __restrict__ du,
                                bf16_t* __restrict__ du_b, unsigned drop_thresh, float drop_scale, unsigned long long drop_seed) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
    const float4 g = *(const float4*)(df + i), x = *(const float4*)(u + i);
    const float gg[4] = {g.x, g.y, g.z, g.w}, xx[4] = {x.x, x.y, x.z, x.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // d/dx [x Phi(x)] = Phi(x) + x phi(x)
      const float cdf = 0.5f * (1.0f + erff(xx[k] * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * xx[k] * xx[k]);
      // the FFN's first Dropout sits between GELU and the second Linear (Transformer.py:41): same mask as the forward
      const float gk = drop_thresh ? (dropout_keep_(drop_seed, i + k, drop_thresh) ? gg[k] * drop_scale : 0.f) : gg[k];
      o[k] = gk * (cdf + xx[k] * pdf);
    }
    *(float4*)(du + i) = make_float4(o[0], o[1], o[2], o[3]);
    uint2 w; w.x = pack_bf16x2(o[0], o[1]); w.y = pack_bf16x2(o[2], o[3]);
    *(uint2*)(du_b + i) = w;
  }
}
void launch_gelu_bwd(const float* df, const float* u, size_t n, float* du, void* du_bf16, hipStream_t s, unsigned drop_thresh,
                     float drop_scale, unsigned long long drop_seed) {
  if (n == 0) return;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  gelu_bwd_kernel<<<(int)blocks, 256, 0, s>>>(df, u, n, du, (bf16_t*)du_bf16, drop_thresh, drop_scale, drop_seed);
}

// gradient entering a dropped-out branch: out = src * mask * scale, as fp32 (bias column sums, wgrad operand) and bf16 (dgrad operand)
__global__ void mask_convert_kernel(const float* __restrict__ src, size_t n, float* __restrict__ out, bf16_t* __restrict__ out_b,
                                    unsigned drop_thresh, float drop_scale, unsigned long long drop_seed, unsigned drop2_thresh,
                                    float drop2_scale, unsigned long long drop2_seed) {
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {
    const float4 v = *(const float4*)(src + i);
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (drop_thresh) o[k] = dropout_keep_(drop_seed, i + k, drop_thresh) ? o[k] * drop_scale : 0.f;
      if (drop2_thresh) o[k] = dropout_keep_(drop2_seed, i + k, drop2_thresh) ? o[k] * drop2_scale : 0.f;
    }
    if (out) *(float4*)(out + i) = make_float4(o[0], o[1], o[2], o[3]);
    uint2 w; w.x = pack_bf16x2(o[0], o[1]); w.y = pack_bf16x2(o[2], o[3]);
    *(uint2*)(out_b + i) = w;
  }
}
void launch_mask_convert(const float* src, size_t n, float* out_f32, void* out_bf16, unsigned drop_thresh, float drop_scale,
                         unsigned long long drop_seed, hipStream_t s, unsigned drop2_thresh, float drop2_scale,
                         unsigned long long drop2_seed) {
  if (n == 0) return;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  mask_convert_kernel<<<(int)blocks, 256, 0, s>>>(src, n, out_f32, (bf16_t*)out_bf16, drop_thresh, drop_scale, drop_seed, drop2_thresh,
                                                  drop2_scale, drop2_seed);
}

// ---- head: logits = LN(x[b,0,:]) Wh^T + bh  (ViT.py:134-138) -----------------------------------------------------
// stage 1 (one workgroup per window b): xhat, y = LN(x0), dy = dlogits[b] . Wh, LayerNorm backward -> dx[b,0,:];
//   scratch[0][b] = xhat, scratch[1][b] = y (the head's input), scratch[2][b] = dy
__global__ __launch_bounds__(256) void vit_head_bwd1_kernel(const float* __restrict__ x, const float* __restrict__ dlogits, int N,
                                                            int E, int C, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                            const float* __restrict__ hw, float* __restrict__ dx, float* __restrict__ scratch,
                                                            int B) {
  __shared__ float red[8];
  __shared__ float sdl[128];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (size_t)b * N * E;
  for (int c = tid; c < C; c += 256) sdl[c] = dlogits[(size_t)b * C + c];
  float s = 0.f;
  for (int c = tid; c < E; c += 256) s += xr[c];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mu = (red[0] + red[1] + red[2] + red[3]) / (float)E;
  float q = 0.f;
  for (int c = tid; c < E; c += 256) { const float d = xr[c] - mu; q += d * d; }
  q = wave_sum(q);
  if (lane == 0) red[4 + wave] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[4] + red[5] + red[6] + red[7]) / (float)E + 1e-5f);
  __syncthreads();
  float* xh = scratch + (size_t)b * E;
  float* yo = scratch + ((size_t)B + b) * E;
  float* dyo = scratch + ((size_t)2 * B + b) * E;
  float s1 = 0.f, s2 = 0.f;
  for (int c = tid; c < E; c += 256) {
    const float xv = (xr[c] - mu) * rstd;
    float dy = 0.f;
    for (int k = 0; k < C; ++k) dy += sdl[k] * hw[(size_t)k * E + c];
    xh[c] = xv; yo[c] = xv * lnw[c] + lnb[c]; dyo[c] = dy;
    const float dxh = dy * lnw[c];
    s1 += dxh; s2 += dxh * xv;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { red[wave] = s1; red[4 + wave] = s2; }
  __syncthreads();
  const float m1 = (red[0] + red[1] + red[2] + red[3]) / (float)E, m2 = (red[4] + red[5] + red[6] + red[7]) / (float)E;
  float* dxr = dx + (size_t)b * N * E;
  for (int c = tid; c < E; c += 256) dxr[c] = rstd * (dyo[c] * lnw[c] - m1 - xh[c] * m2);
}
// stage 2: sums over the batch in window order: d ln weight/bias [E], d head weight [C][E], d head bias [C]
__global__ void vit_head_bwd2_kernel(const float* __restrict__ scratch, const float* __restrict__ dlogits, int B, int E, int C,
                                     float* __restrict__ g_lnw, float* __restrict__ g_lnb, float* __restrict__ g_hw,
                                     float* __restrict__ g_hb) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;                        // k < C: row k of the head weight; k == C: the LayerNorm parameters
  if (c >= E) return;
  if (k == C) {
    float a = 0.f, d = 0.f;
    for (int b = 0; b < B; ++b) {
      const float dy = scratch[((size_t)2 * B + b) * E + c];
      a += dy * scratch[(size_t)b * E + c];
      d += dy;
    }
    g_lnw[c] = a; g_lnb[c] = d;
    if (c < C) {
      float hb = 0.f;
      for (int b = 0; b < B; ++b) hb += dlogits[(size_t)b * C + c];
      g_hb[c] = hb;
    }
  } else {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += dlogits[(size_t)b * C + k] * scratch[((size_t)B + b) * E + c];
    g_hw[(size_t)k * E + c] = a;
  }
}
// dx must be zero-filled by the caller (only token 0 of every window receives a gradient from the head)
void launch_vit_head_bwd(const float* x, const float* dlogits, int B, int N, int E, int C, const float* lnw, const float* lnb,
                         const float* hw, float* dx, float* scratch, float* g_lnw, float* g_lnb, float* g_hw, float* g_hb,
                         hipStream_t s) {
  vit_head_bwd1_kernel<<<B, 256, 0, s>>>(x, dlogits, N, E, C, lnw, lnb, hw, dx, scratch, B);
  dim3 g2((E + 255) / 256, C + 1);
  vit_head_bwd2_kernel<<<g2, 256, 0, s>>>(scratch, dlogits, B, E, C, g_lnw, g_lnb, g_hw, g_hb);
}

// ---- tokens: x[b,n,:] = (n < T ? enc[b,n,:] : cls) + pe[n,:]   (ViT.py:126-129, PositionalEncoding.py:36-41) ------
// grid (N, E/256): d pe[n] = sum_b dx[b,n]; d cls = sum_b dx[b,T]; denc[b*T + n] = dx[b,n] for n < T (compact rows)
__global__ void vit_tokens_bwd_kernel(const float* __restrict__ dx, int B, int T, int E, float* __restrict__ denc,
                                      float* __restrict__ g_pe, float* __restrict__ g_cls, unsigned drop_thresh, float drop_scale,
                                      unsigned long long drop_seed) {
  const int n = blockIdx.x, c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c >= E) return;
  const int N = T + 1;
  float a = 0.f;
  for (int b = 0; b < B; ++b) {
    float v = dx[((size_t)b * N + n) * E + c];
    if (drop_thresh) v = dropout_keep_(drop_seed, ((size_t)b * N + n) * E + c, drop_thresh) ? v * drop_scale : 0.f;   // pe_dropout
    a += v;
    if (n < T) denc[((size_t)b * T + n) * E + c] = v;
  }
  g_pe[(size_t)n * E + c] = a;
  if (n == T) g_cls[c] = a;
}
void launch_vit_tokens_bwd(const float* dx, int B, int T, int E, float* denc, float* g_pe, float* g_cls, hipStream_t s,
                           unsigned drop_thresh, float drop_scale, unsigned long long drop_seed) {
  dim3 g(T + 1, (E + 255) / 256);
  vit_tokens_bwd_kernel<<<g, 256, 0, s>>>(dx, B, T, E, denc, g_pe, g_cls, drop_thresh, drop_scale, drop_seed);
}
