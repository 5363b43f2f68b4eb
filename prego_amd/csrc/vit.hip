// Glue kernels of the "Transformer" (ViTEnc) path, model/transformer_models/ViT.py:117-143:
//   cat_convert   torch.cat((rgb, flow), 2) -> bf16 GEMM operand                       ViT.py:118-123
//   vit_tokens    cat cls token at the END, add learned positional embedding              ViT.py:126-129, PositionalEncoding.py:36-41
//   vit_head      pre_head_ln on token 0 (a frame token, not the cls token: ViT.py:136), mlp_head Linear   ViT.py:134-138
#include "common.h"
#include "kernels.h"

__global__ void cat_convert_kernel(const float* __restrict__ rgb, const float* __restrict__ flow, int rows, int d_rgb,
                                   int d_flow, bf16_t* __restrict__ out) {
  const int din = d_rgb + d_flow;
  for (int r = blockIdx.x; r < rows; r += gridDim.x)
    for (int c = threadIdx.x * 4; c < din; c += blockDim.x * 4) {
      const float4 v = c < d_rgb ? *(const float4*)(rgb + (size_t)r * d_rgb + c)
                                 : (flow ? *(const float4*)(flow + (size_t)r * d_flow + (c - d_rgb)) : make_float4(0, 0, 0, 0));
      uint2 o; o.x = pack_bf16x2(v.x, v.y); o.y = pack_bf16x2(v.z, v.w);
      *(uint2*)(out + (size_t)r * din + c) = o;
    }
}

__global__ void vit_tokens_kernel(const float* __restrict__ enc, const float* __restrict__ cls, const float* __restrict__ pe,
                                  int B, int T, int E, float* __restrict__ x, unsigned drop_thresh, float drop_scale,
                                  unsigned long long drop_seed) {
  const int row = blockIdx.x;                       // b * (T+1) + n
  const int b = row / (T + 1), n = row % (T + 1);
  const float* src = n < T ? enc + ((size_t)b * T + n) * E : cls;
  for (int c = threadIdx.x; c < E; c += blockDim.x) {
    float v = src[c] + pe[(size_t)n * E + c];
    // pe_dropout (ViT.py:130), training only
    if (drop_thresh) v = dropout_keep_(drop_seed, (size_t)row * E + c, drop_thresh) ? v * drop_scale : 0.f;
    x[(size_t)row * E + c] = v;
  }
}

// the cls row of every window: x[b, T] = cls + pe[T] (ViT.py:126-129; the frame rows come out of the encoding GEMM's epilogue)
__global__ void vit_cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pe, int T, int E, float* __restrict__ x) {
  float* xr = x + ((size_t)blockIdx.x * (T + 1) + T) * E;
  for (int c = threadIdx.x; c < E; c += blockDim.x) xr[c] = cls[c] + pe[(size_t)T * E + c];
}

__global__ __launch_bounds__(256) void vit_head_kernel(const float* __restrict__ x, int N, int E, const float* __restrict__ lnw,
                                                       const float* __restrict__ lnb, const float* __restrict__ hw,
                                                       const float* __restrict__ hb, int C, float* __restrict__ out) {
  extern __shared__ float sx[];                    // [E] normalised token + 8 reduction slots
  float* red = sx + E;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (size_t)b * N * E;         // token 0 of clip b
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
  for (int c = tid; c < E; c += 256) sx[c] = (xr[c] - mu) * rstd * lnw[c] + lnb[c];
  __syncthreads();
  // four classes per wave and pass: four independent dot products keep the loads and the shuffle reductions in flight together
  // (one class at a time was a chain of 22 dependent L2 round trips + reductions per wave: 250 us for 256 windows)
  for (int k0 = wave * 4; k0 < C; k0 += 16) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = lane * 4; c < E; c += 256) {
      const float4 xv = *(const float4*)(sx + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + j < C ? k0 + j : C - 1;
        const float4 wv = *(const float4*)(hw + (size_t)k * E + c);
        a[j] += (xv.x * wv.x + xv.y * wv.y) + (xv.z * wv.z + xv.w * wv.w);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = wave_sum(a[j]);
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (k0 + j < C) out[(size_t)b * C + k0 + j] = a[j] + hb[k0 + j];
    }
  }
}

void launch_cat_convert(const float* rgb, const float* flow, int rows, int d_rgb, int d_flow, void* out_bf16, hipStream_t s) {
  if (rows <= 0) return;
  cat_convert_kernel<<<rows < 16384 ? rows : 16384, 256, 0, s>>>(rgb, flow, rows, d_rgb, d_flow, (bf16_t*)out_bf16);
}
void launch_vit_tokens(const float* enc, const float* cls, const float* pe, int B, int T, int E, float* x, hipStream_t s,
                       unsigned drop_thresh, float drop_scale, unsigned long long drop_seed) {
  vit_tokens_kernel<<<B * (T + 1), 256, 0, s>>>(enc, cls, pe, B, T, E, x, drop_thresh, drop_scale, drop_seed);
}
void launch_vit_cls_rows(const float* cls, const float* pe, int B, int T, int E, float* x, hipStream_t s) {
  vit_cls_rows_kernel<<<B, 256, 0, s>>>(cls, pe, T, E, x);
}
void launch_vit_head(const float* x, int B, int N, int E, const float* lnw, const float* lnb, const float* hw,
                     const float* hb, int C, float* out, hipStream_t s) {
  vit_head_kernel<<<B, 256, (E + 8) * sizeof(float), s>>>(x, N, E, lnw, lnb, hw, hb, C, out);
}
