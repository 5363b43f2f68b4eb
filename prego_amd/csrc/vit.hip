// Glue kernels of the "Transformer" (ViTEnc) path, model/transformer_models/ViT.py:117-143:
//   cat_convert   torch.cat((rgb, flow), 2) -> bf16 GEMM operand                       ViT.py:118-123
//   vit_tokens    cat cls token at the END, add learned positional embedding              ViT.py:126-129, PositionalEncoding.py:36-41
//   vit_head      pre_head_ln on token 0 (a frame token, not the cls token: ViT.py:136), mlp_head Linear   ViT.py:134-138
#include "common.h"
#include "kernels.h"

template <typename OT>
__global__ void cat_convert_kernel(const float* __restrict__ rgb, const float* __restrict__ flow, int rows, int d_rgb,
                                   int d_flow, bf16_t* __restrict__ out) {
  const int din = d_rgb + d_flow;
  for (int r = blockIdx.x; r < rows; r += gridDim.x)
    for (int c = threadIdx.x * 4; c < din; c += blockDim.x * 4) {
      const float4 v = c < d_rgb ? *(const float4*)(rgb + (size_t)r * d_rgb + c)
                                 : (flow ? *(const float4*)(flow + (size_t)r * d_flow + (c - d_rgb)) : make_float4(0, 0, 0, 0));
      uint2 o; o.x = op16<OT>::pack2_sat(v.x, v.y); o.y = op16<OT>::pack2_sat(v.z, v.w);
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

// Sliding windows over one video (the per-frame eval runner of the `Transformer` entry, prego_vit_forward_frames): window b of a
// batch ends at frame t0 + b and holds frames [t - T + 1, t], zero feature rows in front of the video (the windowing the training
// loader uses, datasets/dataset.py:53-55,96-103).  The linear encoding was computed ONCE per frame (enc [n_frames][E], bias
// included); the positional row is added per (window, position) here: token j of window b = enc[t - T + 1 + j] + pe[j] (a zero
// feature row encodes to the bias alone), token T = cls + pe[T] (ViT.py:124-129).  One wave per token row.  Outputs, all
// optional: x (fp32 residual stream [B][T+1][E]), xn (bf16 LayerNorm(ln_w, ln_b) of the row: the first block's pre-norm fused in,
// so a one-layer model never materialises x), x0 (fp32 [B][E]: token 0 of every window, the only residual row its last block reads).
template <int MAXV, typename OT>
__global__ __launch_bounds__(256) void vit_sliding_tokens_kernel(const float* __restrict__ enc, const float* __restrict__ enc_b,
                                                                 const float* __restrict__ cls, const float* __restrict__ pe, int t0,
                                                                 int B, int T, int E, float* __restrict__ x,
                                                                 const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                                 bf16_t* __restrict__ xn, float* __restrict__ x0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int N = T + 1, nv = E / 256;                      // 4-column groups per lane
  for (long long row = (long long)blockIdx.x * 4 + wave; row < (long long)B * N; row += (long long)gridDim.x * 4) {
    const int b = (int)(row / N), j = (int)(row - (long long)b * N);
    const int f = t0 + b - T + 1 + j;                     // frame of token j (j < T)
    const float* src = j == T ? cls : (f >= 0 ? enc + (size_t)f * E : enc_b);
    const float* per = pe + (size_t)j * E;
    float v[MAXV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (i < nv) {
        const int c = (i * 64 + lane) * 4;
        const float4 a = *(const float4*)(src + c), p = *(const float4*)(per + c);
        v[i][0] = a.x + p.x; v[i][1] = a.y + p.y; v[i][2] = a.z + p.z; v[i][3] = a.w + p.w;
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        if (x) *(float4*)(x + (size_t)row * E + c) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
        if (x0 && j == 0) *(float4*)(x0 + (size_t)b * E + c) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
      }
    if (xn == nullptr) continue;
    const float mu = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (i < nv) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float d = v[i][k] - mu; q += d * d; }
      }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + 1e-5f);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (i < nv) {
        const int c = (i * 64 + lane) * 4;
        const float4 g = *(const float4*)(ln_w + c), bb = *(const float4*)(ln_b + c);
        uint2 o;
        o.x = op16<OT>::pack2_sat((v[i][0] - mu) * rstd * g.x + bb.x, (v[i][1] - mu) * rstd * g.y + bb.y);
        o.y = op16<OT>::pack2_sat((v[i][2] - mu) * rstd * g.z + bb.z, (v[i][3] - mu) * rstd * g.w + bb.w);
        *(uint2*)(xn + (size_t)row * E + c) = o;
      }
  }
}

__global__ __launch_bounds__(256) void vit_head_kernel(const float* __restrict__ x, int N, int E, const float* __restrict__ lnw,
                                                       const float* __restrict__ lnb, const float* __restrict__ hw,
                                                       const float* __restrict__ hb, int C, float* __restrict__ out,
                                                       int* __restrict__ argmax /*nullable: np.argmax of the C logits, first max wins*/) {
  extern __shared__ float sx[];                    // [E] normalised token + 8 reduction slots + C logits (argmax)
  float* red = sx + E;
  float* slog = red + 8;
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
      for (int j = 0; j < 4; ++j) if (k0 + j < C) { const float lg = a[j] + hb[k0 + j]; out[(size_t)b * C + k0 + j] = lg; slog[k0 + j] = lg; }
    }
  }
  if (argmax == nullptr) return;                   // block-uniform
  __syncthreads();
  if (wave == 0) {
    float mx = -INFINITY; int mi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) if (slog[c] > mx) { mx = slog[c]; mi = c; }      // ascending c inside a lane: first max wins
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float omx = __shfl_xor(mx, o, 64); const int omi = __shfl_xor(mi, o, 64);
      if (omx > mx || (omx == mx && omi < mi)) { mx = omx; mi = omi; }
    }
    if (lane == 0) argmax[b] = mi;
  }
}

void launch_cat_convert(const float* rgb, const float* flow, int rows, int d_rgb, int d_flow, void* out_bf16, hipStream_t s, bool f16) {
  if (rows <= 0) return;
  if (f16) cat_convert_kernel<f16_t><<<rows < 16384 ? rows : 16384, 256, 0, s>>>(rgb, flow, rows, d_rgb, d_flow, (bf16_t*)out_bf16);
  else cat_convert_kernel<bf16_t><<<rows < 16384 ? rows : 16384, 256, 0, s>>>(rgb, flow, rows, d_rgb, d_flow, (bf16_t*)out_bf16);
}
void launch_vit_tokens(const float* enc, const float* cls, const float* pe, int B, int T, int E, float* x, hipStream_t s,
                       unsigned drop_thresh, float drop_scale, unsigned long long drop_seed) {
  vit_tokens_kernel<<<B * (T + 1), 256, 0, s>>>(enc, cls, pe, B, T, E, x, drop_thresh, drop_scale, drop_seed);
}
void launch_vit_cls_rows(const float* cls, const float* pe, int B, int T, int E, float* x, hipStream_t s) {
  vit_cls_rows_kernel<<<B, 256, 0, s>>>(cls, pe, T, E, x);
}
void launch_vit_head(const float* x, int B, int N, int E, const float* lnw, const float* lnb, const float* hw,
                     const float* hb, int C, float* out, hipStream_t s, int* argmax) {
  vit_head_kernel<<<B, 256, (E + 8 + C) * sizeof(float), s>>>(x, N, E, lnw, lnb, hw, hb, C, out, argmax);
}
void launch_vit_sliding_tokens(const float* enc, const float* enc_b, const float* cls, const float* pe, int t0, int B, int T, int E,
                               float* x, const float* ln_w, const float* ln_b, void* xn, float* x0, hipStream_t s, bool f16) {
  long long rows = (long long)B * (T + 1);
  int grid = (int)((rows + 3) / 4 < 32768 ? (rows + 3) / 4 : 32768);
#define VST(MV, OT) vit_sliding_tokens_kernel<MV, OT><<<grid, 256, 0, s>>>(enc, enc_b, cls, pe, t0, B, T, E, x, ln_w, ln_b, (bf16_t*)xn, x0)
  if (E <= 2048) { if (f16) VST(8, f16_t); else VST(8, bf16_t); }
  else { if (f16) VST(16, f16_t); else VST(16, bf16_t); }
#undef VST
}
