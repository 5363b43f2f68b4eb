// fp32-operand kernels of the Transformer path (parity mode: PREGO_F32 handles of prego_vit_* / prego_attention_layer_*).
// The projections run on the exact-fp32 MFMA GEMM (gemm.hip: gemm_f32_nt); what is here is the rest:
//   attention_f32_kernel   softmax(scale * q k^T [+ causal mask]) v over fp32 rows, one query per wave, online softmax
//                          (Attention.py:30-38; attn.py:41-52 with TriangularCausalMask attn.py:10-18)
//   gelu_f32_kernel        exact-erf GELU in place (Transformer.py:40)
//   add_rows_kernel        x += y (the residual adds of Transformer.py:60-77)
//   cat_rows_f32_kernel    [rgb | flow] -> one fp32 row (ViT.py:122-123; a missing flow half reads as zeros)
// Throughput is not the point of this mode (it is what the 1e-3 fp32 tolerance of the north star is checked on); every element
// is still a coalesced access and the attention keeps K / V rows in L2.
#include "common.h"
#include "kernels.h"

// qkv [B*N, ld] fp32: q at column q_off + h*DH, k at k_off + h*DH, v at v_off + h*DH; out [B*N, heads*DH].
// One wave per (batch, head, query); lane l holds elements l, l + 64, ... of the head dimension.
template <int DH>
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                            float* __restrict__ out, int B, int N, int Nq, int heads, int causal,
                                                            float scale) {
  constexpr int R = DH / 64;
  const int lane = threadIdx.x & 63;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long total = (long long)B * heads * Nq;
  if (wid >= total) return;
  const int i = (int)(wid % Nq);
  const int hd = (int)((wid / Nq) % heads);
  const int b = (int)(wid / ((long long)Nq * heads));
  const float* qp = qkv + ((size_t)b * N + i) * ld + q_off + hd * DH;
  float q[R], o[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { q[r] = qp[lane + 64 * r] * scale; o[r] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const int jend = causal ? i + 1 : N;
  for (int j = 0; j < jend; ++j) {
    const float* kp = qkv + ((size_t)b * N + j) * ld + k_off + hd * DH;
    const float* vp = qkv + ((size_t)b * N + j) * ld + v_off + hd * DH;
    float sdot = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) sdot += q[r] * kp[lane + 64 * r];
    sdot = wave_sum(sdot);
    const float mn = fmaxf(m, sdot);
    const float a = expf(m - mn), p = expf(sdot - mn);
    l = l * a + p;
#pragma unroll
    for (int r = 0; r < R; ++r) o[r] = o[r] * a + p * vp[lane + 64 * r];
    m = mn;
  }
  float* op = out + ((size_t)b * Nq + i) * (heads * DH) + hd * DH;
  const float inv = 1.0f / l;
#pragma unroll
  for (int r = 0; r < R; ++r) op[lane + 64 * r] = o[r] * inv;
}

int launch_attention_f32(const float* qkv, int ld, int q_off, int k_off, int v_off, float* out, int B, int N, int Nq, int heads,
                         int dh, int causal, float scale, hipStream_t s) {
  const long long waves = (long long)B * heads * Nq;
  if (waves <= 0) return 0;
  const unsigned grid = (unsigned)((waves + 3) / 4);
  if (dh == 64) attention_f32_kernel<64><<<grid, 256, 0, s>>>(qkv, ld, q_off, k_off, v_off, out, B, N, Nq, heads, causal, scale);
  else if (dh == 128) attention_f32_kernel<128><<<grid, 256, 0, s>>>(qkv, ld, q_off, k_off, v_off, out, B, N, Nq, heads, causal, scale);
  else if (dh == 256) attention_f32_kernel<256><<<grid, 256, 0, s>>>(qkv, ld, q_off, k_off, v_off, out, B, N, Nq, heads, causal, scale);
  else return -1;
  return 0;
}

__global__ void gelu_f32_kernel(float* __restrict__ u, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) u[i] = gelu_erf_(u[i]);
}
void launch_gelu_f32(float* u, size_t n, hipStream_t s) {
  if (!n) return;
  const size_t g = (n + 255) / 256;
  gelu_f32_kernel<<<(unsigned)(g > 16384 ? 16384 : g), 256, 0, s>>>(u, n);
}

__global__ void add_rows_kernel(float* __restrict__ x, const float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] += y[i];
}
void launch_add_rows(float* x, const float* y, size_t n, hipStream_t s) {
  if (!n) return;
  const size_t g = (n + 255) / 256;
  add_rows_kernel<<<(unsigned)(g > 16384 ? 16384 : g), 256, 0, s>>>(x, y, n);
}

__global__ void cat_rows_f32_kernel(const float* __restrict__ rgb, const float* __restrict__ flow, int rows, int d_rgb, int d_flow,
                                    float* __restrict__ out) {
  const int din = d_rgb + d_flow;
  const size_t n = (size_t)rows * din;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / din;
    const int c = (int)(i - r * din);
    float v = 0.f;
    if (c < d_rgb) { if (rgb) v = rgb[r * d_rgb + c]; }
    else if (flow) v = flow[r * d_flow + (c - d_rgb)];
    out[i] = v;
  }
}
void launch_cat_rows_f32(const float* rgb, const float* flow, int rows, int d_rgb, int d_flow, float* out, hipStream_t s) {
  const size_t n = (size_t)rows * (d_rgb + d_flow);
  if (!n) return;
  const size_t g = (n + 255) / 256;
  cat_rows_f32_kernel<<<(unsigned)(g > 16384 ? 16384 : g), 256, 0, s>>>(rgb, flow, rows, d_rgb, d_flow, out);
}
