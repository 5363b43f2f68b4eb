// Training-path kernels of MiniROAD (forward keeps activations; this file is the backward side):
//   oad_loss_kernel          OadLoss.end_loss/mlce_loss      criterions/loss.py:15-34  (+ dL/dlogits)
//   gather_dlogits_kernel    caller [clip][t][C] -> packed time-major rows (operand dtype, class-padded)
//   transpose_convert_kernel [M,N] -> [N, Mpad] (the reduction dim of every weight gradient is the row index,
//                            so wgrads run on the same NT MFMA GEMM after one transpose of each operand)
//   colsum_*                 bias gradients, deterministic two-stage column sums
//   gru_bwd_step_kernel      one reverse time step of BPTT through nn.GRU's cell (rnn.py:61), elementwise part
//   ln_relu_bwd_rows_kernel  Dropout/ReLU/LayerNorm backward (rnn.py:41-43) + per-block partials of dgamma/dbeta
//   relu_mask_kernel         d relu(h) (rnn.py:62)
// The matrix products of the backward pass (dgrad/wgrad, and the per-step dh_{t-1} += dgh . W_hh) reuse
// gemm_bf16_nt / gemm_f32_nt.  Batches here are train.py-sized (16 windows x 128 frames = 2048 rows), so this
// side is written for correctness and determinism first; the eval path is where the frames/s are.
#include "common.h"
#include "kernels.h"

// ---- loss -------------------------------------------------------------------------------------
// one workgroup of 16 waves, wave v takes clips v, v+16, ... (the per-clip work - zeroing dlogits of the T-1 unused frames -
// is what costs time); the per-clip terms are summed by one thread IN CLIP ORDER, so the loss is bit-identical to a serial
// loop over the clips.  loss = mean_b sum_k -(y/||y||)_k log_softmax(l)_k
#define OAD_WAVES 16
#define OAD_MAX_LDS_CLIPS 4096
__global__ __launch_bounds__(64 * OAD_WAVES) void oad_loss_kernel(
    const float* const* __restrict__ logit_ptrs, const float* const* __restrict__ target_ptrs,
    const int* __restrict__ lens, int n_clips, int C, float* __restrict__ loss_out,
    float* const* __restrict__ dlogit_ptrs /*nullable*/, float grad_scale, float denom /* n_clips: reduction='mean'; 1: 'sum' (loss.py:30-33) */) {
  __shared__ float s_per[OAD_MAX_LDS_CLIPS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool serial = n_clips > OAD_MAX_LDS_CLIPS;       // more clips than the LDS table holds: one wave does all, as before
  float total = 0.f;
  for (int b = serial ? 0 : wave; b < n_clips && (!serial || wave == 0); b += serial ? 1 : OAD_WAVES) {
    const int T = lens[b];
    const float* lg = logit_ptrs[b] + (size_t)(T - 1) * C;
    const float* tg = target_ptrs[b] + (size_t)(T - 1) * C;
    float l[2], y[2];
    float mx = -INFINITY, ss = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = lane + 64 * i;
      l[i] = c < C ? lg[c] : -INFINITY;
      y[i] = c < C ? tg[c] : 0.f;
      mx = fmaxf(mx, l[i]);
      ss += y[i] * y[i];
    }
    mx = wave_max(mx);
    const float nrm = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);          // F.normalize eps
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) se += (lane + 64 * i < C) ? expf(l[i] - mx) : 0.f;
    se = wave_sum(se);
    const float lse = mx + logf(se);
    float per = 0.f, ysum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      y[i] /= nrm;
      if (lane + 64 * i < C) { per += -y[i] * (l[i] - lse); ysum += y[i]; }
    }
    per = wave_sum(per);
    ysum = wave_sum(ysum);
    if (serial) total += per;
    else if (lane == 0) s_per[b] = per;
    if (dlogit_ptrs) {
      float* dl = dlogit_ptrs[b];
      // zero everything but the last frame (loss.py:18 uses logits[:, -1, :] only)
      for (size_t i = lane; i < (size_t)(T - 1) * C; i += 64) dl[i] = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = lane + 64 * i;
        if (c < C) dl[(size_t)(T - 1) * C + c] = (expf(l[i] - lse) * ysum - y[i]) * grad_scale / denom;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!serial) for (int b = 0; b < n_clips; ++b) total += s_per[b];
    loss_out[0] = total / denom;
  }
}

// ---- layout helpers -----------------------------------------------------------------------------
template <typename OutT>
__global__ void gather_dlogits_kernel(const float* const* __restrict__ dl_ptrs, const int* __restrict__ rowoff,
                                      const int* __restrict__ sorted_clip, int t_max, int nrows, int C, int Cpad,
                                      OutT* __restrict__ out) {
  const int r = blockIdx.x;
  if (r >= nrows) return;
  const int t = plan_time_of_row(rowoff, t_max, r);
  const int clip = sorted_clip[r - rowoff[t]];
  const float* src = dl_ptrs[clip] + (size_t)t * C;
  for (int c = threadIdx.x; c < Cpad; c += blockDim.x) {
    const float v = c < C ? src[c] : 0.f;
    if constexpr (sizeof(OutT) == 2) out[(size_t)r * Cpad + c] = f2bf(v); else out[(size_t)r * Cpad + c] = v;
  }
}

// dst[n][m] = src[m][n] for m < M, zero for M <= m < Mpad.  32x32 tiles through LDS.
template <typename InT, typename OutT>
__global__ void transpose_convert_kernel(const InT* __restrict__ src, int M, int N, int ld_src, OutT* __restrict__ dst,
                                         int Mpad) {
  __shared__ float tile[32][33];
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 256 threads: 8 rows per pass
  for (int i = ty; i < 32; i += 8) {
    const int m = m0 + i, n = n0 + tx;
    float v = 0.f;
    if (m < M && n < N) {
      if constexpr (sizeof(InT) == 2) v = bf2f(src[(size_t)m * ld_src + n]); else v = src[(size_t)m * ld_src + n];
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, m = m0 + tx;
    if (n < N && m < Mpad) {
      const float v = tile[tx][i];
      if constexpr (sizeof(OutT) == 2) dst[(size_t)n * Mpad + m] = f2bf(v); else dst[(size_t)n * Mpad + m] = v;
    }
  }
}

// column sums, stage 1: block b sums rows [b*RB, (b+1)*RB) -> part[b][N]; stage 2 sums the partials in order.
#define COLSUM_RB 64
template <typename InT>
__global__ void colsum_stage1_kernel(const InT* __restrict__ src, int M, int N, float* __restrict__ part) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int m0 = blockIdx.y * COLSUM_RB;
  float s = 0.f;
  for (int m = m0; m < m0 + COLSUM_RB && m < M; ++m) {
    if constexpr (sizeof(InT) == 2) s += bf2f(src[(size_t)m * N + n]); else s += src[(size_t)m * N + n];
  }
  part[(size_t)blockIdx.y * N + n] = s;
}
// 64 columns x 16 row slices per workgroup: slice y sums its contiguous share of the partials in order, the 16 slice sums are
// added in slice order (fixed order = bit-reproducible; one thread per column walking all nb partials took 30 us at nb = 512)
__global__ __launch_bounds__(1024) void colsum_stage2_kernel(const float* __restrict__ part, int nb, int N, float* __restrict__ out) {
  __shared__ float s_sl[16][64];
  const int cx = threadIdx.x & 63, sy = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cx;
  const int per = (nb + 15) / 16;
  float s = 0.f;
  if (n < N) {
    const int b1 = (sy + 1) * per < nb ? (sy + 1) * per : nb;
    for (int b = sy * per; b < b1; ++b) s += part[(size_t)b * N + n];
  }
  s_sl[sy][cx] = s;
  __syncthreads();
  if (sy == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int y = 0; y < 16; ++y) t += s_sl[y][cx];
    out[n] = t;
  }
}

// ---- GRU backward, elementwise part of one reverse step ---------------------------------------------
// For the clips alive at time t (sorted prefix) and every hidden unit j:
//   dh      = dHout[row] + (clip continues at t+1 ? carry_z[b] + dhpart[b] : 0)
//   dn = dh (1-z); dz = dh (hprev - n); dpre_n = dn (1-n^2); dpre_r = dpre_n * ghn * r (1-r); dpre_z = dz z (1-z)
//   dGI[row]  = [dpre_r | dpre_z | dpre_n]                (gradient of the input projection, fp32 + operand copy)
//   dGH[row]  = [dpre_r | dpre_z | dpre_n * r]            (gradient of W_hh h + b_hh, operand copy + fp32)
//   carry_z[b] = dh * z                                   (the direct path to h_{t-1})
// hprev = h_{t-1} = Hraw[row(t-1, b)] or 0 at t = 0.
template <typename OpT>
__global__ void gru_bwd_step_kernel(int t, int na, int na_next, int row_t, int row_tm1, int H,
                                    const float* __restrict__ dHout, const float* __restrict__ carry_z_in,
                                    const float* __restrict__ dhpart, const float* __restrict__ R,
                                    const float* __restrict__ Z, const float* __restrict__ Nn,
                                    const float* __restrict__ GHN, const float* __restrict__ Hraw,
                                    float* __restrict__ carry_z_out, float* __restrict__ dGI, float* __restrict__ dGH,
                                    OpT* __restrict__ dGIop, OpT* __restrict__ dGHop) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= na * H) return;
  const int b = idx / H, j = idx % H;
  const size_t e = (size_t)(row_t + b) * H + j;
  float dh = dHout[e];
  if (b < na_next) dh += carry_z_in[(size_t)b * H + j] + dhpart[(size_t)b * H + j];
  const float r = R[e], z = Z[e], n = Nn[e], ghn = GHN[e];
  const float hprev = t > 0 ? Hraw[(size_t)(row_tm1 + b) * H + j] : 0.f;
  const float dn = dh * (1.f - z);
  const float dz = dh * (hprev - n);
  const float dpn = dn * (1.f - n * n);
  const float dpr = dpn * ghn * r * (1.f - r);
  const float dpz = dz * z * (1.f - z);
  carry_z_out[(size_t)b * H + j] = dh * z;
  const size_t g = (size_t)(row_t + b) * 3 * H + j;
  dGI[g] = dpr; dGI[g + H] = dpz; dGI[g + 2 * H] = dpn;
  dGH[g] = dpr; dGH[g + H] = dpz; dGH[g + 2 * H] = dpn * r;
  if constexpr (sizeof(OpT) == 2) {
    dGIop[g] = f2bf(dpr); dGIop[g + H] = f2bf(dpz); dGIop[g + 2 * H] = f2bf(dpn);
    dGHop[g] = f2bf(dpr); dGHop[g + H] = f2bf(dpz); dGHop[g + 2 * H] = f2bf(dpn * r);
  } else {
    dGIop[g] = dpr; dGIop[g + H] = dpz; dGIop[g + 2 * H] = dpn;
    dGHop[g] = dpr; dGHop[g + H] = dpz; dGHop[g + 2 * H] = dpn * r;
  }
}

// dHout = dHrelu * (h > 0)
__global__ void relu_mask_kernel(const float* __restrict__ dHrelu, const float* __restrict__ Hraw, size_t n,
                                 float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = Hraw[i] > 0.f ? dHrelu[i] : 0.f;
}

// Hprev operand for dW_hh: row(t, b) -> h_{t-1, b} (zero at t = 0)
template <typename OpT>
__global__ void build_hprev_kernel(const float* __restrict__ Hraw, const int* __restrict__ rowoff,
                                   int t_max, int nrows, int H, OpT* __restrict__ out) {
  const int r = blockIdx.x;
  if (r >= nrows) return;
  const int t = plan_time_of_row(rowoff, t_max, r);
  const int b = r - rowoff[t];
  for (int j = threadIdx.x; j < H; j += blockDim.x) {
    const float v = t > 0 ? Hraw[(size_t)(rowoff[t - 1] + b) * H + j] : 0.f;
    if constexpr (sizeof(OpT) == 2) out[(size_t)r * H + j] = f2bf(v); else out[(size_t)r * H + j] = v;
  }
}

// ---- Dropout / ReLU / LayerNorm backward ----------------------------------------------------------------
// one wave per row.  y = pre-LN activations, xhat = (y - mu) rstd, e = dropout(relu(xhat gamma + beta)).
// dY = rstd (dxh - mean(dxh) - xhat mean(dxh xhat)), dxh = de gamma.  Partials of dgamma/dbeta per block
// (4 rows) are written to part[blk][2][E] and summed in order by colsum_stage2 (deterministic).
template <int MAXV>
__global__ __launch_bounds__(256) void ln_relu_bwd_rows_kernel(
    const float* __restrict__ dE, const float* __restrict__ Y, const float* __restrict__ stats,
    const float* __restrict__ gamma, const float* __restrict__ beta, int nrows, int E, float drop_p,
    unsigned long long seed, int row0_abs, float* __restrict__ dY, float* __restrict__ part, int relu, int accumulate,
    bf16_t* __restrict__ dYb /* nullable: the same rows in bf16 = the k-major A operand of layer1's wgrad (gemm_tn.hip) */) {
  __shared__ float sg[4][4096 / 1];   // per-wave dgamma contributions are reduced through LDS below (E <= 4096)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = E / 256;
  const int r = blockIdx.x * 4 + wave;
  const bool live = r < nrows;
  const unsigned thresh = drop_p > 0.f ? (unsigned)(drop_p * 4294967296.0) : 0u;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  float4 dgam[MAXV], dbet[MAXV];
  float mu = 0.f, rstd = 0.f;
  if (live) { mu = stats[2 * r]; rstd = stats[2 * r + 1]; }
  float4 dxh[MAXV], xh[MAXV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (i < nv) {
      const int c = (i * 64 + lane) * 4;
      float4 y = make_float4(0, 0, 0, 0), de = y;
      if (live) { y = *(const float4*)(Y + (size_t)r * E + c); de = *(const float4*)(dE + (size_t)r * E + c); }
      const float4 g = *(const float4*)(gamma + c);
      const float4 bt = *(const float4*)(beta + c);
      float yy[4] = {y.x, y.y, y.z, y.w}, dd[4] = {de.x, de.y, de.z, de.w}, gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {bt.x, bt.y, bt.z, bt.w};
      float xo[4], dxo[4], dgo[4], dbo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float x = (yy[k] - mu) * rstd;
        const float pre = x * gg[k] + bb[k];
        float d = (live && (!relu || pre > 0.f)) ? dd[k] : 0.f;           // ReLU (MiniROAD layer1) / none (Transformer)
        if (thresh) d = dropout_keep_(seed, (size_t)(row0_abs + r) * E + c + k, thresh) ? d * keep_scale : 0.f;
        xo[k] = x; dgo[k] = d * x; dbo[k] = d; dxo[k] = d * gg[k];
        s1 += dxo[k]; s2 += dxo[k] * x;
      }
      xh[i] = make_float4(xo[0], xo[1], xo[2], xo[3]);
      dxh[i] = make_float4(dxo[0], dxo[1], dxo[2], dxo[3]);
      dgam[i] = make_float4(dgo[0], dgo[1], dgo[2], dgo[3]);
      dbet[i] = make_float4(dbo[0], dbo[1], dbo[2], dbo[3]);
    }
  const float m1 = wave_sum(s1) / (float)E, m2 = wave_sum(s2) / (float)E;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (i < nv && live) {
      const int c = (i * 64 + lane) * 4;
      float4 o;
      o.x = rstd * (dxh[i].x - m1 - xh[i].x * m2); o.y = rstd * (dxh[i].y - m1 - xh[i].y * m2);
      o.z = rstd * (dxh[i].z - m1 - xh[i].z * m2); o.w = rstd * (dxh[i].w - m1 - xh[i].w * m2);
      if (accumulate) {                                                  // residual stream: dx += (branch gradient)
        const float4 p = *(const float4*)(dY + (size_t)r * E + c);
        o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
      }
      *(float4*)(dY + (size_t)r * E + c) = o;
      if (dYb) *(uint2*)(dYb + (size_t)r * E + c) = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
    }
  // block partials of dgamma (pass 0) and dbeta (pass 1): fixed order wave 0..3
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (i < nv) {
        const int c = (i * 64 + lane) * 4;
        *(float4*)(&sg[wave][c]) = pass == 0 ? dgam[i] : dbet[i];
      }
    __syncthreads();
    for (int c = threadIdx.x; c < E; c += 256)
      part[((size_t)blockIdx.x * 2 + pass) * E + c] = (sg[0][c] + sg[1][c]) + (sg[2][c] + sg[3][c]);
  }
}

// ---- launchers -------------------------------------------------------------------------------------
void launch_oad_loss(const float* const* logit_ptrs, const float* const* target_ptrs, const int* lens, int n_clips, int C,
                     float* loss_out, float* const* dlogit_ptrs, float grad_scale, hipStream_t s, bool sum) {
  oad_loss_kernel<<<1, 64 * OAD_WAVES, 0, s>>>(logit_ptrs, target_ptrs, lens, n_clips, C, loss_out, dlogit_ptrs, grad_scale, sum ? 1.0f : (float)n_clips);
}
void launch_gather_dlogits(bool bf16, const float* const* dl_ptrs, const int* rowoff, const int* sorted_clip, int t_max,
                           int nrows, int C, int Cpad, void* out, hipStream_t s) {
  if (nrows <= 0) return;
  if (bf16) gather_dlogits_kernel<bf16_t><<<nrows, 128, 0, s>>>(dl_ptrs, rowoff, sorted_clip, t_max, nrows, C, Cpad, (bf16_t*)out);
  else gather_dlogits_kernel<float><<<nrows, 128, 0, s>>>(dl_ptrs, rowoff, sorted_clip, t_max, nrows, C, Cpad, (float*)out);
}
// in_bf16 / out_bf16 select element types
void launch_transpose_convert(bool in_bf16, bool out_bf16, const void* src, int M, int N, int ld_src, void* dst, int Mpad,
                              hipStream_t s) {
  dim3 grid((N + 31) / 32, (Mpad + 31) / 32);
  if (in_bf16 && out_bf16) transpose_convert_kernel<bf16_t, bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)src, M, N, ld_src, (bf16_t*)dst, Mpad);
  else if (in_bf16) transpose_convert_kernel<bf16_t, float><<<grid, 256, 0, s>>>((const bf16_t*)src, M, N, ld_src, (float*)dst, Mpad);
  else if (out_bf16) transpose_convert_kernel<float, bf16_t><<<grid, 256, 0, s>>>((const float*)src, M, N, ld_src, (bf16_t*)dst, Mpad);
  else transpose_convert_kernel<float, float><<<grid, 256, 0, s>>>((const float*)src, M, N, ld_src, (float*)dst, Mpad);
}
// part must hold ceil(M/64) * N floats
void launch_colsum(const float* src, int M, int N, float* part, float* out, hipStream_t s) {
  const int nb = (M + COLSUM_RB - 1) / COLSUM_RB;
  dim3 g1((N + 255) / 256, nb);
  colsum_stage1_kernel<float><<<g1, 256, 0, s>>>(src, M, N, part);
  colsum_stage2_kernel<<<(N + 63) / 64, 1024, 0, s>>>(part, nb, N, out);
}
// the same over bf16 rows (fp32 accumulation)
void launch_colsum_bf16(const void* src, int M, int N, float* part, float* out, hipStream_t s) {
  const int nb = (M + COLSUM_RB - 1) / COLSUM_RB;
  dim3 g1((N + 255) / 256, nb);
  colsum_stage1_kernel<bf16_t><<<g1, 256, 0, s>>>((const bf16_t*)src, M, N, part);
  colsum_stage2_kernel<<<(N + 63) / 64, 1024, 0, s>>>(part, nb, N, out);
}
void launch_colsum_stage2(const float* part, int nb, int N, float* out, hipStream_t s) {
  colsum_stage2_kernel<<<(N + 63) / 64, 1024, 0, s>>>(part, nb, N, out);
}
void launch_gru_bwd_step(bool bf16, int t, int na, int na_next, int row_t, int row_tm1, int H, const float* dHout,
                         const float* carry_in, const float* dhpart, const float* R, const float* Z, const float* Nn,
                         const float* GHN, const float* Hraw, float* carry_out, float* dGI, float* dGH, void* dGIop,
                         void* dGHop, hipStream_t s) {
  const int n = na * H;
  if (n <= 0) return;
  if (bf16)
    gru_bwd_step_kernel<bf16_t><<<(n + 255) / 256, 256, 0, s>>>(t, na, na_next, row_t, row_tm1, H, dHout, carry_in, dhpart, R, Z, Nn, GHN, Hraw, carry_out, dGI, dGH, (bf16_t*)dGIop, (bf16_t*)dGHop);
  else
    gru_bwd_step_kernel<float><<<(n + 255) / 256, 256, 0, s>>>(t, na, na_next, row_t, row_tm1, H, dHout, carry_in, dhpart, R, Z, Nn, GHN, Hraw, carry_out, dGI, dGH, (float*)dGIop, (float*)dGHop);
}
void launch_relu_mask(const float* dHrelu, const float* Hraw, size_t n, float* out, hipStream_t s) {
  int grid = (int)((n + 255) / 256);
  if (grid > 4096) grid = 4096;
  relu_mask_kernel<<<grid, 256, 0, s>>>(dHrelu, Hraw, n, out);
}
void launch_build_hprev(bool bf16, const float* Hraw, const int* rowoff, int t_max, int nrows, int H, void* out,
                        hipStream_t s) {
  if (nrows <= 0) return;
  if (bf16) build_hprev_kernel<bf16_t><<<nrows, 256, 0, s>>>(Hraw, rowoff, t_max, nrows, H, (bf16_t*)out);
  else build_hprev_kernel<float><<<nrows, 256, 0, s>>>(Hraw, rowoff, t_max, nrows, H, (float*)out);
}
// part must hold ceil(nrows/4) * 2 * E floats; returns the number of row blocks
int launch_ln_relu_bwd(const float* dE, const float* Y, const float* stats, const float* gamma, const float* beta, int nrows,
                       int E, float drop_p, unsigned long long seed, int row0_abs, float* dY, float* part, hipStream_t s,
                       int relu, int accumulate, void* dYb) {
  const int nb = (nrows + 3) / 4;
  if (nb <= 0) return 0;
  if (E <= 2048) ln_relu_bwd_rows_kernel<8><<<nb, 256, 0, s>>>(dE, Y, stats, gamma, beta, nrows, E, drop_p, seed, row0_abs, dY, part, relu, accumulate, (bf16_t*)dYb);
  else ln_relu_bwd_rows_kernel<16><<<nb, 256, 0, s>>>(dE, Y, stats, gamma, beta, nrows, E, drop_p, seed, row0_abs, dY, part, relu, accumulate, (bf16_t*)dYb);
  return nb;
}
