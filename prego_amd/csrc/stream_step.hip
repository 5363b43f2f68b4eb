// Streaming inference: ONE new frame for each of n <= 16 independent streams, the GRU state carried by
// the caller - MROAD.forward (model/rnn/rnn.py:51-71) with T = 1 and h0 = the state the previous call left.  The reference never
// exposes this (its eval loop runs whole videos), but it is how an online detector is driven.
//
// With one row per stream every layer is a skinny product that reads its whole weight matrix once (36 MB of bf16 per frame):
// HBM-latency bound, so the design goal is "everything in flight at once, as few launches as possible" - not the batched
// path's tiling.  Three (<= 4 streams: LayerNorm inside the second product) or four launches per frame instead of the batched path's seven plus its plan / table staging:
//   stream_gemv        layer1: y = [rgb | flow] W1^T + b1            (fp32 features converted in registers, no pack kernel)
//   ln_relu_rows       LayerNorm + ReLU (the batched path's kernel, rowwise.hip)
//   stream_gemv x 2    gi = e W_ih^T + b_ih (+ b_hh for r, z)  and  gh = h W_hh^T   in ONE launch (two problem descriptors)
//   stream_gates_head  GRU gates + state update, ReLU, classifier, softmax, argmax: one workgroup per stream
// stream_gemv: a workgroup owns 16 output features (one MFMA M tile), wave q the K-quarter [q K/4, (q+1) K/4); every weight
// fragment of the wave (up to 32 x 16 B per lane) is requested before the first MFMA, the streams ride on the MFMA N dimension
// (lanes whose stream index is >= n load nothing), the four partial tiles meet in LDS and are added in K order.
#include "common.h"
#include "kernels.h"

#define SG_MAXKS 32            // k-steps of 32 per wave: K <= 4096

struct GemvProb {
  const bf16_t* W;             // [Nout][K] bf16, row-major
  const void* X;               // input columns [0, kx1): [n][ldx], fp32 or bf16
  const void* X2;              // input columns [kx1, K): [n][ldx2]; nullptr = zeros (all-zero flow half)
  const float* bias;           // [Nout], nullable
  float* Y;                    // [n][Nout] fp32
  const float* ln_g;           // non-null: X is the fp32 PRE-LayerNorm row [n][K] (n <= 4): the workgroup normalises it itself
  const float* ln_b;
  float ln_eps;
  int Nout, K, kx1, ldx, ldx2, x_bf16, block0;
};
struct GemvArgs { GemvProb p[2]; int nprob, n, rows; };

// MAXKS: k-steps of 32 a wave may hold (32: K <= 4096, one workgroup per CU; 16: K <= 2048, 192 registers, two per CU).
// a.rows: output features per workgroup, 8 or 16 (8 = half an MFMA M tile: the matrix pipe is idle anyway, and 2048 outputs
// then make 256 workgroups - every CU pulls its share of the weights; the per-CU request rate, not HBM, bounds this kernel).
// A lane requests 32 CONTIGUOUS bytes of its weight row per pair of k-steps (the four lanes of a row cover one 128-byte line);
// the contraction index is permuted accordingly - MFMA 2p takes columns 16 g .. 16 g + 7 of the pair's 64, MFMA 2p + 1 columns
// 16 g + 8 .. 16 g + 15 - and the input fragment is loaded with the same permutation.
// LNX: problems whose ln_g is set take their input through LayerNorm + ReLU (rnn.py:41-42) inside the kernel: the workgroup loads
// the <= 4 pre-LayerNorm rows (requests issued BEFORE the weight requests, so that waiting for them leaves the weights in
// flight), computes the two-pass statistics with two block reductions, and reads its input fragments from the normalised bf16
// rows in LDS.  384 workgroups repeat the same 8 KB row: cheaper than the extra launch of a LayerNorm kernel (5.9 us).
// OT: 16-bit operand type tag (bf16_t / f16_t, common.h)
template <int MAXKS, bool LNX, typename OT = bf16_t>
__global__ __launch_bounds__(256, MAXKS > 16 ? 1 : 2) void stream_gemv_kernel(GemvArgs a) {
  __shared__ f32x4 red[4][64];
  __shared__ __attribute__((aligned(16))) bf16_t xs[LNX ? 4 : 1][LNX ? MAXKS * 128 : 8];
  __shared__ float sred[2][4][4];
  const int pi = (a.nprob > 1 && (int)blockIdx.x >= a.p[1].block0) ? 1 : 0;
  const GemvProb p = a.p[pi];
  const int tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int j0 = ((int)blockIdx.x - p.block0) * a.rows;
  const int kq = p.K >> 2, npair = kq >> 6;                  // K % 256 == 0
  const bool live = l15 < a.n, wlive = l15 < a.rows;
  const bf16_t* wrow = p.W + (size_t)(j0 + (wlive ? l15 : 0)) * p.K + q * kq + 16 * g;
  constexpr int MAXP = MAXKS / 2;
  // every global request of the kernel goes out up front: a dependent round trip to memory costs ~2 us here, the arithmetic nothing
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr && q == 0 && 4 * g < a.rows) bias4 = *(const f32x4*)(p.bias + j0 + 4 * g);
  const bool ln = LNX && p.ln_g != nullptr;                  // workgroup-uniform
  constexpr int EPL = MAXKS / 2;                             // elements of a row per thread at the largest K (K / 256)
  float yv[LNX ? 4 : 1][LNX ? EPL : 1], gv[LNX ? EPL : 1], bv[LNX ? EPL : 1];
  const int epl = p.K >> 8;                                  // K % 2048 == 0 in this mode: 8 or 16 elements per thread
  if constexpr (LNX) {
    if (ln) {
#pragma unroll
      for (int c = 0; c < EPL; c += 4)
        if (c < epl) {
          const f32x4 g4 = *(const f32x4*)(p.ln_g + tid * epl + c), b4 = *(const f32x4*)(p.ln_b + tid * epl + c);
#pragma unroll
          for (int k = 0; k < 4; ++k) { gv[c + k] = g4[k]; bv[c + k] = b4[k]; }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const f32x4 y4 = r < a.n ? *(const f32x4*)((const float*)p.X + (size_t)r * p.ldx + tid * epl + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) yv[r][c + k] = y4[k];
          }
        }
    }
  }

  u32x4 wa[MAXP][2];
  u32x4 xr[MAXP][4];
#pragma unroll
  for (int pr = 0; pr < MAXP; ++pr) {
    wa[pr][0] = (u32x4){0u, 0u, 0u, 0u}; wa[pr][1] = (u32x4){0u, 0u, 0u, 0u};
    if (pr < npair && wlive) {                               // read once: streams past the L2
      wa[pr][0] = __builtin_nontemporal_load((const u32x4*)(wrow + pr * 64));
      wa[pr][1] = __builtin_nontemporal_load((const u32x4*)(wrow + pr * 64 + 8));
    }
  }
  if constexpr (LNX) {
    if (ln) {
      const float invK = 1.0f / (float)p.K;
      float mu[4], rstd[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) if (r < a.n) {
        float sacc = 0.f;
#pragma unroll
        for (int c = 0; c < EPL; ++c) if (c < epl) sacc += yv[r][c];
        sacc = wave_sum(sacc);
        if (lane == 0) sred[0][q][r] = sacc;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) if (r < a.n) {
        mu[r] = ((sred[0][0][r] + sred[0][1][r]) + (sred[0][2][r] + sred[0][3][r])) * invK;
        float qacc = 0.f;
#pragma unroll
        for (int c = 0; c < EPL; ++c) if (c < epl) { const float d = yv[r][c] - mu[r]; qacc += d * d; }
        qacc = wave_sum(qacc);
        if (lane == 0) sred[1][q][r] = qacc;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) if (r < a.n) {
        rstd[r] = 1.0f / sqrtf(((sred[1][0][r] + sred[1][1][r]) + (sred[1][2][r] + sred[1][3][r])) * invK + p.ln_eps);
#pragma unroll
        for (int c = 0; c < EPL; c += 2)
          if (c < epl) {
            const float o0 = fmaxf((yv[r][c] - mu[r]) * rstd[r] * gv[c] + bv[c], 0.f);
            const float o1 = fmaxf((yv[r][c + 1] - mu[r]) * rstd[r] * gv[c + 1] + bv[c + 1], 0.f);
            *(unsigned*)(&xs[r][tid * epl + c]) = op16<OT>::pack2_sat(o0, o1);
          }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int pr = 0; pr < MAXP; ++pr) {
#pragma unroll
    for (int v = 0; v < 4; ++v) xr[pr][v] = (u32x4){0u, 0u, 0u, 0u};
    if (LNX && ln) {
      if (pr < npair && live) {                              // n <= 4 rows: live lanes are l15 < n
        const bf16_t* src = &xs[l15][q * kq + pr * 64 + 16 * g];
        xr[pr][0] = *(const u32x4*)src; xr[pr][1] = *(const u32x4*)(src + 8);
      }
    } else if (pr < npair) {
      const int kb = q * kq + pr * 64;                       // wave-uniform: a pair lies in ONE input half (kx1 % 64 == 0)
      const bool first = kb < p.kx1;
      const void* base = first ? p.X : p.X2;
      const int ld = first ? p.ldx : p.ldx2;
      const size_t off = (size_t)l15 * ld + (first ? kb : kb - p.kx1) + 16 * g;
      if (base != nullptr && live) {
        if (p.x_bf16) {
          xr[pr][0] = *(const u32x4*)((const bf16_t*)base + off); xr[pr][1] = *(const u32x4*)((const bf16_t*)base + off + 8);
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v) xr[pr][v] = *(const u32x4*)((const float*)base + off + 4 * v);
        }
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);                         // every request is out before the first use
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pr = 0; pr < MAXP; ++pr)
    if (pr < npair) {
      u32x4 x0 = xr[pr][0], x1 = xr[pr][1];
      if (!p.x_bf16 && !(LNX && ln)) {
        const f32x4 f0 = __builtin_bit_cast(f32x4, xr[pr][0]), f1 = __builtin_bit_cast(f32x4, xr[pr][1]);
        const f32x4 f2 = __builtin_bit_cast(f32x4, xr[pr][2]), f3 = __builtin_bit_cast(f32x4, xr[pr][3]);
        x0 = (u32x4){op16<OT>::pack2_sat(f0[0], f0[1]), op16<OT>::pack2_sat(f0[2], f0[3]), op16<OT>::pack2_sat(f1[0], f1[1]), op16<OT>::pack2_sat(f1[2], f1[3])};
        x1 = (u32x4){op16<OT>::pack2_sat(f2[0], f2[1]), op16<OT>::pack2_sat(f2[2], f2[3]), op16<OT>::pack2_sat(f3[0], f3[1]), op16<OT>::pack2_sat(f3[2], f3[3])};
      }
      acc = op16<OT>::mfma(__builtin_bit_cast(bf16x8, wa[pr][0]), __builtin_bit_cast(bf16x8, x0), acc);
      acc = op16<OT>::mfma(__builtin_bit_cast(bf16x8, wa[pr][1]), __builtin_bit_cast(bf16x8, x1), acc);
    }
  red[q][lane] = acc;
  __syncthreads();
  if (q == 0 && 4 * g < a.rows) {
    // accumulator element e of lane (column l15 = stream, g): output feature j0 + 4 g + e
    f32x4 r = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    r += bias4;
    if (live) *(f32x4*)(p.Y + (size_t)l15 * p.Nout + j0 + 4 * g) = r;
  }
}

// GRU gates + state update (rnn.py:61, torch.nn.GRU's equations), ReLU + classifier (rnn.py:62-64), eval softmax (rnn.py:66-70),
// np.argmax (eval.py:53, first max wins).  gi already holds b_ih (+ b_hh for the r, z rows), gh = h W_hh^T without bias.
// One workgroup per stream.  The classifier is the batched head's arithmetic for one frame: logits^T = W_c relu(h)^T on the MFMA
// with the frame in column 0, wave q the K-quarter of every class tile (all NT x 8 weight fragments requested at once), the four
// partials added in K order.
template <int NT, typename OT = bf16_t>
__global__ __launch_bounds__(256, 1) void stream_gates_head_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                                   const float* __restrict__ b_hn, float* __restrict__ h_state,
                                                                   const bf16_t* __restrict__ wc, const float* __restrict__ bc, int C,
                                                                   int softmax, float* __restrict__ out, int* __restrict__ argmax) {
  constexpr int H = 1024;
  __shared__ __attribute__((aligned(16))) bf16_t shb[H];      // relu(h_t) as the bf16 the head multiplies
  __shared__ f32x4 redh[4][NT][4];
  __shared__ float sl[128];
  const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  // classifier weights first: they do not depend on the gates (requests in flight under the gate math)
  bf16x8 wa[NT][8];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) wa[ct][ks] = *(const bf16x8*)(wc + (size_t)(ct * 16 + l15) * H + q * 256 + ks * 32 + 8 * g);
  const float bc_t = tid < C ? bc[tid] : 0.f;               // requested with everything else (a late load is one more ~2 us round trip)
  const float* gis = gi + (size_t)s * 3 * H;
  const float* ghs = gh + (size_t)s * 3 * H;
  {
    const int u = tid * 4;                                    // four consecutive hidden units per thread
    const f32x4 ir = *(const f32x4*)(gis + u), iz = *(const f32x4*)(gis + H + u), in_ = *(const f32x4*)(gis + 2 * H + u);
    const f32x4 hr = *(const f32x4*)(ghs + u), hz = *(const f32x4*)(ghs + H + u), hn_ = *(const f32x4*)(ghs + 2 * H + u);
    const f32x4 bn = *(const f32x4*)(b_hn + u), hp = *(const f32x4*)(h_state + (size_t)s * H + u);
    f32x4 hnew;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float r = sigmoidf_(ir[e] + hr[e]);
      const float z = sigmoidf_(iz[e] + hz[e]);
      const float n = tanhf_(in_[e] + r * (hn_[e] + bn[e]));
      hnew[e] = (1.0f - z) * n + z * hp[e];
    }
    *(f32x4*)(h_state + (size_t)s * H + u) = hnew;
    u32x2 w; w[0] = op16<OT>::pack2(fmaxf(hnew[0], 0.f), fmaxf(hnew[1], 0.f)); w[1] = op16<OT>::pack2(fmaxf(hnew[2], 0.f), fmaxf(hnew[3], 0.f));
    *(u32x2*)(shb + u) = w;
  }
  __syncthreads();
  f32x4 acc[NT];
#pragma unroll
  for (int ct = 0; ct < NT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    u32x4 hb = *(const u32x4*)(shb + q * 256 + ks * 32 + 8 * g);
    if (l15 != 0) hb = (u32x4){0u, 0u, 0u, 0u};              // the frame is column 0 of the N dimension
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) acc[ct] = op16<OT>::mfma(wa[ct][ks], __builtin_bit_cast(bf16x8, hb), acc[ct]);
  }
  if (l15 == 0) {
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) redh[q][ct][g] = acc[ct];
  }
  __syncthreads();
  if (tid < C) {                                              // class c = tile c / 16, row c % 16 = 4 g + e
    const int ct = tid >> 4, r = tid & 15, gg = r >> 2, e = r & 3;
    sl[tid] = ((redh[0][ct][gg][e] + redh[1][ct][gg][e]) + (redh[2][ct][gg][e] + redh[3][ct][gg][e])) + bc_t;
  }
  __syncthreads();
  if (q == 0) {
    // C <= 128: two classes per lane
    const float v0 = lane < C ? sl[lane] : -INFINITY, v1 = lane + 64 < C ? sl[lane + 64] : -INFINITY;
    const float mx = wave_max(fmaxf(v0, v1));
    // first index holding the maximum (np.argmax)
    int cand = v0 == mx ? lane : (v1 == mx ? lane + 64 : 0x7fffffff);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(cand, o, 64); cand = other < cand ? other : cand; }
    if (argmax != nullptr && lane == 0) argmax[s] = cand;
    if (out != nullptr) {
      if (softmax) {
        const float e0 = lane < C ? __expf(v0 - mx) : 0.f, e1 = lane + 64 < C ? __expf(v1 - mx) : 0.f;
        const float inv = 1.0f / wave_sum(e0 + e1);
        if (lane < C) out[(size_t)s * C + lane] = e0 * inv;
        if (lane + 64 < C) out[(size_t)s * C + lane + 64] = e1 * inv;
      } else {
        if (lane < C) out[(size_t)s * C + lane] = v0;
        if (lane + 64 < C) out[(size_t)s * C + lane + 64] = v1;
      }
    }
  }
}

// y[n][Nout] = x[n][K] W^T + bias for one or two problems in one launch.  Returns -1 on an unsupported shape.
int launch_stream_gemv(int nprob, const StreamGemv* pr, int n, hipStream_t s, bool f16) {
  if (nprob < 1 || nprob > 2 || n < 1 || n > 16) return -1;
  GemvArgs a{};
  a.nprob = nprob; a.n = n;
  int kmax = 0, tiles16 = 0;
  for (int i = 0; i < nprob; ++i) {
    if (pr[i].Nout % 16 || pr[i].K % 256 || pr[i].K > 128 * SG_MAXKS || pr[i].kx1 % 64 || pr[i].kx1 > pr[i].K) return -1;
    kmax = pr[i].K > kmax ? pr[i].K : kmax;
    tiles16 += pr[i].Nout / 16;
  }
  a.rows = tiles16 < 200 ? 8 : 16;                           // fewer than ~one workgroup per CU at 16 rows: halve the tile
  int blocks = 0;
  for (int i = 0; i < nprob; ++i) {
    a.p[i] = GemvProb{(const bf16_t*)pr[i].W, pr[i].X, pr[i].X2, pr[i].bias, pr[i].Y, pr[i].ln_g, pr[i].ln_b, pr[i].ln_eps, pr[i].Nout, pr[i].K,
                      pr[i].kx1, pr[i].ldx, pr[i].ldx2, pr[i].x_bf16, blocks};
    blocks += pr[i].Nout / a.rows;
  }
  bool any_ln = false;
  for (int i = 0; i < nprob; ++i)
    if (pr[i].ln_g != nullptr) {
      if (n > 4 || pr[i].K % 2048 || pr[i].kx1 != pr[i].K || pr[i].x_bf16 || !pr[i].ln_b) return -1;
      any_ln = true;
    }
  if (f16) {
    if (kmax > 2048) { if (any_ln) stream_gemv_kernel<32, true, f16_t><<<blocks, 256, 0, s>>>(a); else stream_gemv_kernel<32, false, f16_t><<<blocks, 256, 0, s>>>(a); }
    else { if (any_ln) stream_gemv_kernel<16, true, f16_t><<<blocks, 256, 0, s>>>(a); else stream_gemv_kernel<16, false, f16_t><<<blocks, 256, 0, s>>>(a); }
    return 0;
  }
  if (kmax > 2048) { if (any_ln) stream_gemv_kernel<32, true><<<blocks, 256, 0, s>>>(a); else stream_gemv_kernel<32, false><<<blocks, 256, 0, s>>>(a); }
  else { if (any_ln) stream_gemv_kernel<16, true><<<blocks, 256, 0, s>>>(a); else stream_gemv_kernel<16, false><<<blocks, 256, 0, s>>>(a); }
  return 0;
}

// H == 1024 (the handle's hidden size); C <= 128, wc holds ceil(C / 16) * 16 rows
int launch_stream_gates_head(const float* gi, const float* gh, const float* b_hn, float* h_state, const void* wc, const float* bc, int n,
                             int H, int C, int softmax, float* out, int* argmax, hipStream_t s, bool f16) {
  if (H != 1024 || C < 1 || C > 128) return -1;
#define SGH(NT)                                                                                                                              \
  do {                                                                                                                                       \
    if (f16) stream_gates_head_kernel<NT, f16_t><<<n, 256, 0, s>>>(gi, gh, b_hn, h_state, (const bf16_t*)wc, bc, C, softmax, out, argmax);   \
    else stream_gates_head_kernel<NT><<<n, 256, 0, s>>>(gi, gh, b_hn, h_state, (const bf16_t*)wc, bc, C, softmax, out, argmax);              \
  } while (0)
  switch ((C + 15) / 16) {
    case 1: SGH(1); break; case 2: SGH(2); break; case 3: SGH(3); break; case 4: SGH(4); break;
    case 5: SGH(5); break; case 6: SGH(6); break; case 7: SGH(7); break; default: SGH(8); break;
  }
#undef SGH
  return 0;
}
