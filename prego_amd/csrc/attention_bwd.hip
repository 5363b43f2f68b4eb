// Flash-style attention BACKWARD for the "Transformer" path (training of ViTEnc, trainer/train.py:20-24 over
// model/transformer_models/Attention.py:21-41):   a = softmax(q k^T * dh^-0.5) [mask],  o = a v
//   dv = a^T do;   da = do v^T;   ds = a * (da - rowsum(da * a));   dq = ds k * dh^-0.5;   dk = ds^T q * dh^-0.5
// The [B,h,N,N] probabilities are never stored: P is recomputed from Q', K and the forward's log-sum-exp per query row
// (Q' = q * dh^-0.5 is what the QKV GEMM epilogue wrote, so scores = Q' K^T and dk = ds^T Q' carries the scale already),
// and rowsum(da * a) = rowsum(do * o) = delta comes from a small row kernel.
//
// One kernel body, two passes (deterministic: no atomics, every output element is summed by ONE wave in a fixed order):
//   BYKEY = true   a workgroup owns 64 keys   (16 per wave) and sweeps the query tiles:  dV, dK
//   BYKEY = false  a workgroup owns 64 queries (16 per wave) and sweeps the key tiles:    dQ
// "Own index on the lane" (cdna_hip_programming.md, Attention backward): with the swept 64 rows as the MFMA M dimension and
// the own 16 indices as N, the accumulators of  X = tile . own^T  (rows = swept index in the registers, column = own index on
// the lane) are ALREADY the B operand of the products that sum over the swept index:
//   BYKEY:  S = Q'_i K_j^T, dP = dO_i V_j^T -> P, dS;   dV_j^T += dO_i^T P,   dK_j^T += Q'_i^T dS
//   BYQ  :  S^T = K_j Q'_i^T, dP^T = V_j dO_i^T -> dS^T;                       dQ_i^T += K_j^T dS^T
// (two 16-row accumulator tiles give the 8 k-elements of a lane in the order {4g..4g+3, 16+4g..16+4g+3}; the A operand of
// those products is the swept tile read TRANSPOSED from its row-major LDS image with ds_read_b64_tr_b16 in exactly that k
// order: no LDS round trip for P or dS and one LDS image per tile).  The transposed accumulators (row = head-dim index,
// column = own index) leave as 8-byte stores of four consecutive head-dim elements.
// Layouts (bf16): Qs, K, V [B, h, N, DH]; dO, O [B, N, h*DH]; lse, delta fp32 [B, h, N]; output dqkv [B*N, 3*h*DH] with
// column = which * (h*DH) + head * DH + d (the layout of qkv.reshape(B, N, 3, h, dh), Attention.py:23-27).
#include "common.h"
#include "kernels.h"

#define BQ 64      // swept rows per LDS tile

typedef short v4s __attribute__((ext_vector_type(4)));

template <int DH, bool BYKEY>
__global__ __launch_bounds__(256, 1) void attention_bwd_kernel(
    const bf16_t* __restrict__ Qs, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
    const bf16_t* __restrict__ dO, const float* __restrict__ lse, const float* __restrict__ delta,
    bf16_t* __restrict__ dqkv, int N, int heads, int causal, float q_scale, unsigned drop_thresh, float drop_scale,
    unsigned long long drop_seed) {
  // drop_thresh != 0: the forward dropped the probabilities (mask of (seed, (bh * N + query) * N + key), scale 1 / (1 - p)):
  // dV uses the dropped P, the gradient of the probabilities passes through the same mask; delta = rowsum(dO * O) still holds
  constexpr int KS = DH / 32;            // k-steps over the head dim (S, dP)
  constexpr int DT = DH / 16;            // 16-row tiles of the transposed outputs
  constexpr int LD = DH + 8;             // LDS row pitch (elements); (DH+8)*2 bytes is a multiple of 8 (ds_read_b64_tr_b16)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* s1 = (bf16_t*)smem;            // swept tile 1: BYKEY ? Q' : K     [BQ][LD]
  bf16_t* s2 = s1 + BQ * LD;             // swept tile 2: BYKEY ? dO : V     [BQ][LD]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  // causal: the linear workgroup id (handed to the XCDs round-robin) is split as (rank of the block by work, bh): heaviest blocks
  // first, the same mix of heavy and light blocks on every XCD (see flash_attention_v2_kernel)
  const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y, n_bh = (int)gridDim.y;
  const int bh = causal ? lin % n_bh : (int)blockIdx.y, b = bh / heads, hd = bh % heads;
  const int blk = !causal ? (int)blockIdx.x : (BYKEY ? lin / n_bh : (int)gridDim.x - 1 - lin / n_bh);
  const int E = heads * DH;
  const int own = blk * 64 + wave * 16 + l15;                   // my key (BYKEY) / my query
  const int own_ld = own < N ? own : N - 1;
  const size_t hbase = (size_t)bh * N * DH;                     // [B,h,N,DH] tensors
  const size_t abase = (size_t)b * N * E + (size_t)hd * DH;     // [B,N,E] tensors, this head's columns

  // own operands (B fragments: column = own index, k = head dim)
  bf16x8 o1[KS], o2[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if constexpr (BYKEY) {
      o1[ks] = *(const bf16x8*)(K + hbase + (size_t)own_ld * DH + ks * 32 + 8 * g);
      o2[ks] = *(const bf16x8*)(V + hbase + (size_t)own_ld * DH + ks * 32 + 8 * g);
    } else {
      o1[ks] = *(const bf16x8*)(Qs + hbase + (size_t)own_ld * DH + ks * 32 + 8 * g);
      o2[ks] = *(const bf16x8*)(dO + abase + (size_t)own_ld * E + ks * 32 + 8 * g);
    }
  }
  float own_lse = 0.f, own_delta = 0.f;
  if constexpr (!BYKEY) { own_lse = lse[(size_t)bh * N + own_ld]; own_delta = delta[(size_t)bh * N + own_ld]; }

  f32x4 acc1[BYKEY ? DT : 1], acc2[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) acc2[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (BYKEY) {
#pragma unroll
    for (int d = 0; d < DT; ++d) acc1[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // swept range: causal -> a key only meets queries >= key, a query only keys <= query
  const int blk_lo = blk * 64, blk_hi = blk_lo + 63;
  const int n_tiles = (N + BQ - 1) / BQ;
  const int t_begin = (causal && BYKEY) ? blk_lo / BQ : 0;
  const int t_end = (causal && !BYKEY) ? ((blk_hi < N - 1 ? blk_hi : N - 1) / BQ + 1) : n_tiles;

  for (int it = t_begin; it < t_end; ++it) {
    const int r0 = it * BQ;
    __syncthreads();                                             // previous tile fully consumed
    for (int c = tid; c < BQ * (DH / 8); c += 256) {
      const int r = c / (DH / 8), ch = c % (DH / 8);
      int rr = r0 + r; if (rr > N - 1) rr = N - 1;               // clamped; masked below
      if constexpr (BYKEY) {
        *(uint4*)(s1 + r * LD + ch * 8) = *(const uint4*)(Qs + hbase + (size_t)rr * DH + ch * 8);
        *(uint4*)(s2 + r * LD + ch * 8) = *(const uint4*)(dO + abase + (size_t)rr * E + ch * 8);
      } else {
        *(uint4*)(s1 + r * LD + ch * 8) = *(const uint4*)(K + hbase + (size_t)rr * DH + ch * 8);
        *(uint4*)(s2 + r * LD + ch * 8) = *(const uint4*)(V + hbase + (size_t)rr * DH + ch * 8);
      }
    }
    __syncthreads();

    // X1 = tile1 . own1^T (scores), X2 = tile2 . own2^T (dP): rows = swept index (16 t + 4 g + e), column = own index
    f32x4 x1[4], x2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { x1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; x2[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 a1 = *(const bf16x8*)(s1 + (t * 16 + l15) * LD + ks * 32 + 8 * g);
        const bf16x8 a2 = *(const bf16x8*)(s2 + (t * 16 + l15) * LD + ks * 32 + 8 * g);
        x1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, o1[ks], x1[t], 0, 0, 0);
        x2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, o2[ks], x2[t], 0, 0, 0);
      }
    // P = exp(S - lse[query]) (0 where masked / out of range), dS = P (dP - delta[query])
    unsigned pw[2][4], dw[2][4];                                 // B fragments (packed bf16 pairs) of the two k-steps over the 64 swept rows
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float p[4], ds[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int sw = r0 + t * 16 + 4 * g + e;                  // swept index of this accumulator element
        const int q = BYKEY ? sw : own, key = BYKEY ? own : sw;
        const bool ok = sw < N && own < N && (!causal || key <= q);
        float l_, d_;
        if constexpr (BYKEY) {
          const int qi = sw < N ? sw : N - 1;
          l_ = lse[(size_t)bh * N + qi]; d_ = delta[(size_t)bh * N + qi];
        } else { l_ = own_lse; d_ = own_delta; }
        p[e] = ok ? __expf(x1[t][e] - l_) : 0.f;
        float dp = x2[t][e];
        float pd = p[e];                                         // the probability as it entered P.V
        if (drop_thresh) {
          const int qc = q < N ? q : N - 1, kc = key < N ? key : N - 1;
          const float mk = dropout_keep_(drop_seed, ((size_t)bh * N + qc) * N + kc, drop_thresh) ? drop_scale : 0.f;
          dp *= mk; pd *= mk;
        }
        ds[e] = p[e] * (dp - d_);
        p[e] = pd;
      }
      const int s = t >> 1, half = t & 1;                        // k-step, which four of its eight elements
      pw[s][2 * half] = pack_bf16x2(p[0], p[1]); pw[s][2 * half + 1] = pack_bf16x2(p[2], p[3]);
      dw[s][2 * half] = pack_bf16x2(ds[0], ds[1]); dw[s][2 * half + 1] = pack_bf16x2(ds[2], ds[3]);
    }
    bf16x8 pf[2], dsf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      pf[s] = __builtin_bit_cast(bf16x8, (u32x4){pw[s][0], pw[s][1], pw[s][2], pw[s][3]});
      dsf[s] = __builtin_bit_cast(bf16x8, (u32x4){dw[s][0], dw[s][1], dw[s][2], dw[s][3]});
    }
    // acc1^T += tile2^T . P   (BYKEY: dV^T += dO^T P);   acc2^T += tile1^T . dS   (dK^T += Q'^T dS  /  dQ^T += K^T dS^T)
    // A operand = the tile read transposed: lane (g, li) of the 16x16x32 fragment holds rows d = 16 dt + li, k elements
    // {32 s + 4 g + 0..3, 32 s + 16 + 4 g + 0..3}; ds_read_b64_tr_b16 delivers "column li of a 4-row block", the lane that
    // supplies the address of block row q', columns 4 p .. 4 p + 3 is lane 4 q' + p of the 16-lane group.
    const int qp = l15 >> 2, pp = l15 & 3;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int ra = (32 * s + 4 * g + qp) * LD + dt * 16 + 4 * pp;
        const int rb = ra + 16 * LD;
        typedef __attribute__((address_space(3))) v4s* lds_v4s;
        const v4s t1a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(s1 + ra));
        const v4s t1b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(s1 + rb));
        bf16x8 a1 = {t1a[0], t1a[1], t1a[2], t1a[3], t1b[0], t1b[1], t1b[2], t1b[3]};
        acc2[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, dsf[s], acc2[dt], 0, 0, 0);
        if constexpr (BYKEY) {
          const v4s t2a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(s2 + ra));
          const v4s t2b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(s2 + rb));
          bf16x8 a2 = {t2a[0], t2a[1], t2a[2], t2a[3], t2b[0], t2b[1], t2b[2], t2b[3]};
          acc1[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, pf[s], acc1[dt], 0, 0, 0);
        }
      }
  }

  // outputs: acc^T[dt][e] = grad[own][d = 16 dt + 4 g + e]: four consecutive head-dim elements = one 8-byte store
  if (own < N) {
    bf16_t* row = dqkv + ((size_t)b * N + own) * (3 * E) + (size_t)hd * DH;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + 4 * g;
      if constexpr (BYKEY) {
        uint2 kv, vv;
        kv.x = pack_bf16x2(acc2[dt][0], acc2[dt][1]); kv.y = pack_bf16x2(acc2[dt][2], acc2[dt][3]);
        vv.x = pack_bf16x2(acc1[dt][0], acc1[dt][1]); vv.y = pack_bf16x2(acc1[dt][2], acc1[dt][3]);
        *(uint2*)(row + E + d) = kv;                              // dk
        *(uint2*)(row + 2 * E + d) = vv;                          // dv
      } else {
        uint2 qv;
        qv.x = pack_bf16x2(acc2[dt][0] * q_scale, acc2[dt][1] * q_scale);
        qv.y = pack_bf16x2(acc2[dt][2] * q_scale, acc2[dt][3] * q_scale);
        *(uint2*)(row + d) = qv;                                  // dq = scale * dq'
      }
    }
  }
}

// delta[b,h,n] = sum_d dO[b,n,h*DH+d] * O[b,n,h*DH+d]   (= rowsum(dP * P)); one wave per (row, head)
__global__ __launch_bounds__(256) void attention_delta_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
                                                              float* __restrict__ delta, int B, int N, int heads, int DH) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);              // (b*N + n) * heads + h
  if (w >= B * N * heads) return;
  const int hd = w % heads, m = w / heads, b = m / N, n = m % N;
  const size_t base = (size_t)m * heads * DH + (size_t)hd * DH;
  float s = 0.f;
  for (int d = lane * 4; d < DH; d += 256) {
    const uint2 a = *(const uint2*)(dO + base + d), c = *(const uint2*)(O + base + d);
    s += __uint_as_float(a.x << 16) * __uint_as_float(c.x << 16) + __uint_as_float(a.x & 0xFFFF0000u) * __uint_as_float(c.x & 0xFFFF0000u);
    s += __uint_as_float(a.y << 16) * __uint_as_float(c.y << 16) + __uint_as_float(a.y & 0xFFFF0000u) * __uint_as_float(c.y & 0xFFFF0000u);
  }
  s = wave_sum(s);
  if (lane == 0) delta[((size_t)b * heads + hd) * N + n] = s;
}

// dqkv [B*N, 3*heads*dh] bf16 := gradients of the fused qkv projection's output.  q_scale = dh^-0.5 (Attention.py:14).
int launch_attention_bwd(const void* Qs, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                         float* delta, void* dqkv, int B, int N, int heads, int dh, int causal, float q_scale, hipStream_t s,
                         unsigned drop_thresh, float drop_scale, unsigned long long drop_seed) {
  if (dh % 4) return -1;
  attention_delta_kernel<<<(B * N * heads + 3) / 4, 256, 0, s>>>((const bf16_t*)dO, (const bf16_t*)O, delta, B, N, heads, dh);
  dim3 grid((N + 63) / 64, B * heads);
#define AB(D)                                                                                                                 \
  do {                                                                                                                        \
    const size_t lds = (size_t)2 * BQ * (D + 8) * 2;                                                                          \
    static DeviceOnce once;                                                                                                   \
    once.run([&] {                                                                                                            \
      (void)hipFuncSetAttribute((const void*)attention_bwd_kernel<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      (void)hipFuncSetAttribute((const void*)attention_bwd_kernel<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    });                                                                                                                       \
    attention_bwd_kernel<D, true><<<grid, 256, lds, s>>>((const bf16_t*)Qs, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)dO, \
                                                         lse, delta, (bf16_t*)dqkv, N, heads, causal, q_scale, drop_thresh, drop_scale, drop_seed); \
    attention_bwd_kernel<D, false><<<grid, 256, lds, s>>>((const bf16_t*)Qs, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)dO, \
                                                          lse, delta, (bf16_t*)dqkv, N, heads, causal, q_scale, drop_thresh, drop_scale, drop_seed); \
  } while (0)
  if (dh == 256) AB(256);
  else if (dh == 128) AB(128);
  else if (dh == 64) AB(64);
  else return -1;
#undef AB
  return 0;
}
