// Flash-style multi-head attention for the "Transformer" path (ViTEnc) and for the causal AttentionLayer:
//   SelfAttention.forward   model/transformer_models/Attention.py:21-41   softmax(q k^T * dh^-0.5) v, no mask
//   FullAttention.forward   model/transformer_models/attn.py:35-57        masked_fill(triu(1), -inf) then softmax(scale * s)
// The reference materialises the [B,h,L,L] score tensor (33.6 MB per window at L = 1025); here scores never leave
// registers: per workgroup 64 queries (4 waves x 16 rows), K and V^T tiles of 64 keys staged in LDS, S = Q K^T on
// MFMA 16x16x32 bf16, online softmax with the row reduce done by 16-lane shuffles over the accumulator layout
// (col = lane&15, row = (lane>>4)*4 + reg), P re-laid out through a per-wave LDS tile, O += P V on MFMA.
// Causal: key tiles beyond the diagonal are skipped.  The softmax scale is folded into Q by the QKV GEMM epilogue.
// Layouts (bf16): Q, K [B, h, N, DH]; V^T [B, h, DH, Npad] (Npad multiple of 64, pad zeroed); out [B, N, h*DH].
#include "common.h"
#include "kernels.h"

#define AQ 64      // queries per workgroup
#define AK 64      // keys per tile

template <int DH>
__global__ __launch_bounds__(256) void flash_attention_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ Vt, bf16_t* __restrict__ out,
    int N, int Npad, int heads, int causal, float* __restrict__ lse /*nullable: [B,h,N] log-sum-exp of the scaled scores (training)*/) {
  constexpr int KS = DH / 32;            // k-steps of QK^T
  constexpr int DT = DH / 16;            // output column tiles
  constexpr int KLD = DH + 8;            // LDS row pitch of the K tile (elements): 16-byte skew per row
  constexpr int VLD = AK + 8;            // LDS row pitch of the V^T tile and of P
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* sK = (bf16_t*)smem;                         // [AK][KLD]
  bf16_t* sV = sK + AK * KLD;                         // [DH][VLD]
  bf16_t* sP = sV + DH * VLD;                         // [4 waves][16][VLD]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int bh = blockIdx.y;
  const int q0 = blockIdx.x * AQ;
  const bf16_t* Qb = Q + (size_t)bh * N * DH;
  const bf16_t* Kb = K + (size_t)bh * N * DH;
  const bf16_t* Vb = Vt + (size_t)bh * DH * Npad;

  // Q fragments (A operand: row = query, k = d)
  int qrow = q0 + wave * 16 + l15;
  const int qrow_ld = qrow < N ? qrow : N - 1;
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(Qb + (size_t)qrow_ld * DH + ks * 32 + 8 * l4);

  f32x4 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { m_run[e] = -INFINITY; l_run[e] = 0.f; }

  const int q_hi = (q0 + AQ - 1 < N - 1) ? q0 + AQ - 1 : N - 1;            // last query of this block
  const int n_tiles = causal ? (q_hi / AK + 1) : (N + AK - 1) / AK;
  for (int kt = 0; kt < n_tiles; ++kt) {
    const int k0 = kt * AK;
    __syncthreads();                                  // previous tile fully consumed
    // stage K tile: AK rows x DH elements (rows >= N clamped; masked below)
    for (int c = tid; c < AK * (DH / 8); c += 256) {
      const int r = c / (DH / 8), ch = c % (DH / 8);
      int kr = k0 + r; if (kr > N - 1) kr = N - 1;
      *(uint4*)(sK + r * KLD + ch * 8) = *(const uint4*)(Kb + (size_t)kr * DH + ch * 8);
    }
    // stage V^T tile: DH rows x AK keys (in bounds: Npad is a multiple of AK)
    for (int c = tid; c < DH * (AK / 8); c += 256) {
      const int r = c / (AK / 8), ch = c % (AK / 8);
      *(uint4*)(sV + r * VLD + ch * 8) = *(const uint4*)(Vb + (size_t)r * Npad + k0 + ch * 8);
    }
    __syncthreads();

    // S = Q K^T  (4 key tiles of 16)
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 kf = *(const bf16x8*)(sK + (j * 16 + l15) * KLD + ks * 32 + 8 * l4);
        s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kf, s[j], 0, 0, 0);
      }
    // mask + online softmax; this lane: rows qr = q0 + wave*16 + l4*4 + e, key columns k0 + j*16 + l15
    float alpha[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int qr = q0 + wave * 16 + l4 * 4 + e;
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = k0 + j * 16 + l15;
        const bool dead = key >= N || (causal && key > qr);
        const float v = dead ? -INFINITY : s[j][e];
        s[j][e] = v;
        mx = fmaxf(mx, v);
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      const float m_new = fmaxf(m_run[e], mx);
      const float msafe = (m_new == -INFINITY) ? 0.f : m_new;     // fully masked so far (rows >= N only)
      alpha[e] = __expf(m_run[e] - msafe);
      float rs = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p = __expf(s[j][e] - msafe);
        s[j][e] = p;
        rs += p;
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) rs += __shfl_xor(rs, off, 64);
      l_run[e] = l_run[e] * alpha[e] + rs;
      m_run[e] = m_new;
    }
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int e = 0; e < 4; ++e) o[d][e] *= alpha[e];
    // P -> LDS as [query row][key], bf16 (re-layout from accumulator form to A-operand form)
    bf16_t* pw = sP + wave * 16 * VLD;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) pw[(l4 * 4 + e) * VLD + j * 16 + l15] = f2bf(s[j][e]);
    __syncthreads();
    // O += P V : A = P (row = query, k = key), B = V^T tile (col = d, k = key)
#pragma unroll
    for (int ks = 0; ks < AK / 32; ++ks) {
      const bf16x8 pf = *(const bf16x8*)(pw + l15 * VLD + ks * 32 + 8 * l4);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const bf16x8 vf = *(const bf16x8*)(sV + (d * 16 + l15) * VLD + ks * 32 + 8 * l4);
        o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[d], 0, 0, 0);
      }
    }
  }
  // epilogue: out[b, q, head*DH + d] = O / l
  const int b = bh / heads, hd = bh % heads;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int qr = q0 + wave * 16 + l4 * 4 + e;
    if (qr < N) {
      const float inv = 1.0f / l_run[e];
      if (lse != nullptr && l15 == 0) lse[(size_t)bh * N + qr] = m_run[e] + logf(l_run[e]);
      bf16_t* orow = out + ((size_t)b * N + qr) * heads * DH + hd * DH;
#pragma unroll
      for (int d = 0; d < DT; ++d) orow[d * 16 + l15] = f2bf(o[d][e] * inv);
    }
  }
}

int launch_flash_attention(const void* Q, const void* K, const void* Vt, void* out, int B, int N, int Npad, int heads,
                           int dh, int causal, hipStream_t s, float* lse) {
  if (Npad % AK) return -1;
  dim3 grid((N + AQ - 1) / AQ, B * heads);
#define FA(D)                                                                                                   \
  do {                                                                                                          \
    const size_t lds = ((size_t)AK * (D + 8) + (size_t)D * (AK + 8) + 4 * 16 * (AK + 8)) * 2;                   \
    static DeviceOnce once;                                                                                     \
    once.run([&] { (void)hipFuncSetAttribute((const void*)flash_attention_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }); \
    flash_attention_kernel<D><<<grid, 256, lds, s>>>((const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)Vt, (bf16_t*)out, N, \
                                                     Npad, heads, causal, lse);                                 \
  } while (0)
  if (dh == 256) FA(256);
  else if (dh == 128) FA(128);
  else if (dh == 64) FA(64);
  else return -1;
#undef FA
  return 0;
}
