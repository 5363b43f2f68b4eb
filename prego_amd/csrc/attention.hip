// Flash-style multi-head attention for the "Transformer" path (ViTEnc) and for the causal AttentionLayer:
//   SelfAttention.forward   model/transformer_models/Attention.py:21-41   softmax(q k^T * dh^-0.5) v, no mask
//   FullAttention.forward   model/transformer_models/attn.py:35-57        masked_fill(triu(1), -inf) then softmax(scale * s)
// The reference materialises the [B,h,L,L] score tensor (33.6 MB per window at L = 1025); here scores never leave registers.
// The softmax scale is folded into Q by the QKV GEMM epilogue.  Causal: key tiles beyond the diagonal are skipped and the
// heaviest query blocks are dispatched first.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

#define AK 64      // keys per tile

// ------------------------------------------------------------------------------------------------------
// "Query on the lane" (swapped QK^T, cdna_hip_programming.md T12 / Attention backward): S^T = K_tile Q^T has the keys in
// the accumulator registers and the query on the lane, so
//   * the softmax statistics of a query are per-LANE scalars (the reduce over keys is 16 in-register values + two shuffles),
//   * P^T is ALREADY the B operand of O^T += V_tile^T P^T (two 16-row accumulator tiles = the 8 k-elements of a lane, in the
//     order {4g..4g+3, 16+4g..16+4g+3}): no LDS round trip for P,
//   * V stays ROW-major [key][d] and is read transposed by ds_read_b64_tr_b16 in that k order: no V^T tensor at all,
//   * the output leaves as 16-byte stores of eight consecutive head-dim elements (one lane-row swap per pair of tiles).
// A wave owns QG groups of 16 queries (QG = 2: 128 queries per workgroup): every K / V fragment read from LDS feeds QG MFMAs,
// which halves the LDS bytes per FLOP (dh = 256 is otherwise LDS-bound: each wave re-reads the whole 64 KB tile pair).
// K and V tiles of 64 keys are double-buffered in LDS: tile kt+1 travels global -> registers while tile kt is multiplied and
// lands in the other buffer behind ONE barrier per tile.
// Layouts (bf16): Q [B,h,Nq,DH] (pre-scaled), K, V [B,h,N,DH]; out [B,Nq,h*DH]; lse fp32 [B,h,Nq] (nullable).
// Query i sits at sequence position i (causal: keys <= i).
// ------------------------------------------------------------------------------------------------------
typedef short v4s_t __attribute__((ext_vector_type(4)));

// 16-byte LDS-DMA piece as inline asm: lane l's 16 bytes at src land at LDS byte address lds_addr + 16 l (lds_addr wave-uniform).
// The compiler does not see the transfer, so it does not put a conservative s_waitcnt vmcnt(0) in front of the first
// ds_read_b64_tr_b16 that follows (it cannot tell the read's buffer from the one being filled); the kernel's own
// vmcnt(0) + barrier at the end of every tile orders the pieces before any read of them.
__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_addr) : "memory", "m0");
}

// reductions across the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48) on the VALU: v_permlane16_swap /
// v_permlane32_swap hand every lane its own value and its partner row's, no LDS crossbar round trip (ds_bpermute) in the chain
__device__ __forceinline__ float rows_max(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float v) {
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// OT: 16-bit operand type tag (bf16_t / f16_t, common.h): Q, K, V, the probabilities P and the output share it
template <int DH, int QG, int NW, typename OT = bf16_t>
__global__ __launch_bounds__(64 * NW, 1) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void flash_attention_v2_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, bf16_t* __restrict__ out,
    int Nq, int N, int heads, int causal, float* __restrict__ lse, unsigned drop_thresh, float drop_scale,
    unsigned long long drop_seed) {
  // drop_thresh != 0: training-mode nn.Dropout on the attention probabilities (Attention.py:17,36): a stateless hash mask of
  // (seed, (bh * Nq + query) * N + key) multiplies P where it enters P.V; the softmax denominator sums the undropped P
  constexpr int KS = DH / 32, DT = DH / 16;
  constexpr int CPR = DH / 8;                         // 16-byte chunks per row
  constexpr int SW = CPR < 16 ? CPR : 16;             // XOR swizzle period: chunk position = chunk ^ (row & (SW - 1))
  constexpr int RPI = 64 / CPR;                       // rows per LDS-DMA wave-instruction (1 KiB)
  constexpr int NI = AK / RPI / NW;                   // DMA instructions per wave, tile and tensor
  constexpr int QB = 16 * QG * NW;                    // queries per workgroup
  constexpr int TILE = AK * DH;                       // elements of one tile image (unpadded rows: the swizzle spreads the banks)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* sK = (bf16_t*)smem;                         // [2][AK][DH]
  bf16_t* sV = sK + 2 * TILE;                         // [2][AK][DH]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  // causal: the heaviest query blocks (most key tiles) are dispatched first ACROSS the whole grid: workgroups are handed to the
  // 8 XCDs round-robin in linear order, so "x = query block" would give one XCD all the 16-tile blocks and another all the
  // 2-tile ones; with the linear id split as (query block, bh) every XCD gets the same mix and the blocks of one (batch, head)
  // still meet on one XCD's L2 whenever B * heads is a multiple of 8
  const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y, n_bh = (int)gridDim.y;
  const int bh = causal ? lin % n_bh : (int)blockIdx.y, b = bh / heads, hd = bh % heads;
  const int qb = causal ? (int)gridDim.x - 1 - lin / n_bh : (int)blockIdx.x;
  const int q0 = qb * QB;
  const bf16_t* Qb = Q + (size_t)bh * Nq * DH;
  const bf16_t* Kb = K + (size_t)bh * N * DH;
  const bf16_t* Vb = V + (size_t)bh * N * DH;

  int own[QG];
  bf16x8 qf[QG][KS];
#pragma unroll
  for (int u = 0; u < QG; ++u) {
    own[u] = q0 + (wave * QG + u) * 16 + l15;
    const int ld = own[u] < Nq ? own[u] : Nq - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[u][ks] = *(const bf16x8*)(Qb + (size_t)ld * DH + ks * 32 + 8 * g);
  }
  f32x4 o[QG][DT];
  float m_run[QG], l_run[QG];
#pragma unroll
  for (int u = 0; u < QG; ++u) {
    m_run[u] = -INFINITY; l_run[u] = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d) o[u][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int q_hi = (q0 + QB - 1 < Nq - 1) ? q0 + QB - 1 : Nq - 1;
  const int n_tiles = causal ? ((q_hi < N - 1 ? q_hi : N - 1) / AK + 1) : (N + AK - 1) / AK;

  // HBM -> LDS by 16-byte LDS-DMA, no staging registers: lane -> (row of the instruction, chunk position); the SOURCE chunk is
  // the swizzled one, so the image is lane-linear (what the DMA writes) and the reads apply the same XOR
  const int d_row = lane / CPR, d_pos = lane % CPR;
  // the V image has its own key: a 32-lane half of a transposed read takes 8 consecutive rows x two adjacent chunks, so the
  // key is EVEN and distinct over 8 rows - 2 (row & 7) - and the half covers all 16 bank slots (row & 15 as the key pairs rows
  // r and r ^ 1 on the same two slots: 2-way, measured as one third of all LDS cycles)
  auto vkey = [](int r) { return CPR >= 16 ? 2 * (r & 7) : (r & (SW - 1)); };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r0 = (wave * NI + i) * RPI;                                  // wave-uniform first row of this instruction
      const int r = r0 + d_row;
      int kr = kt * AK + r; if (kr > N - 1) kr = N - 1;                      // clamped; masked below
      const size_t src = (size_t)kr * DH + ((d_pos ^ (r & (SW - 1))) << 3);
      const size_t srcv = (size_t)kr * DH + ((d_pos ^ vkey(r)) << 3);
      lds_dma16(Kb + src, lds0 + (unsigned)(buf * TILE + r0 * DH) * 2u);
      lds_dma16(Vb + srcv, lds0 + (unsigned)((2 + buf) * TILE + r0 * DH) * 2u);
    }
  };
  dma(0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);             /* vmcnt(0) */
  __syncthreads();                                                           // tile 0 landed
  const int qp = l15 >> 2, pp = l15 & 3;
  // per-lane LDS element offsets, computed ONCE: everything that varies inside the tile loop (key sub-tile t, buffer, k-step
  // half, +16 rows) is a compile-time constant that folds into the ds_read offset field
  //   row fragments of K: row 16 t + l15, chunk (4 ks + g) ^ (l15 & (SW-1))            -> kbase[ks] + t * 16 * DH
  //   transposed reads of V: row 32 s + 4 g + qp (+16), chunk (2 dt + (pp >> 1)) ^ vkey(row) -> vbase[dt & 7] + (dt >> 3) * 128 + ...
  int kbase[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kbase[ks] = l15 * DH + (((ks * 4 + g) ^ (l15 & (SW - 1))) << 3);
  constexpr int NVB = DT < 8 ? DT : 8;
  int vbase[NVB];
  const int r15 = vkey(4 * g + qp);
#pragma unroll
  for (int d = 0; d < NVB; ++d) vbase[d] = (4 * g + qp) * DH + (((2 * d + (pp >> 1)) ^ r15) << 3) + 4 * (pp & 1);
  for (int kt = 0; kt < n_tiles; ++kt) {
    const int buf = kt & 1, k0 = kt * AK;
    if (kt + 1 < n_tiles) dma(buf ^ 1, kt + 1);                              // lands under this tile's MFMAs
    const bf16_t* cK = sK + buf * TILE;
    const bf16_t* cV = sV + buf * TILE;
    // causal: a tile that starts past the wave's last query is fully masked for this wave (the upper waves of the block still
    // need it): no MFMAs, only the staging and the barrier
    if (!causal || k0 <= q0 + (wave + 1) * QG * 16 - 1) {
    // S^T = K_tile . Q^T: rows = key (16 t + 4 g + e), column = query (lane & 15); one K fragment feeds QG MFMAs
    f32x4 st[QG][4];
#pragma unroll
    for (int u = 0; u < QG; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) st[u][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K fragments travel LDS -> registers in groups of 8 through a two-slot ring, two groups ahead of the MFMAs that eat them
    // (the compiler left to itself re-uses ONE fragment register: read, wait, MFMA, 32 exposed LDS latencies per tile)
    constexpr int NGK = KS * 4 / 8 > 0 ? KS * 4 / 8 : 1, GK = KS * 4 / NGK;
    bf16x8 kf[2][GK];
#pragma unroll
    for (int gi = 0; gi < 2 && gi < NGK; ++gi)
#pragma unroll
      for (int j = 0; j < GK; ++j) kf[gi][j] = *(const bf16x8*)(cK + kbase[(gi * GK + j) >> 2] + ((gi * GK + j) & 3) * 16 * DH);
#pragma unroll
    for (int gi = 0; gi < NGK; ++gi) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < GK; ++j) {
        const int ks = (gi * GK + j) >> 2, t = (gi * GK + j) & 3;
#pragma unroll
        for (int u = 0; u < QG; ++u) st[u][t] = op16<OT>::mfma(kf[gi & 1][j], qf[u][ks], st[u][t]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (gi + 2 < NGK) {
#pragma unroll
        for (int j = 0; j < GK; ++j)
          kf[gi & 1][j] = *(const bf16x8*)(cK + kbase[((gi + 2) * GK + j) >> 2] + (((gi + 2) * GK + j) & 3) * 16 * DH);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // online softmax, per lane = per query
    bf16x8 pf[QG][2];
    // only the tile that crosses the end of the sequence and (causal) the tiles that reach past the wave's first query need the
    // mask; everywhere else the 32 compare/select pairs per query group are skipped (wave-uniform branch)
    const bool need_mask = k0 + AK > N || (causal && k0 + AK - 1 > q0 + wave * QG * 16);
#pragma unroll
    for (int u = 0; u < QG; ++u) {
      float mx = -INFINITY;
      if (need_mask) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int key = k0 + t * 16 + 4 * g + e;
            const bool dead = key >= N || (causal && key > own[u]);
            st[u][t][e] = dead ? -INFINITY : st[u][t][e];
          }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) mx = fmaxf(mx, fmaxf(fmaxf(st[u][t][0], st[u][t][1]), fmaxf(st[u][t][2], st[u][t][3])));
      mx = rows_max(mx);
      // deferred rescale: the running reference maximum only moves when the tile's maximum exceeds it by more than 8 (for some
      // query of the wave), so exp(s - m) stays below e^8 and the O / l rescale (128 accumulator registers through the VALU) is
      // rare instead of once per tile; the result is the same softmax (any reference point cancels in O / l)
      const bool bump = __any(mx > m_run[u] + 8.0f) || m_run[u] == -INFINITY;
      const float m_new = bump ? fmaxf(m_run[u], mx) : m_run[u];
      const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = bump ? __expf(m_run[u] - msafe) : 1.0f;
      float rs = 0.f;
      unsigned pw[2][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { p[e] = __expf(st[u][t][e] - msafe); rs += p[e]; }
        if (drop_thresh) {
          const size_t rowi = ((size_t)bh * Nq + (own[u] < Nq ? own[u] : Nq - 1)) * N + k0 + t * 16 + 4 * g;
#pragma unroll
          for (int e = 0; e < 4; ++e) p[e] = dropout_keep_(drop_seed, rowi + e, drop_thresh) ? p[e] : 0.f;
        }
        pw[t >> 1][2 * (t & 1)] = op16<OT>::pack2(p[0], p[1]);
        pw[t >> 1][2 * (t & 1) + 1] = op16<OT>::pack2(p[2], p[3]);
      }
      rs = rows_sum(rs);
      l_run[u] = l_run[u] * alpha + rs;
      m_run[u] = m_new;
      if (bump) {                                                            // wave-uniform
#pragma unroll
        for (int d = 0; d < DT; ++d) { o[u][d][0] *= alpha; o[u][d][1] *= alpha; o[u][d][2] *= alpha; o[u][d][3] *= alpha; }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) pf[u][s2] = __builtin_bit_cast(bf16x8, (u32x4){pw[s2][0], pw[s2][1], pw[s2][2], pw[s2][3]});
    }
    // O^T += V_tile^T . P^T: A = V read transposed (rows d = 16 dt + lane & 15, k = keys {32 s + 4 g + 0..3, 32 s + 16 + 4 g + 0..3});
    // the lane that supplies the address of block row q', columns 4 p .. 4 p + 3 is lane 4 q' + p of its 16-lane group
    // same two-slot ring for the V fragments (fragment f = s2 * DT + dt; rows 32 s2 + 4 g + qp and + 16, same swizzle key: both
    // = 4 g + qp mod 16; dt >= 8 is 16 chunks = 128 elements further)
    constexpr int NGV = 2 * DT / 8, GV = 8;
    bf16x8 vf[2][GV];
    auto ldv = [&](int f) -> bf16x8 {
      const int s2 = f / DT, dt = f % DT;
      const int off = vbase[dt & (NVB - 1)] + (dt >= NVB ? 128 : 0) + 32 * s2 * DH;
      typedef __attribute__((address_space(3))) v4s_t* lds_v4s;
      const v4s_t ta = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(cV + off));
      const v4s_t tb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(cV + off + 16 * DH));
      typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
      const u32x2_t ua = __builtin_bit_cast(u32x2_t, ta), ub = __builtin_bit_cast(u32x2_t, tb);      // register concatenation, no ALU
      return __builtin_bit_cast(bf16x8, (u32x4){ua[0], ua[1], ub[0], ub[1]});
    };
#pragma unroll
    for (int gi = 0; gi < 2 && gi < NGV; ++gi)
#pragma unroll
      for (int j = 0; j < GV; ++j) vf[gi][j] = ldv(gi * GV + j);
#pragma unroll
    for (int gi = 0; gi < NGV; ++gi) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < GV; ++j) {
        const int s2 = (gi * GV + j) / DT, dt = (gi * GV + j) % DT;
#pragma unroll
        for (int u = 0; u < QG; ++u) o[u][dt] = op16<OT>::mfma(vf[gi & 1][j], pf[u][s2], o[u][dt]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (gi + 2 < NGV) {
#pragma unroll
        for (int j = 0; j < GV; ++j) vf[gi & 1][j] = ldv((gi + 2) * GV + j);
      }
    }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);             /* vmcnt(0) */   // this wave's pieces of the next tile have landed ...
    __syncthreads();                                   // ... everyone's have, and everyone is done reading this one
  }
  // epilogue: out[b, q, head*DH + d] = O^T[d][q] / l ; a lane holds d = 16 dt + 4 g .. + 3 (8 bytes) per dt.  Lanes g and g ^ 1 trade
  // halves of two dt's (v_permlane16_swap) so every lane stores 16 contiguous bytes: 8 stores per query instead of 16 (the
  // store tail is issue-bound).  The swaps run on all lanes; only the stores are masked.
#pragma unroll
  for (int u = 0; u < QG; ++u) {
    const bool live = own[u] < Nq;
    const float inv = drop_scale / l_run[u];                                 // 1 / (1 - p) of the probability dropout (1 without it)
    if (live && lse != nullptr && g == 0) lse[(size_t)bh * Nq + own[u]] = m_run[u] + logf(l_run[u]);
    bf16_t* orow = out + ((size_t)b * Nq + (live ? own[u] : 0)) * heads * DH + (size_t)hd * DH;
#pragma unroll
    for (int dt = 0; dt < DT; dt += 2) {
      const unsigned a0 = op16<OT>::pack2_sat(o[u][dt][0] * inv, o[u][dt][1] * inv), a1 = op16<OT>::pack2_sat(o[u][dt][2] * inv, o[u][dt][3] * inv);
      const unsigned b0 = op16<OT>::pack2_sat(o[u][dt + 1][0] * inv, o[u][dt + 1][1] * inv), b1 = op16<OT>::pack2_sat(o[u][dt + 1][2] * inv, o[u][dt + 1][3] * inv);
      // odd 16-lane rows of (a0, a1) <-> even rows of (b0, b1): g even keeps a (its own d's of dt) and receives g + 1's;
      // g odd ends with both halves of dt + 1
      const auto x = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
      const auto y = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
      if (live) *(uint4*)(orow + (dt + (g & 1)) * 16 + 8 * (g >> 1)) = make_uint4(x[0], y[0], x[1], y[1]);
    }
  }
}

// V row-major [B,h,N,dh]; Nq queries per (batch, head) at positions 0..Nq-1 (Nq == N for self-attention; Nq = 1: only token 0,
// the last encoder layer of ViTEnc whose output is read at token 0 only, ViT.py:136)
int launch_flash_attention_v2(const void* Q, const void* K, const void* V, void* out, int B, int Nq, int N, int heads, int dh,
                              int causal, hipStream_t s, float* lse, unsigned drop_thresh, float drop_scale,
                              unsigned long long drop_seed, bool f16) {
  if (!drop_thresh) drop_scale = 1.f;
  // long sequences: 8 waves x 16 queries (two waves per SIMD: one wave's softmax and LDS waits hide under the other's MFMAs);
  // 129-token windows: 4 waves x 16 queries = 3 x 64 query slots instead of 2 x 128
  static const int nw_env = prego_tune_env("PREGO_ATTN_NW") ? atoi(prego_tune_env("PREGO_ATTN_NW")) : 0;      // A/B knob: 4 = round-2 shape (4 waves x 32 queries)
  const bool longq = Nq > 192;
  const int nw = nw_env == 4 ? 4 : (nw_env == 8 ? 8 : (longq ? 8 : 4));
  const int qg = longq && nw == 4 ? 2 : 1;
  const int qb = 16 * qg * nw;
  dim3 grid((Nq + qb - 1) / qb, B * heads);
#define FA2(D, G, W)                                                                                             \
  do {                                                                                                           \
    const size_t lds = (size_t)4 * AK * D * 2;                                                                   \
    static DeviceOnce once;                                                                                      \
    once.run([&] { (void)hipFuncSetAttribute((const void*)flash_attention_v2_kernel<D, G, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
                   (void)hipFuncSetAttribute((const void*)flash_attention_v2_kernel<D, G, W, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }); \
    if (f16) flash_attention_v2_kernel<D, G, W, f16_t><<<grid, 64 * W, lds, s>>>((const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)out, Nq, N, \
                                                                 heads, causal, lse, drop_thresh, drop_scale, drop_seed);  \
    else flash_attention_v2_kernel<D, G, W><<<grid, 64 * W, lds, s>>>((const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)out, Nq, N, \
                                                                 heads, causal, lse, drop_thresh, drop_scale, drop_seed);  \
  } while (0)
#define FA2D(D) do { if (nw == 8) FA2(D, 1, 8); else if (qg == 2) FA2(D, 2, 4); else FA2(D, 1, 4); } while (0)
  if (dh == 256) FA2D(256);
  else if (dh == 128) FA2D(128);
  else if (dh == 64) FA2D(64);
  else return -1;
#undef FA2D
#undef FA2
  return 0;
}
