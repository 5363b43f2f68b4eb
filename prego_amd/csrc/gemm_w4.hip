// 16-bit NT GEMM, 256 x 256 x 64 tiles, FOUR waves with 128 x 128 wave tiles (round 5 experiment).
//   C[M,N] = A[M,K] . B[N,K]^T + bias        (the layer1 / W_ih projections, model/rnn/rnn.py:38-42,61)
// The production kernel (gemm_pp.hip: 8 waves, 128 x 64 wave tiles, ping-pong) moves 192 KB of fragments + 64 KB of LDS-DMA through LDS per
// K tile = 2 048 cycles at 128 B/clk - exactly the issue time of its MFMAs: LDS and the matrix pipe are co-limiters.  A 128 x 128 wave tile
// reads 16 + 16 fragments for 128 MFMAs instead of 16 + 8 for 64: 128 KB + 64 KB per K tile = 1 536 cycles of LDS for 2 048 of MFMA.  The price
// is ONE wave per SIMD (256 accumulator registers): no partner wave hides the fragment reads and the DMA issue, they have to sit between this
// wave's own MFMAs.
// LDS: two stages of [A 256 rows x 64 k | B 256 rows x 64 k] (64 KB each), images = lane-linear LDS-DMA pieces of 8 rows x 128 B with the
// 16-byte-chunk XOR swizzle of gemm_pp.hip on the source address and on the reads (same keys, same permuted B rows, same transposed products:
// a lane ends up with 8 consecutive columns of a row).
#include "common.h"
#include "kernels.h"

#define WBM 256
#define WBN 256
#define WBK 64
#define WSTAGE 65536

__device__ __forceinline__ int w4_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
__device__ __forceinline__ int w4_key_b(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }

// DIAG (timing-only builds behind prego_debug_gemm_bf16 variants 21-23, wrong results): 1 = no LDS-DMA in the loop, 2 = no fragment reads in
// the loop, 3 = neither
template <bool OUT16, typename OT, int DIAG = 0>
__global__ __launch_bounds__(256, 1) void gemm_bf16_nt_w4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                                 const float* __restrict__ bias, void* __restrict__ Cv, int M, int N, int K,
                                                                 int lda, int ldb, int ldc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntn = N / WBN, ntm = (M + WBM - 1) / WBM;
  const int tile = w4_xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * WBM, n0 = (tile % ntn) * WBN;
  const int nk = K / WBK;

  // ---- LDS-DMA sources: wave w stages rows 64 w .. 64 w + 63 of both tiles, 8 pieces of 8 rows each
  const int sr = lane >> 3, scp = lane & 7;
  int a_off[8], b_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = (wave * 8 + i) * 8 + sr;
    a_off[i] = r * lda * 2 + ((scp ^ ((r >> 1) & 7)) << 4);
    b_off[i] = r * ldb * 2 + ((scp ^ w4_key_b(r)) << 4);
  }
  const int rows = M - m0 < WBM ? M - m0 : WBM;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, rows * lda * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)n0 * ldb), 0, WBN * ldb * 2, 0x00020000);
  auto stage = [&](int kt) {
    char* da = smem + (kt & 1) * WSTAGE + wave * 8192;
    char* db = da + 32768;
    const int so = kt * (WBK * 2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(da + i * 1024), 16, a_off[i], so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(db + i * 1024), 16, b_off[i], so, 0, 0);
    }
  };
  // ---- fragment addresses: two lane constants per operand (k-step 0 / 1), everything else is an immediate
  const int fr = lane & 15, fq = lane >> 4;
  int a_rd[2], b_rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_rd[ks] = (wr * 128 + fr) * 128 + (((ks * 4 + fq) ^ ((fr >> 1) & 7)) << 4);
    const int rb = wc * 128 + (fr >> 2) * 8 + (fr & 3);
    b_rd[ks] = 32768 + rb * 128 + (((ks * 4 + fq) ^ w4_key_b(rb)) << 4);
  }
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- software pipeline (one wave per SIMD: everything that is not an MFMA sits BETWEEN this wave's MFMAs, one memory instruction per
  // four of them).  K tile kt lives in stage kt & 1.  Per K tile:
  //   phase A: 64 MFMAs on the k-step-0 fragments (read during the previous tile's phase B) | the 16 reads of this tile's k-step-1 fragments
  //   sync:    lgkmcnt(0) - my reads of this stage are done -, vmcnt(0) - my pieces of tile kt + 1 have landed -, ONE barrier
  //   phase B: 64 MFMAs on the k-step-1 fragments | the 16 LDS-DMA pieces of tile kt + 2 into the stage just freed, and the 16 reads of tile
  //            kt + 1's k-step-0 fragments
  // The last two tiles re-stage tile nk - 1 into a stage nobody reads again and read fragments nobody uses (branch-free loop body).
  bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];
  auto rd_a = [&](const char* base, int ks, int i) -> bf16x8 { return *(const bf16x8*)(base + a_rd[ks] + i * 2048); };
  auto rd_b = [&](const char* base, int ks, int j) -> bf16x8 { return *(const bf16x8*)(base + b_rd[ks] + (j >> 1) * 4096 + (j & 1) * 512); };
  auto dma_piece = [&](int kt, int p) {          // piece p = 0 .. 15 of this wave's share of tile kt: A pieces 0-7, B pieces 8-15
    char* d = smem + (kt & 1) * WSTAGE + wave * 8192 + (p >> 3) * 32768 + (p & 7) * 1024;
    const int kk = kt < nk ? kt : nk - 1;
    if (p < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)d, 16, a_off[p & 7], kk * (WBK * 2), 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)d, 16, b_off[p & 7], kk * (WBK * 2), 0, 0);
  };
#define W4_FENCE() __builtin_amdgcn_sched_barrier(0)
  stage(0);
  stage(1 < nk ? 1 : 0);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");            // tile 0 has landed (the 16 pieces of tile 1 may still be in flight)
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) { fa0[i] = rd_a(smem, 0, i); fb0[i] = rd_b(smem, 0, i); fa1[i] = rd_a(smem, 1, i); fb1[i] = rd_b(smem, 1, i); }
  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = smem + (kt & 1) * WSTAGE;
    const char* nxt = smem + ((kt + 1) & 1) * WSTAGE;
    // ---- phase A
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      W4_FENCE();
      if constexpr (!(DIAG & 2)) { if (g < 8) fa1[g] = rd_a(cur, 1, g); else fb1[g - 8] = rd_b(cur, 1, g - 8); }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = (g * 4 + e) >> 3, j = (g * 4 + e) & 7;
        acc[i][j] = op16<OT>::mfma(fb0[j], fa0[i], acc[i][j]);
      }
    }
    W4_FENCE();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    W4_FENCE();
    // ---- phase B
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      W4_FENCE();
      if constexpr (!(DIAG & 1)) dma_piece(kt + 2, g);
      if constexpr (!(DIAG & 2)) { if (g < 8) fa0[g] = rd_a(nxt, 0, g); else fb0[g - 8] = rd_b(nxt, 0, g - 8); }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = (g * 4 + e) >> 3, j = (g * 4 + e) & 7;
        acc[i][j] = op16<OT>::mfma(fb1[j], fa1[i], acc[i][j]);
      }
    }
    W4_FENCE();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the re-staged pieces of the last two tiles
#undef W4_FENCE
  // ---- epilogue: lane (fr, fq) of accumulator pair (i, jp) holds row m0 + wr 128 + 16 i + fr, columns n0 + wc 128 + 32 jp + 8 fq .. + 7
#pragma unroll
  for (int jp = 0; jp < 4; ++jp) {
    const int n = n0 + wc * 128 + jp * 32 + fq * 8;
    const float4 bv0 = bias ? *(const float4*)(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bv1 = bias ? *(const float4*)(bias + n + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wr * 128 + i * 16 + fr;
      if (m >= M) continue;
      const f32x4 v0 = acc[i][2 * jp], v1 = acc[i][2 * jp + 1];
      if constexpr (OUT16) {
        u32x4 pk;
        pk[0] = op16<OT>::pack2_sat(v0[0] + bv0.x, v0[1] + bv0.y); pk[1] = op16<OT>::pack2_sat(v0[2] + bv0.z, v0[3] + bv0.w);
        pk[2] = op16<OT>::pack2_sat(v1[0] + bv1.x, v1[1] + bv1.y); pk[3] = op16<OT>::pack2_sat(v1[2] + bv1.z, v1[3] + bv1.w);
        *(u32x4*)((bf16_t*)Cv + (size_t)m * ldc + n) = pk;
      } else {
        float* c = (float*)Cv + (size_t)m * ldc + n;
        *(float4*)c = make_float4(v0[0] + bv0.x, v0[1] + bv0.y, v0[2] + bv0.z, v0[3] + bv0.w);
        *(float4*)(c + 4) = make_float4(v1[0] + bv1.x, v1[1] + bv1.y, v1[2] + bv1.z, v1[3] + bv1.w);
      }
    }
  }
}

// 0 on success, -1: shape not supported (N % 256, K % 64, K >= 128)
void launch_gemm_bf16_w4_diag(int diag, const void* A, int lda, const void* B, int ldb, const float* bias, void* C, int ldc, int M, int N, int K, hipStream_t s) {
  const int ntm = (M + WBM - 1) / WBM, ntn = N / WBN;
  const size_t lds = 2 * WSTAGE;
  const bf16_t* a = (const bf16_t*)A; const bf16_t* b = (const bf16_t*)B;
  (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_w4_kernel<false, bf16_t, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_w4_kernel<false, bf16_t, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_w4_kernel<false, bf16_t, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (diag == 1) gemm_bf16_nt_w4_kernel<false, bf16_t, 1><<<ntm * ntn, 256, lds, s>>>(a, b, bias, C, M, N, K, lda, ldb, ldc);
  else if (diag == 2) gemm_bf16_nt_w4_kernel<false, bf16_t, 2><<<ntm * ntn, 256, lds, s>>>(a, b, bias, C, M, N, K, lda, ldb, ldc);
  else gemm_bf16_nt_w4_kernel<false, bf16_t, 3><<<ntm * ntn, 256, lds, s>>>(a, b, bias, C, M, N, K, lda, ldb, ldc);
}
int launch_gemm_bf16_w4(const void* A, int lda, const void* B, int ldb, const float* bias, void* C, int ldc, int M, int N, int K, bool out16,
                        hipStream_t s, bool f16) {
  if (M <= 0 || N % WBN || K % WBK || K < 2 * WBK) return -1;
  const int ntm = (M + WBM - 1) / WBM, ntn = N / WBN;
  const size_t lds = 2 * WSTAGE;
  static DeviceOnce once;
  once.run([&] {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_w4_kernel<false, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_w4_kernel<true, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_w4_kernel<true, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  const bf16_t* a = (const bf16_t*)A; const bf16_t* b = (const bf16_t*)B;
  if (out16 && f16) gemm_bf16_nt_w4_kernel<true, f16_t><<<ntm * ntn, 256, lds, s>>>(a, b, bias, C, M, N, K, lda, ldb, ldc);
  else if (out16) gemm_bf16_nt_w4_kernel<true, bf16_t><<<ntm * ntn, 256, lds, s>>>(a, b, bias, C, M, N, K, lda, ldb, ldc);
  else gemm_bf16_nt_w4_kernel<false, bf16_t><<<ntm * ntn, 256, lds, s>>>(a, b, bias, C, M, N, K, lda, ldb, ldc);
  return 0;
}
