// Training-step GEMMs whose operands are stored with the CONTRACTION index as the row index (trainer/train.py:23, loss.backward()):
//   wgrad  dW[Nout, Kin] = dY^T . X     dY [rows, Nout], X [rows, Kin]: both "k-major"            (TA = true,  TB = true)
//   dgrad  dX[rows, Kin] = dY . W       dY [rows, Nout] row-major, W [Nout, Kin] as [K][N]         (TA = false, TB = true)
// Round 3 brought every such operand into nn.Linear's NT form first (10 transpose_convert launches per step) and summed the bias
// gradients in separate two-stage column-sum launches.  Here a k-major operand is staged into LDS as it lies in memory ([64 k][128
// columns], 256-byte rows, LDS-DMA with the 16-byte-chunk XOR swizzle of cdna_hip_programming.md T10 image (b) on the source
// address) and its MFMA fragments are read TRANSPOSED by ds_read_b64_tr_b16: no transposed tensor exists anywhere.  The bias
// gradient db[m] = sum_k dY[k][m] rides along as one more MFMA per k-step against a fragment of ones (COLSUM: column 0 of the
// extra accumulator), written by the n-tile-0 workgroups.
// 128 x 128 x 64 tiles, 4 waves of 64 x 64 (16x16x32 bf16 MFMA, fp32 accumulate), two LDS buffers, one barrier per K tile - the
// structure of gemm.hip's 128-square kernel; these GEMMs are train.py-sized (rows = 16 x 128), 0.3 ms in all.
// Round 5: (1) the transposed fragment reads are inline asm - behind the compiler's own ds_read_b64_tr_b16 every K tile began with
// s_waitcnt vmcnt(0) (comment at tr_issue): load and multiply never overlapped; (2) launches of at most one tile per CU run the
// eight-wave split-K workgroup (comment at the template).  wgrad 53 -> 36 us, dgrad 40 -> 30 us, training step 1.30 -> 1.21 ms.
// Measured and not kept: a four-stage 128 x 128 x 32 form of the wgrad (LDS-DMA two K tiles ahead behind a counted vmcnt, one raw
// barrier per 16 MFMAs) - 1.329 against 1.286 ms per training step: the barrier per 32-wide K tile costs more than the exposed load
// latency it removes (two workgroups per CU already cover for each other).
#include "common.h"
#include "kernels.h"

#define TBM 128
#define TBN 128
#define TBK 64

typedef short tn_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tn_v4s* tn_lds_v4s;

__device__ __forceinline__ int tn_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}
// image (b): byte offset of 16-byte chunk ch (0..15) of row `row` of a [rows][128 x 16-bit] tile
__device__ __forceinline__ int tn_key(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// C[M,N] fp32 = op(A) . op(B) (+ bias[n]);  TA: A stored [K][M] (lda = elements per k row) else [M][K];  TB: B stored [K][N] else [N][K].
// Rows / columns past the logical extents read as zeros (buffer resources), M and N need not be tile multiples; K % 64 == 0.
// SPLITK (round 5): launches of at most one workgroup per CU - every training-step shape: rows = 16 x 128 make 8 ... 256 tiles - run
// EIGHT waves: waves 0-3 multiply the even K tiles, waves 4-7 the odd ones, each group with its own two LDS buffers, and the second
// group's accumulators are added through LDS at the end.  With one wave per SIMD the loop was bound by what ONE instruction stream has
// to issue per K tile (8 LDS-DMA pieces at ~100+ cycles each, 32 transposed reads, 32 MFMAs: 1.25 us even with every load latency
// hidden - a four-buffer form measured 40 us per wgrad against 53); two waves per SIMD issue their memory instructions under each
// other's MFMAs.
template <bool TA, bool TB, bool COLSUM, bool SPLITK = false>
__global__ __launch_bounds__(SPLITK ? 512 : 256, SPLITK ? 1 : 2) void gemm_bf16_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                              const float* __restrict__ bias, float* __restrict__ C,
                                                              float* __restrict__ colsum_out, int M, int N, int K, int k_valid,
                                                              int lda, int ldb, int ldc, bf16_t* __restrict__ C16 /* nullable: bf16 C instead */) {
  extern __shared__ __attribute__((aligned(16))) char smem_all[];   // per K group: [2][A 16 KB | B 16 KB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);
  const int grp = SPLITK ? __builtin_amdgcn_readfirstlane(tid >> 8) : 0;       // K group: tiles kt = grp (mod 2)
  char* smem = smem_all + grp * 65536;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (N + TBN - 1) / TBN, ntm = (M + TBM - 1) / TBM;
  const int tile = tn_xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * TBM, n0 = (tile % ntn) * TBN;

  // ---- LDS-DMA sources.  Row-major operand ([rows of the tile][64 k], 128-byte rows): 8 rows x 128 B per piece, chunk XOR (r >> 1) & 7.
  // k-major operand ([64 k][128 columns], 256-byte rows): 4 rows x 256 B per piece, chunk XOR tn_key(row).  Out-of-range rows /
  // columns fall outside the buffer resource and read as zeros.
  __amdgpu_buffer_rsrc_t rs_a, rs_b;
  int a_off[4], b_off[4];
  if constexpr (TA) {
    // rows = k (valid: k_valid), columns m0 .. m0 + 127 (valid: < M).  num_records bounds the ROWS; columns are clamped per lane.
    rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)k_valid * lda * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 4 + (lane >> 4);                // k row inside the K tile, 0..63
      const int ch = (lane & 15) ^ tn_key(row);                        // source chunk that lands at LDS position lane & 15
      // columns >= M only feed output rows that are never stored (an MFMA row depends on its own A row alone): no clamp needed,
      // and reads past the end of the array fall outside num_records (zeros, no fault)
      a_off[i] = (row * lda + m0 + ch * 8) * 2;
    }
  } else {
    const int rows = M - m0 < TBM ? M - m0 : TBM;
    rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, (unsigned)((size_t)rows * lda * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wave * 4 + i) * 8 + (lane >> 3);
      a_off[i] = (r * lda + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2;
    }
  }
  if constexpr (TB) {
    rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)((size_t)k_valid * ldb * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 4 + (lane >> 4);
      const int ch = (lane & 15) ^ tn_key(row);
      b_off[i] = (row * ldb + n0 + ch * 8) * 2;
    }
  } else {
    const int rows = N - n0 < TBN ? N - n0 : TBN;
    rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)n0 * ldb), 0, (unsigned)((size_t)rows * ldb * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wave * 4 + i) * 8 + (lane >> 3);
      b_off[i] = (r * ldb + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2;
    }
  }
  auto stage = [&](int buf, int kt) {
    char* la = smem + buf * 32768;
    char* lb = la + 16384;
    const int so_a = TA ? kt * TBK * lda * 2 : kt * TBK * 2;
    const int so_b = TB ? kt * TBK * ldb * 2 : kt * TBK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(la + (wave * 4 + i) * 1024), 16, a_off[i], so_a, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(lb + (wave * 4 + i) * 1024), 16, b_off[i], so_b, 0, 0);
    }
  };

  f32x4 acc[4][4];
  f32x4 cs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    cs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int fr = lane & 15, fq = lane >> 4;
  const bool do_cs = COLSUM && n0 == 0 && wn == 0;
  const bf16x8 ones = __builtin_bit_cast(bf16x8, (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u});
  // transposed fragment of the 16-column block cb at k-step ks of a k-major image: two 4-row x 16-column blocks (rows 8 g + 4 h + q)
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  // The transposed reads are issued as inline asm: behind the compiler's own ds_read_b64_tr_b16 (the builtin carries no address the
  // alias analysis could tell from the LDS-DMA destinations) every K tile began with s_waitcnt vmcnt(0) - the NEXT tile's pieces, just
  // issued, had to land before this tile's first fragment was read: load and multiply ran strictly one after the other (rounds 3-4:
  // 1.65 us per K tile whatever the tile held).  The asm reads of a k-step are followed by ONE s_waitcnt lgkmcnt(0) that names
  // their registers as in/out operands, so no MFMA that uses them can move above it.
  auto tr_issue = [&](const char* img, int ks, int cb, tn_v4s& ha, tn_v4s& hb) {
    const int r0 = ks * 32 + 8 * fq + tq;
    const int ch = 2 * cb + (tp >> 1);
    const int oa = 256 * r0 + 16 * (ch ^ tn_key(r0)) + 8 * (tp & 1);
    const int r1 = r0 + 4;
    const int ob = 256 * r1 + 16 * (ch ^ tn_key(r1)) + 8 * (tp & 1);
    const unsigned aa = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)(img + oa);
    const unsigned ab = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)(img + ob);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ha) : "v"(aa));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hb) : "v"(ab));
  };
  auto tr_join = [&](tn_v4s ha, tn_v4s hb) -> bf16x8 {
    const u32x2 ua = __builtin_bit_cast(u32x2, ha), ub = __builtin_bit_cast(u32x2, hb);
    return __builtin_bit_cast(bf16x8, (u32x4){ua[0], ua[1], ub[0], ub[1]});
  };
  auto row_frag = [&](const char* img, int ks, int r) -> bf16x8 {
    return *(const bf16x8*)(img + r * 128 + (((ks * 4 + fq) ^ ((r >> 1) & 7)) << 4));
  };
  auto compute = [&](int buf) {
    const char* la = smem + buf * 32768;
    const char* lb = la + 16384;
    tn_v4s ha[2][4][2], hb[2][4][2];
    bf16x8 af[4], bfr[4];
    auto issue = [&](int ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (TA) tr_issue(la, ks, wm * 4 + i, ha[ks][i][0], ha[ks][i][1]);
        if constexpr (TB) tr_issue(lb, ks, wn * 4 + i, hb[ks][i][0], hb[ks][i][1]);
      }
    };
    auto landed = [&](int ks) {
      if constexpr (TA && TB)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ha[ks][0][0]), "+v"(ha[ks][0][1]), "+v"(ha[ks][1][0]), "+v"(ha[ks][1][1]), "+v"(ha[ks][2][0]),
                     "+v"(ha[ks][2][1]), "+v"(ha[ks][3][0]), "+v"(ha[ks][3][1]), "+v"(hb[ks][0][0]), "+v"(hb[ks][0][1]), "+v"(hb[ks][1][0]),
                     "+v"(hb[ks][1][1]), "+v"(hb[ks][2][0]), "+v"(hb[ks][2][1]), "+v"(hb[ks][3][0]), "+v"(hb[ks][3][1]));
      else if constexpr (TB)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(hb[ks][0][0]), "+v"(hb[ks][0][1]), "+v"(hb[ks][1][0]), "+v"(hb[ks][1][1]), "+v"(hb[ks][2][0]),
                     "+v"(hb[ks][2][1]), "+v"(hb[ks][3][0]), "+v"(hb[ks][3][1]));
    };
    __builtin_amdgcn_sched_barrier(0);
    issue(0);
    issue(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      landed(ks);              // k-step 0: both k-steps' reads are out, lgkmcnt(0) waits for all of them (the second wait is free)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (TA) af[i] = tr_join(ha[ks][i][0], ha[ks][i][1]); else af[i] = row_frag(la, ks, wm * 64 + i * 16 + fr);
        if constexpr (TB) bfr[i] = tr_join(hb[ks][i][0], hb[ks][i][1]); else bfr[i] = row_frag(lb, ks, wn * 64 + i * 16 + fr);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = op16<bf16_t>::mfma(af[i], bfr[j], acc[i][j]);
        if constexpr (COLSUM) { if (do_cs) cs[i] = op16<bf16_t>::mfma(af[i], ones, cs[i]); }
      }
    }
  };

  const int nk = K / TBK;
  // The barrier that hands a staged tile to the fragment reads must see the LDS-DMA pieces LANDED: the transposed reads are inline asm
  // without a memory operand (above), so nothing in the source tells the compiler that they read what stage() writes, and a
  // workgroup-scope fence does not imply vmcnt on gfx9.  hipcc 7.2 happens to emit the wait in front of the loop's s_barrier (advisor,
  // round 5: checked in the disassembly); the explicit wait makes it a property of the source instead of the toolchain.
  auto landed_and_sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  if constexpr (!SPLITK) {
    stage(0, 0);
    landed_and_sync();
    int cur = 0;
    for (int kt = 0; kt < nk - 1; ++kt) {
      stage(cur ^ 1, kt + 1);
      compute(cur);
      landed_and_sync();
      cur ^= 1;
    }
    compute(cur);
  } else {
    const int rounds = (nk + 1) >> 1;                      // group g multiplies tile 2 i + g in round i (the last round may have none for g = 1)
    if (grp < nk) stage(0, grp);
    landed_and_sync();
    int cur = 0;
    for (int i = 0; i < rounds; ++i) {
      if (2 * (i + 1) + grp < nk) stage(cur ^ 1, 2 * (i + 1) + grp);
      if (2 * i + grp < nk) compute(cur);
      landed_and_sync();
      cur ^= 1;
    }
    // the odd tiles' sums join the even ones through LDS (both groups are past their last fragment read): 64 KB of accumulators into
    // group 0's buffers, the bias-gradient column into group 1's
    float* red = (float*)smem_all;
    float* red_cs = (float*)(smem_all + 65536);
    const int t256 = tid & 255;
    if (grp == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) red[((i * 4 + j) * 4 + e) * 256 + t256] = acc[i][j][e];
        if constexpr (COLSUM) {
#pragma unroll
          for (int e = 0; e < 4; ++e) red_cs[(i * 4 + e) * 256 + t256] = cs[i][e];
        }
      }
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] += red[((i * 4 + j) * 4 + e) * 256 + t256];
      if constexpr (COLSUM) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[i][e] += red_cs[(i * 4 + e) * 256 + t256];
      }
    }
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fr;
      const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * 64 + i * 16 + fq * 4 + e;
        if (m < M && n < N) {
          if (C16) C16[(size_t)m * ldc + n] = f2bf(acc[i][j][e] + bv);
          else C[(size_t)m * ldc + n] = acc[i][j][e] + bv;
        }
      }
    }
    if constexpr (COLSUM) {
      if (do_cs && fr == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + wm * 64 + i * 16 + fq * 4 + e;
          if (m < M) colsum_out[m] = cs[i][e];
        }
      }
    }
  }
}

// C[M,N] fp32 = op(A) . op(B) + bias.  ta: A is stored [K][M] (else [M][K]); tb: B is stored [K][N] (else [N][K], nn.Linear's weight:
// only with !ta and at most 256 tiles - the split-K form).
// k_valid <= K: contraction rows that exist in memory (the rest reads as zeros; K itself a multiple of 64).  colsum_out (ta only,
// nullable): [M] column sums of A over k = the bias gradient of a wgrad.  C16 (nullable): store C as bf16 there instead of fp32 into C.
// Returns -1 for an unsupported shape.
int launch_gemm_bf16_tn(bool ta, bool tb, const void* A, int lda, const void* B, int ldb, const float* bias, float* C, int ldc, int M, int N,
                        int K, int k_valid, float* colsum_out, hipStream_t s, void* C16) {
  if (K % TBK || K <= 0 || M <= 0 || N <= 0 || (!tb && ta) || (colsum_out && !ta) || (lda % 8) || (ldb % 8)) return -1;
  const int ntm = (M + TBM - 1) / TBM, ntn = (N + TBN - 1) / TBN;
  const bf16_t* a = (const bf16_t*)A; const bf16_t* b = (const bf16_t*)B;
  const size_t lds = 65536;
  static DeviceOnce once;
  once.run([&] {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_tn_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_tn_kernel<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_tn_kernel<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_tn_kernel<true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_tn_kernel<true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_tn_kernel<false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_tn_kernel<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (int)lds);
  });
  if (!tb) {
    // nn.Linear's own form C = A . B^T (both row-major over k) for launches of at most one 128 x 128 tile per CU: the split-K workgroup
    // above on row-major images (the training step's forward projections, rows = 16 x 128).  Anything larger: gemm.hip's kernels.
    if (ntm * ntn > 256 || K < 2 * TBK || k_valid != K) return -1;
    gemm_bf16_tn_kernel<false, false, false, true><<<ntm * ntn, 512, 2 * lds, s>>>(a, b, bias, C, nullptr, M, N, K, k_valid, lda, ldb, ldc, (bf16_t*)C16);
    return 0;
  }
  static const bool split_off = [] { const char* e = prego_tune_env("PREGO_TN_NO_SPLITK"); return e && *e == '1'; }();
  if (ntm * ntn <= 256 && K >= 2 * TBK && !split_off) {          // at most one workgroup per CU: two K groups of waves share it
    if (ta && colsum_out) gemm_bf16_tn_kernel<true, true, true, true><<<ntm * ntn, 512, 2 * lds, s>>>(a, b, bias, C, colsum_out, M, N, K, k_valid, lda, ldb, ldc, (bf16_t*)C16);
    else if (ta) gemm_bf16_tn_kernel<true, true, false, true><<<ntm * ntn, 512, 2 * lds, s>>>(a, b, bias, C, nullptr, M, N, K, k_valid, lda, ldb, ldc, (bf16_t*)C16);
    else gemm_bf16_tn_kernel<false, true, false, true><<<ntm * ntn, 512, 2 * lds, s>>>(a, b, bias, C, nullptr, M, N, K, k_valid, lda, ldb, ldc, (bf16_t*)C16);
    return 0;
  }
  if (ta && colsum_out) gemm_bf16_tn_kernel<true, true, true><<<ntm * ntn, 256, lds, s>>>(a, b, bias, C, colsum_out, M, N, K, k_valid, lda, ldb, ldc, (bf16_t*)C16);
  else if (ta) gemm_bf16_tn_kernel<true, true, false><<<ntm * ntn, 256, lds, s>>>(a, b, bias, C, nullptr, M, N, K, k_valid, lda, ldb, ldc, (bf16_t*)C16);
  else gemm_bf16_tn_kernel<false, true, false><<<ntm * ntn, 256, lds, s>>>(a, b, bias, C, nullptr, M, N, K, k_valid, lda, ldb, ldc, (bf16_t*)C16);
  return 0;
}
