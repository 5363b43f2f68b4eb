// bf16 NT GEMM, 256x256x64 tiles, PING-PONG schedule (the 8-phase structure of cdna_hip_programming.md section 5):
//   C[M,N] (fp32) = A[M,K] (bf16) * B[N,K]^T (bf16) + bias      layer1 / W_ih projections, model/rnn/rnn.py:38-42,61
//
// 8 waves = two groups of four (group = wave >> 2: the two waves that share a SIMD are in different groups).  The K loop
// is cut into phases of 16 MFMAs per wave (one 64x32 quadrant of the wave's 128x64 output over the 64-wide K tile), and
// every phase is {memory part; s_barrier; MFMA part; s_barrier}.  Group 1 runs ONE BARRIER BEHIND group 0, so in every
// barrier interval one wave of each SIMD is in its MFMA part (256 cycles of matrix pipe) while the other issues its
// ds_reads and LDS-DMA: the matrix pipe always has a wave feeding it, and the memory instructions never sit in front of
// MFMAs of the same wave.
//
// LDS: two K-tile buffers x four half-tiles [A0 | A1 | B0 | B1] of 16 KB (128 rows x 64 k), image = lane-linear LDS-DMA
// pieces (8 rows x 128 B) with the 16-byte-chunk XOR swizzle on the SOURCE address and on the ds_read.  A wave's 128
// rows are 64 from A0 and 64 from A1, its 64 columns 32 from B0 and 32 from B1, so every wave consumes the half-tiles
// in the same order and a slot is dead for all waves at the same phase:
//     phase        ds_read (into)        MFMA quadrant     LDS-DMA issued (one half-tile = 2 pieces per lane)
//     1 of tile t  B0(t)   -> b0 (4)     (a0, b0)          A1(t+1)
//     2            B1(t)   -> b1 (4)     (a0, b1)          A0(t+2)
//     3            A1(t)   -> a1 (8)     (a1, b1)          B0(t+2)
//     4            A0(t+1) -> a0 (8)     (a1, b0)          B1(t+2)
// Every half-tile is issued 6 phases before it is read and waited for 5 phases after its issue, which is ONE counted
// wait per phase: s_waitcnt vmcnt(10) (the five younger half-tiles stay in flight across the barriers; never 0 in the
// steady state).  The wait sits in the memory part of the phase BEFORE the one that reads the half-tile, so both groups
// have passed it (and a barrier) before either reads; a slot is re-staged two phases after its last ds_read, when the
// later group's lgkmcnt(0) has retired those reads too.
#include "common.h"
#include "kernels.h"

#define PBM 256
#define PBN 256
#define PBK 64
#define PHALF 16384
#define PBUF 65536

__device__ __forceinline__ int pp_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

#define PP_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_pingpong_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][A0 | A1 | B0 | B1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;                     // group (= M position wr), N position
  const int ntn = N / PBN;
  const int ntm = (M + PBM - 1) / PBM;
  const int tile = pp_xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * PBM, n0 = (tile % ntn) * PBN;
  const int nk = K / PBK;

  // ---- LDS-DMA sources: this lane's two pieces (8 rows x 128 B) of every half-tile -----------------------------------
  const int sr = lane >> 3, scp = lane & 7;
  const bf16_t* a_src[2][2];       // [half][piece]
  const bf16_t* b_src[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + sr;                      // row inside the half-tile, 0..127
    const int c = scp ^ ((r >> 1) & 7);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      int ar = m0 + hf * 128 + r; if (ar > M - 1) ar = M - 1;
      a_src[hf][i] = A + (size_t)ar * lda + c * 8;
      b_src[hf][i] = B + (size_t)(n0 + hf * 128 + r) * ldb + c * 8;
    }
  }
  // slot X of buffer b: A0 = 0, A1 = 1, B0 = 2, B1 = 3
  auto stage = [&](const bf16_t* const (&src)[2], int slot, int kt) {
    char* dst = smem + (kt & 1) * PBUF + slot * PHALF + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)kt * PBK),
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
  };

  // ---- fragments ------------------------------------------------------------------------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  bf16x8 a0[2][4], a1[2][4], b0[2][2], b1[2][2];                // [k-step][tile]
  auto read_a = [&](bf16x8 (&af)[2][4], int slot, int kt) {
    const char* base = smem + (kt & 1) * PBUF + slot * PHALF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = grp * 64 + i * 16 + fr;
        af[ks][i] = *(const bf16x8*)(base + r * 128 + (((ks * 4 + fq) ^ ((r >> 1) & 7)) << 4));
      }
  };
  auto read_b = [&](bf16x8 (&bf)[2][2], int slot, int kt) {
    const char* base = smem + (kt & 1) * PBUF + slot * PHALF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = wc * 32 + j * 16 + fr;
        bf[ks][j] = *(const bf16x8*)(base + r * 128 + (((ks * 4 + fq) ^ ((r >> 1) & 7)) << 4));
      }
  };
  f32x4 acc[2][2][4][2];                                         // [A half][B half][row tile][col tile]
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[x][y][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mma = [&](f32x4 (&c)[4][2], const bf16x8 (&af)[2][4], const bf16x8 (&bf)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][j], af[ks][i], c[i][j], 0, 0, 0);   // D^T: see epilogue
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: the stream up to (not including) phase 1 of tile 0's issue ------------------------------------------
  stage(a_src[0], 0, 0); stage(b_src[0], 2, 0); stage(b_src[1], 3, 0); stage(a_src[1], 1, 0);     // A0 B0 B1 A1 of tile 0
  if (nk > 1) { stage(a_src[0], 0, 1); stage(b_src[0], 2, 1); stage(b_src[1], 3, 1); PP_WAIT(10); }   // A0 B0 B1 of tile 1
  else PP_WAIT(0);
  bar();
  read_a(a0, 0, 0);
  if (grp == 1) bar();                                           // group 1 runs one barrier behind

  for (int kt = 0; kt < nk; ++kt) {
    const bool full = kt + 2 < nk;                               // every issue of this tile's phases is real
    // phase 1
    read_b(b0, 2, kt);
    if (kt + 1 < nk) stage(a_src[1], 1, kt + 1);
    if (full) PP_WAIT(10); else PP_WAIT(0);
    bar();
    mma(acc[0][0], a0, b0);
    bar();
    // phase 2
    read_b(b1, 3, kt);
    if (full) { stage(a_src[0], 0, kt + 2); PP_WAIT(10); } else PP_WAIT(0);
    bar();
    mma(acc[0][1], a0, b1);
    bar();
    // phase 3
    read_a(a1, 1, kt);
    if (full) { stage(b_src[0], 2, kt + 2); PP_WAIT(10); } else PP_WAIT(0);
    bar();
    mma(acc[1][1], a1, b1);
    bar();
    // phase 4
    if (kt + 1 < nk) read_a(a0, 0, kt + 1);
    if (full) { stage(b_src[1], 3, kt + 2); PP_WAIT(10); } else PP_WAIT(0);
    bar();
    mma(acc[1][0], a1, b0);
    bar();
  }
  if (grp == 0) bar();                                           // pairs with group 1's last barrier

  // ---- epilogue: the products were taken as (B-fragment) x (A-fragment), i.e. transposed 16x16 tiles, so a lane holds FOUR
  // CONSECUTIVE COLUMNS of one row of C: 16-byte stores (4x fewer store instructions than the row-major accumulator
  // layout's dword stores; the store tail of a tile is issue-bound)
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + y * 128 + wc * 32 + j * 16 + fq * 4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *(const float4*)(bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + x * 128 + grp * 64 + i * 16 + fr;
          if (m < M) {
            const f32x4 v = acc[x][y][i][j];
            *(float4*)(C + (size_t)m * ldc + n) = make_float4(v[0] + bv.x, v[1] + bv.y, v[2] + bv.z, v[3] + bv.w);
          }
        }
      }
}

int launch_gemm_bf16_pingpong(const void* A, int lda, const void* B, int ldb, const float* bias, float* C, int ldc, int M, int N,
                              int K, hipStream_t s) {
  if (N % PBN || K % PBK || K < 2 * PBK) return -1;
  const int ntm = (M + PBM - 1) / PBM, ntn = N / PBN;
  (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
  gemm_bf16_nt_pingpong_kernel<<<ntm * ntn, 512, 2 * PBUF, s>>>((const bf16_t*)A, (const bf16_t*)B, bias, C, M, N, K, lda, ldb, ldc);
  return 0;
}
