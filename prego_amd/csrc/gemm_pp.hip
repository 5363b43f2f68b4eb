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
#include <cstdlib>
#include <cstdio>

#define PBM 256
#define PBN 256
#define PBK 64
#define PHALF 16384
#define PBUF 65536

__device__ __forceinline__ int pp_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

#ifdef PP_DIAG
__device__ unsigned long long g_pp_diag[8];      // diagnostic build only: cycle sums over workgroups (wave 0)
#endif
// B half-tiles are read with their rows permuted (read_b below), so their 16-byte-chunk XOR key is a different function of the
// row than A's (r >> 1) & 7: distinct over the 16 rows one fragment read touches, {8a + 4j + b : a, b = 0..3}
__device__ __forceinline__ int pp_key_b(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }
#define PP_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// EPI = what happens to acc + bias (kernels.h): EPI_STORE (fp32 C), EPI_STORE_BF16 (bf16 C: the inference path keeps the
// layer1 and W_ih projections in bf16 between the kernels), and the Transformer path's fused epilogues EPI_RESIDUAL (fp32
// x += .), EPI_GELU_BF16 (bf16(gelu(.))), EPI_QKV (head split into Q, K [B,h,N,dh] and V^T [B,h,dh,Npad]).  A lane holds 8
// consecutive columns of a row, so every epilogue but V^T's is 16-byte vector accesses.
// OT = operand type tag (common.h: bf16_t or f16_t): selects the MFMA opcode and the 16-bit output conversion only.
// WORKER (probe of DESIGN 5c, prego_debug_gemm_worker): a persistent workgroup that leaves at once on XCDs below epi.xcd_lo and
// otherwise claims tiles from the atomic counter epi.counter until none is left (m-major order: consecutive tiles share A rows).
// SPLIT (fp16x2 operands, common.h): A rows are [K hi | .. | K lo at column epi.split_a_lo], B rows likewise at epi.split_b_lo.  A K
// tile then covers 32 values of k and its 128-byte LDS row holds [32 hi | 32 lo]: the 16-byte chunks 0-3 of a row come from the hi
// half, chunks 4-7 from the lo half of the same k range (a per-lane constant in the DMA source offset, nothing else changes in the
// staging, the LDS ring or the wait counts), so the k-step-0 fragments ARE the hi operands and the k-step-1 fragments the lo
// operands, and a phase multiplies a_hi.b_lo, a_lo.b_hi, a_hi.b_hi (small terms first): 24 MFMAs behind the same 12 fragment
// reads and 2 DMA pieces that serve 16 in the 16-bit kernel - a third less LDS traffic and DMA per product than walking three K
// segments through the unchanged loop (the first fp16x2 form: 165 -> 136 ms per pass of the bench workload, 1.49 PFLOP/s of executed MFMA work).  The epilogue multiplies the
// weights' power-of-two scale out (epi.acc_scale: device pointer to 1 / scale) before the bias.
template <int EPI, typename OT = bf16_t, bool WORKER = false, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_pingpong_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, const float* __restrict__ bias,
    void* __restrict__ Cv, int M, int N, int K, int lda, int ldb, int ldc, GemmEpi epi) {
  float* C = (float*)Cv;
  constexpr bool OUT_BF16 = EPI == EPI_STORE_BF16;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][A0 | A1 | B0 | B1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;                     // group (= M position wr), N position
  const int ntn = N / PBN;
  const int ntm = (M + PBM - 1) / PBM;
  const int ntiles = ntm * ntn;
  const int nk = SPLIT ? K / (PBK / 2) : K / PBK;               // SPLIT: a K tile = 32 values of k (hi and lo halves side by side)
  int idx = blockIdx.x;
  int tile = WORKER ? 0 : pp_xcd_remap(idx, ntiles);
  int m0 = (tile / ntn) * PBM, n0 = (tile % ntn) * PBN;
  __shared__ int s_tile;
  if constexpr (WORKER) {
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;           // HW_REG_XCC_ID
    if (xcc < epi.xcd_lo) return;
  }

  // ---- LDS-DMA sources: this lane's two pieces (8 rows x 128 B) of every half-tile.  One buffer resource per operand and
  // tile (base = the tile's first row, num_records = its valid rows: rows past M read as zeros, no clamp), a 32-bit lane
  // offset per piece, and the half-tile / K-tile displacement in the scalar offset: 4 VGPRs of addressing in all
  const int sr = lane >> 3, scp = lane & 7;
  int a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + sr;                      // row inside the half-tile, 0..127
    const int c = scp ^ ((r >> 1) & 7);
    const int cb = scp ^ pp_key_b(r);
    if constexpr (SPLIT) {                                      // chunks 0-3: k .. k + 31 of the hi half; chunks 4-7: the same k of the lo half
      a_off[i] = r * lda * 2 + (c < 4 ? c * 16 : epi.split_a_lo * 2 + (c - 4) * 16);
      b_off[i] = r * ldb * 2 + (cb < 4 ? cb * 16 : epi.split_b_lo * 2 + (cb - 4) * 16);
    } else {
      a_off[i] = r * lda * 2 + c * 16;
      b_off[i] = r * ldb * 2 + cb * 16;
    }
  }
  __amdgpu_buffer_rsrc_t rs_a, rs_b;
  auto set_sources = [&](int tm0, int tn0) {
    const int rows = M - tm0 < PBM ? M - tm0 : PBM;
    rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)tm0 * lda), 0, rows * lda * 2, 0x00020000);
    rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)tn0 * ldb), 0, PBN * ldb * 2, 0x00020000);
  };
  if constexpr (!WORKER) set_sources(m0, n0);
  // slot X of buffer b: A0 = 0, A1 = 1, B0 = 2, B1 = 3
  // timing-only diagnostic builds (scripts/probes/pp_skip.sh; never the product library): -DPP_SKIP_DMA issues no LDS-DMA,
  // -DPP_SKIP_READS re-reads no fragment inside the K loop - wrong results, the K loop's instruction mix minus that part
  auto stage_a = [&](int hf, int kt) {
#ifdef PP_SKIP_DMA
    return;
#endif
    char* dst = smem + (kt & 1) * PBUF + hf * PHALF + wave * 2048;
    const int so = hf * 128 * lda * 2 + kt * (SPLIT ? PBK : PBK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, a_off[i], so, 0, 0);
  };
  auto stage_b = [&](int hf, int kt) {
#ifdef PP_SKIP_DMA
    return;
#endif
    char* dst = smem + (kt & 1) * PBUF + (2 + hf) * PHALF + wave * 2048;
    const int so = hf * 128 * ldb * 2 + kt * (SPLIT ? PBK : PBK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, b_off[i], so, 0, 0);
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
        // MFMA row fr of tile j <- column 8*(fr >> 2) + 4*j + (fr & 3) of the wave's 32: the transposed accumulators of the two
        // tiles then hold EIGHT CONSECUTIVE columns per lane (4*fq' .. : j = 0 | j = 1), one 16-byte store in bf16, 32 B in fp32
        const int r = wc * 32 + (fr >> 2) * 8 + j * 4 + (fr & 3);
        bf[ks][j] = *(const bf16x8*)(base + r * 128 + (((ks * 4 + fq) ^ pp_key_b(r)) << 4));
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
#ifndef PP_PRIO
#define PP_PRIO 1
#endif
    __builtin_amdgcn_s_setprio(PP_PRIO);
    if constexpr (SPLIT) {               // k-step 0 = hi fragments, k-step 1 = lo fragments of the same 32 k: three products, small first
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          c[i][j] = op16<OT>::mfma(bf[1][j], af[0][i], c[i][j]);
          c[i][j] = op16<OT>::mfma(bf[0][j], af[1][i], c[i][j]);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) c[i][j] = op16<OT>::mfma(bf[0][j], af[0][i], c[i][j]);
    } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) c[i][j] = op16<OT>::mfma(bf[ks][j], af[ks][i], c[i][j]);   // D^T: see epilogue
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: the stream up to (not including) phase 1 of tile 0's issue ------------------------------------------
  auto prologue = [&]() {
    stage_a(0, 0); stage_b(0, 0); stage_b(1, 0); stage_a(1, 0);     // A0 B0 B1 A1 of K tile 0
    stage_a(0, 1); stage_b(0, 1); stage_b(1, 1);                            // A0 B0 B1 of K tile 1 (nk >= 2)
  };
  for (;;) {                                                     // one pass unless WORKER
  if constexpr (WORKER) {
    if (tid == 0) s_tile = (int)atomicAdd(epi.counter, 1u);
    __syncthreads();
    tile = s_tile;
    __syncthreads();
    if (tile >= ntiles) break;
    m0 = (tile / ntn) * PBM; n0 = (tile % ntn) * PBN;
    set_sources(m0, n0);
  }
  prologue();
  PP_WAIT(10);
  bar();
  read_a(a0, 0, 0);
  if (grp == 1) bar();                                           // group 1 runs one barrier behind
#ifdef PP_DIAG
  const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
  unsigned long long dacc[5] = {0, 0, 0, 0, 0}, dt = t_loop0;
#define DS(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); dacc[i] += n_ - dt; dt = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DS(i) do { } while (0)
#endif

#ifdef PP_SKIP_READS
  read_b(b0, 2, 0); read_b(b1, 3, 0); read_a(a1, 1, 0);
#define read_a(...) do { } while (0)
#define read_b(...) do { } while (0)
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const bool full = kt + 2 < nk;                               // every issue of this tile's phases is real
    // phase 1
    read_b(b0, 2, kt);
    if (kt + 1 < nk) stage_a(1, kt + 1);
    if (full) PP_WAIT(10); else PP_WAIT(0);
    bar();
    mma(acc[0][0], a0, b0);
    bar();
    // phase 2
    read_b(b1, 3, kt);
    if (full) { stage_a(0, kt + 2); PP_WAIT(10); } else PP_WAIT(0);
    bar();
    mma(acc[0][1], a0, b1);
    bar();
    // phase 3
    DS(4);
    read_a(a1, 1, kt);
    if (full) { stage_b(0, kt + 2); DS(0); PP_WAIT(10); } else { DS(0); PP_WAIT(0); }
    DS(1);
    bar();
    DS(2);
    mma(acc[1][1], a1, b1);
    DS(3);
    bar();
    DS(2);
    // phase 4
    if (kt + 1 < nk) read_a(a0, 0, kt + 1);
    if (full) { stage_b(1, kt + 2); PP_WAIT(10); } else PP_WAIT(0);
    bar();
    mma(acc[1][0], a1, b0);
    bar();
  }
#ifdef PP_SKIP_READS
#undef read_a
#undef read_b
#endif
  if (grp == 0) bar();                                           // pairs with group 1's last barrier: every LDS read of this tile has retired

  const int cm0 = m0, cn0 = n0;
  // lane (fq, fr) holds, for row tile i of quadrant (x, y): row fr, columns y*128 + wc*32 + fq*8 + j*4 + e  (j = tile, e = register)
  float4 bv[2][2];
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[y][j] = *(const float4*)(bias + cn0 + y * 128 + wc * 32 + fq * 8 + j * 4);
  // ---- epilogue: the products were taken as (B-fragment) x (A-fragment), i.e. transposed 16x16 tiles, and the B rows were
  // permuted on the way in, so a lane holds EIGHT CONSECUTIVE COLUMNS of one row of C: one 16-byte store per row in bf16,
  // two in fp32 (the store tail of a tile is bound by the number of store instructions first, by bytes second)
#ifdef PP_DIAG
  const unsigned long long t_epi0 = __builtin_amdgcn_s_memtime();
#endif
  const bool whole = cm0 + PBM <= M;                             // wave-uniform: all the stores of this wave are issued
  float inv_scale = 1.f;
  if constexpr (SPLIT) { if (epi.acc_scale) inv_scale = *epi.acc_scale; }
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int n = cn0 + y * 128 + wc * 32 + fq * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = cm0 + x * 128 + grp * 64 + i * 16 + fr;
        f32x4 v0 = acc[x][y][i][0], v1 = acc[x][y][i][1];
        if constexpr (SPLIT) { v0 *= inv_scale; v1 *= inv_scale; }
        if (whole || m < M) {
          float o[8] = {v0[0] + bv[y][0].x, v0[1] + bv[y][0].y, v0[2] + bv[y][0].z, v0[3] + bv[y][0].w,
                        v1[0] + bv[y][1].x, v1[1] + bv[y][1].y, v1[2] + bv[y][1].z, v1[3] + bv[y][1].w};
          if constexpr (EPI == EPI_STORE_BF16 || EPI == EPI_GELU_BF16) {
            if constexpr (EPI == EPI_GELU_BF16) {
              if (epi.pre_f32) {                                         // training: keep the pre-activation for gelu'
                *(float4*)(epi.pre_f32 + (size_t)m * ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
                *(float4*)(epi.pre_f32 + (size_t)m * ldc + n + 4) = make_float4(o[4], o[5], o[6], o[7]);
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = gelu_erf_(o[e]);
              if (epi.drop_thresh) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = dropout_keep_(epi.drop_seed, (size_t)m * ldc + n + e, epi.drop_thresh) ? o[e] * epi.drop_scale : 0.f;
              }
            }
            uint4 pk;
            pk.x = op16<OT>::pack2_sat(o[0], o[1]); pk.y = op16<OT>::pack2_sat(o[2], o[3]); pk.z = op16<OT>::pack2_sat(o[4], o[5]); pk.w = op16<OT>::pack2_sat(o[6], o[7]);
            bf16_t* ob = EPI == EPI_GELU_BF16 ? (bf16_t*)epi.out_b : (bf16_t*)Cv;
            *(uint4*)(ob + (size_t)m * ldc + n) = pk;
          } else if constexpr (EPI == EPI_RESIDUAL) {                  // fp32 residual stream, in place
            if (epi.drop_thresh) {
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = dropout_keep_(epi.drop_seed, (size_t)m * ldc + n + e, epi.drop_thresh) ? o[e] * epi.drop_scale : 0.f;
            }
            if (epi.drop2_thresh) {
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = dropout_keep_(epi.drop2_seed, (size_t)m * ldc + n + e, epi.drop2_thresh) ? o[e] * epi.drop2_scale : 0.f;
            }
            float4* xr = (float4*)(C + (size_t)m * ldc + n);
            const float4 r0 = xr[0], r1 = xr[1];
            xr[0] = make_float4(r0.x + o[0], r0.y + o[1], r0.z + o[2], r0.w + o[3]);
            xr[1] = make_float4(r1.x + o[4], r1.y + o[5], r1.z + o[6], r1.w + o[7]);
          } else if constexpr (EPI == EPI_TOKENS) {                     // token rows of the ViT residual stream: + pe[t], one row per window skipped
            const int b = m / epi.n_tok, t = m - b * epi.n_tok;
            const float4* pr = (const float4*)(epi.pe + (size_t)t * ldc + n);
            const float4 p0 = pr[0], p1 = pr[1];
            float4* xr = (float4*)(C + (size_t)(m + b) * ldc + n);
            xr[0] = make_float4(o[0] + p0.x, o[1] + p0.y, o[2] + p0.z, o[3] + p0.w);
            xr[1] = make_float4(o[4] + p1.x, o[5] + p1.y, o[6] + p1.z, o[7] + p1.w);
          } else if constexpr (EPI == EPI_QKV) {                        // 8 consecutive columns = one head, one of q / k / v
            const int b = m / epi.n_tok, t = m - b * epi.n_tok;
            const int blk = n / epi.emb, r = n - blk * epi.emb, which = blk + epi.which0;
            const int hd = r / epi.dh, d = r - hd * epi.dh;
            const size_t bh = (size_t)b * epi.heads + hd;
            if (which < 2) {
              const float sc = which == 0 ? epi.q_scale : 1.f;
              uint4 pk;
              pk.x = op16<OT>::pack2_sat(o[0] * sc, o[1] * sc); pk.y = op16<OT>::pack2_sat(o[2] * sc, o[3] * sc);
              pk.z = op16<OT>::pack2_sat(o[4] * sc, o[5] * sc); pk.w = op16<OT>::pack2_sat(o[6] * sc, o[7] * sc);
              bf16_t* dst = which == 0 ? (bf16_t*)epi.q : (bf16_t*)epi.k;
              *(uint4*)(dst + (bh * epi.n_tok + t) * epi.dh + d) = pk;
            } else {
              uint4 pk;                                                  // V by rows [B,h,n_tok,dh]: one 16-byte store
              pk.x = op16<OT>::pack2_sat(o[0], o[1]); pk.y = op16<OT>::pack2_sat(o[2], o[3]); pk.z = op16<OT>::pack2_sat(o[4], o[5]); pk.w = op16<OT>::pack2_sat(o[6], o[7]);
              *(uint4*)((bf16_t*)epi.vn + (bh * epi.n_tok + t) * epi.dh + d) = pk;
            }
          } else {
            *(float4*)(C + (size_t)m * ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
            *(float4*)(C + (size_t)m * ldc + n + 4) = make_float4(o[4], o[5], o[6], o[7]);
          }
        }
        acc[x][y][i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[x][y][i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  if constexpr (!WORKER) break;
  }
#ifdef PP_DIAG
  if (wave == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
      atomicAdd(&g_pp_diag[0], t1 - t_epi0);      // store issue
      atomicAdd(&g_pp_diag[1], t2 - t1);          // drain after the last store was issued
      atomicAdd(&g_pp_diag[2], t_epi0 - t_loop0); // K loop
      atomicAdd(&g_pp_diag[3], 1ull);
      atomicAdd(&g_pp_diag[4], dacc[0]); atomicAdd(&g_pp_diag[5], dacc[1]); atomicAdd(&g_pp_diag[6], dacc[2]); atomicAdd(&g_pp_diag[7], dacc[3]);
    }
  }
#endif
}

__device__ float g_pp_zero_bias[8192];        // stands in for a NULL bias (the kernel's bias loads are unconditional)

// C[M,N] = epilogue(A[M,K] . B[N,K]^T + bias) with the epilogue of epi.mode; bias may be NULL.  -1 = shape not supported.
int launch_gemm_bf16_pingpong_epi(const void* A, int lda, const void* B, int ldb, const float* bias, void* C, int ldc, int M, int N,
                                  int K, GemmEpi epi, hipStream_t s) {
  if (N % PBN || K % PBK || K < 2 * PBK || (bias == nullptr && N > 8192)) return -1;
  if (epi.mode == EPI_QKV && (epi.dh % 8 || epi.emb % 8)) return -1;
  const int ntm = (M + PBM - 1) / PBM, ntn = N / PBN, ntiles = ntm * ntn;
  static DeviceOnce once;
  static float* zero_bias[64] = {nullptr};
  const int dev = once.run([&] {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_STORE_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_RESIDUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_GELU_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_QKV>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_TOKENS>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_STORE, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_STORE_BF16, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_RESIDUAL, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_GELU_BF16, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_QKV, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_TOKENS, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    int d = 0;
    (void)hipGetDevice(&d);
    if (d >= 0 && d < 64) (void)hipGetSymbolAddress((void**)&zero_bias[d], HIP_SYMBOL(g_pp_zero_bias));
  });
  if (!bias) bias = zero_bias[dev];
  if (!bias) return -1;
  const bf16_t* a = (const bf16_t*)A; const bf16_t* b = (const bf16_t*)B;
#define PPL(E) gemm_bf16_nt_pingpong_kernel<E><<<ntiles, 512, 2 * PBUF, s>>>(a, b, bias, C, M, N, K, lda, ldb, ldc, epi)
  if (epi.f16) {                 // IEEE fp16 operands / 16-bit outputs: same kernel, other MFMA opcode and conversions
#define PPL16(E) gemm_bf16_nt_pingpong_kernel<E, f16_t><<<ntiles, 512, 2 * PBUF, s>>>(a, b, bias, C, M, N, K, lda, ldb, ldc, epi)
    switch (epi.mode) {
      case EPI_STORE: PPL16(EPI_STORE); break;
      case EPI_STORE_BF16: PPL16(EPI_STORE_BF16); break;
      case EPI_RESIDUAL: PPL16(EPI_RESIDUAL); break;
      case EPI_GELU_BF16: PPL16(EPI_GELU_BF16); break;
      case EPI_TOKENS: PPL16(EPI_TOKENS); break;
      default: PPL16(EPI_QKV); break;
    }
#undef PPL16
    return 0;
  }
  switch (epi.mode) {
    case EPI_STORE: PPL(EPI_STORE); break;
    case EPI_STORE_BF16: PPL(EPI_STORE_BF16); break;
    case EPI_RESIDUAL: PPL(EPI_RESIDUAL); break;
    case EPI_GELU_BF16: PPL(EPI_GELU_BF16); break;
    case EPI_TOKENS: PPL(EPI_TOKENS); break;
    default: PPL(EPI_QKV); break;
  }
#undef PPL
  return 0;
}

// split-operand (fp16x2) projection: C[M,N] fp32 = (A_hi + A_lo)[M,K] . (B_hi + B_lo)[N,K]^T * inv_scale + bias, three fp16 products.
// A rows [.. lda fp16 ..] hold hi at column 0 and lo at column a_lo; B rows hi at 0 and lo at b_lo; any M (rows past M read as zeros).
int launch_gemm_x2_pingpong(const void* A, int lda, int a_lo, const void* B, int ldb, int b_lo, const float* inv_scale, const float* bias,
                            float* C, int ldc, int M, int N, int K, hipStream_t s) {
  if (N % PBN || K % (PBK / 2) || K < PBK || !bias || M <= 0) return -1;
  const int ntm = (M + PBM - 1) / PBM, ntn = N / PBN, ntiles = ntm * ntn;
  static DeviceOnce once;
  once.run([&] {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_STORE, f16_t, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
  });
  GemmEpi epi{};
  epi.mode = EPI_STORE; epi.f16 = 1; epi.split_a_lo = a_lo; epi.split_b_lo = b_lo; epi.acc_scale = inv_scale;
  gemm_bf16_nt_pingpong_kernel<EPI_STORE, f16_t, false, true><<<ntiles, 512, 2 * PBUF, s>>>((const bf16_t*)A, (const bf16_t*)B, bias, C, M, N, K, lda, ldb,
                                                                                              ldc, epi);
  return 0;
}

// The ping-pong kernel as a persistent worker: `grid` workgroups claim tiles from *counter (zero at launch) in m-major order; those that
// land on XCDs below xcd_lo leave at once.  out16: 16-bit C of the operand type (the inference projections), else fp32 C.
int launch_gemm_bf16_pingpong_worker(const void* A, int lda, const void* B, int ldb, const float* bias, void* C, int ldc, int M, int N,
                                     int K, int xcd_lo, unsigned* counter, int grid, hipStream_t s, bool out16, bool f16) {
  if (N % PBN || K % PBK || K < 2 * PBK || !bias || !counter) return -1;
  static DeviceOnce once;
  once.run([&] {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_STORE, bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_STORE_BF16, bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_pingpong_kernel<EPI_STORE_BF16, f16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PBUF);
  });
  GemmEpi epi{};
  epi.mode = out16 ? EPI_STORE_BF16 : EPI_STORE; epi.xcd_lo = xcd_lo; epi.counter = counter; epi.f16 = f16 ? 1 : 0;
  const bf16_t* a = (const bf16_t*)A; const bf16_t* b = (const bf16_t*)B;
  if (!out16) gemm_bf16_nt_pingpong_kernel<EPI_STORE, bf16_t, true><<<grid, 512, 2 * PBUF, s>>>(a, b, bias, C, M, N, K, lda, ldb, ldc, epi);
  else if (f16) gemm_bf16_nt_pingpong_kernel<EPI_STORE_BF16, f16_t, true><<<grid, 512, 2 * PBUF, s>>>(a, b, bias, C, M, N, K, lda, ldb, ldc, epi);
  else gemm_bf16_nt_pingpong_kernel<EPI_STORE_BF16, bf16_t, true><<<grid, 512, 2 * PBUF, s>>>(a, b, bias, C, M, N, K, lda, ldb, ldc, epi);
  return 0;
}

// fp32 or bf16 C with the bias epilogue (the MiniROAD projections); bias must not be NULL here
int launch_gemm_bf16_pingpong_mode(int mode, const void* A, int lda, const void* B, int ldb, const float* bias, void* C, int ldc,
                                   int M, int N, int K, bool out_bf16, hipStream_t s, bool f16) {
  (void)mode;
  if (bias == nullptr) return -1;
  GemmEpi epi{};
  epi.f16 = f16 ? 1 : 0;
  epi.mode = out_bf16 ? EPI_STORE_BF16 : EPI_STORE;
  return launch_gemm_bf16_pingpong_epi(A, lda, B, ldb, bias, C, ldc, M, N, K, epi, s);
}
#ifdef PP_DIAG
void pp_diag_print() {
  unsigned long long h[8];
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pp_diag), sizeof h);
  if (h[3]) printf("pp diag per tile (wave 0, cycles): K loop %.0f, store issue %.0f, drain after last issue %.0f (tiles %llu)\n",
                   (double)h[2] / h[3], (double)h[0] / h[3], (double)h[1] / h[3], h[3]);
  if (h[3]) printf("   phase 3 per tile: reads+DMA issue %.0f, vmcnt wait %.0f, two barriers %.0f, 16 MFMA %.0f\n", (double)h[4] / h[3], (double)h[5] / h[3],
                   (double)h[6] / h[3], (double)h[7] / h[3]);
  unsigned long long z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pp_diag), z, sizeof z);
}
#endif
int launch_gemm_bf16_pingpong(const void* A, int lda, const void* B, int ldb, const float* bias, float* C, int ldc, int M, int N,
                              int K, hipStream_t s) {
  return launch_gemm_bf16_pingpong_mode(0, A, lda, B, ldb, bias, C, ldc, M, N, K, false, s);
}
