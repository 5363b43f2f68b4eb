// Dense "NT" GEMMs of the MiniROAD path:  C[M,N] = A[M,K] . B[N,K]^T + bias[N]
// (nn.Linear semantics: both operands K-contiguous).  Call sites replaced:
//   layer1[0]  Linear(4096,2048)      model/rnn/rnn.py:40     M = packed rows, N = 2048, K = 4096
//   GRU input projection W_ih          model/rnn/rnn.py:38,61  M = packed rows, N = 3072, K = 2048
//
// bf16 kernel (the product path): 128x128x64 tiles, 4 waves (2x2, 64x64 each), 16x16x32 bf16 MFMA,
// fp32 accumulate.  Operands go HBM -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4); the LDS image
// is lane-linear per wave-instruction (8 rows x 128 B), the 16-byte chunk XOR-swizzle sits on the SOURCE
// address and on the ds_read address (cdna_hip_programming.md rule 21) so the ds_read_b128 fragments are
// bank-conflict free.  Two LDS buffers, one barrier per K tile.  Block ids are remapped so that one XCD's
// L2 sees consecutive N tiles of the same M tile (A rows are re-used from L2).
//
// fp32 kernel (parity mode): exact-fp32 MFMA 16x16x4 (bit-for-bit an fmaf chain), register staged.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

#define BM 128
#define BN 128
#define BK 64

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // bijective XCD-aware remap (cdna_hip_programming.md, T1): blocks b and b+8 share an XCD
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// exact (erf) GELU = nn.GELU() default (Transformer.py:40)

template <typename OutT, int EPI, typename OT = bf16_t>
__global__ __launch_bounds__(256, 2) void gemm_bf16_nt_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, const float* __restrict__ bias,
    OutT* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc, GemmEpi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A 16 KB | B 16 KB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = N / BN;
  const int ntm = (M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

  // per-lane staging source (row within the 8-row group of a wave-instruction, swizzled chunk)
  const int sr = lane >> 3;          // 0..7
  const int scp = lane & 7;          // chunk position in LDS
  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + sr;                 // tile row 0..127
    const int c = scp ^ ((r >> 1) & 7);                    // source chunk that lands at position scp
    int ar = m0 + r; if (ar > M - 1) ar = M - 1;           // clamp: rows past M are never stored
    a_src[i] = A + (size_t)ar * lda + c * 8;
    b_src[i] = B + (size_t)(n0 + r) * ldb + c * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* la = smem + buf * 32768;
    char* lb = la + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = (wave * 4 + i) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + (size_t)kt * BK),
                                       (__attribute__((address_space(3))) void*)(la + off), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + (size_t)kt * BK),
                                       (__attribute__((address_space(3))) void*)(lb + off), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = w*64 + i*16 + (lane&15); chunk = ks*4 + (lane>>4), swizzled
  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](int buf) {
    const char* la = smem + buf * 32768;
    const char* lb = la + 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = wm * 64 + i * 16 + fr;
        af[i] = *(const bf16x8*)(la + ra * 128 + (((ks * 4 + fq) ^ ((ra >> 1) & 7)) << 4));
        const int rb = wn * 64 + i * 16 + fr;
        bfr[i] = *(const bf16x8*)(lb + rb * 128 + (((ks * 4 + fq) ^ ((rb >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = op16<OT>::mfma(af[i], bfr[j], acc[i][j]);
    }
  };

  const int nk = K / BK;
  stage(0, 0);
  __syncthreads();                      // emits vmcnt(0): tile 0 landed
  int cur = 0;
  for (int kt = 0; kt < nk - 1; ++kt) {
    stage(cur ^ 1, kt + 1);
    compute(cur);
    __syncthreads();                    // vmcnt(0) + barrier: next tile landed, everyone done reading cur
    cur ^= 1;
  }
  compute(cur);

  // epilogue: D layout col = lane&15, row = (lane>>4)*4 + j
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fr;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * 64 + i * 16 + fq * 4 + e;
        if (m < M) {
          float v = acc[i][j][e] + bv;
          if constexpr (EPI == EPI_STORE) {
            if constexpr (sizeof(OutT) == 2) C[(size_t)m * ldc + n] = op16<OT>::cvt_sat(v);
            else C[(size_t)m * ldc + n] = v;
          } else if constexpr (EPI == EPI_RESIDUAL) {                 // x += v  (fp32 residual stream, in place)
            if (epi.drop_thresh) v = dropout_keep_(epi.drop_seed, (size_t)m * ldc + n, epi.drop_thresh) ? v * epi.drop_scale : 0.f;
            if (epi.drop2_thresh) v = dropout_keep_(epi.drop2_seed, (size_t)m * ldc + n, epi.drop2_thresh) ? v * epi.drop2_scale : 0.f;
            float* xr = (float*)C + (size_t)m * ldc + n;
            *xr = *xr + v;
          } else if constexpr (EPI == EPI_GELU_BF16) {                // FFN-1: bf16(gelu(v))
            if (epi.pre_f32) epi.pre_f32[(size_t)m * ldc + n] = v;   // training: keep the pre-activation for gelu'
            float gv = gelu_erf_(v);
            if (epi.drop_thresh) gv = dropout_keep_(epi.drop_seed, (size_t)m * ldc + n, epi.drop_thresh) ? gv * epi.drop_scale : 0.f;
            ((bf16_t*)epi.out_b)[(size_t)m * ldc + n] = op16<OT>::cvt_sat(gv);
          } else if constexpr (EPI == EPI_TOKENS) {                   // x[b, t] = v + pe[t], row m = b n_tok + t -> m + b
            const int b = m / epi.n_tok, t = m - b * epi.n_tok;
            ((float*)C)[(size_t)(m + b) * ldc + n] = v + epi.pe[(size_t)t * ldc + n];
          } else if constexpr (EPI == EPI_STORE_BF16) {
            ((bf16_t*)epi.out_b)[(size_t)m * ldc + n] = op16<OT>::cvt_sat(v);
          } else {                                               // EPI_QKV: split into Q, K, V [B,h,Ntok,dh]
            const int b = m / epi.n_tok, t = m - b * epi.n_tok;
            const int blk = n / epi.emb, r = n - blk * epi.emb, which = blk + epi.which0;
            const int hd = r / epi.dh, d = r - hd * epi.dh;
            const size_t bh = (size_t)b * epi.heads + hd;
            if (which == 0) ((bf16_t*)epi.q)[(bh * epi.n_tok + t) * epi.dh + d] = op16<OT>::cvt_sat(v * epi.q_scale);
            else if (which == 1) ((bf16_t*)epi.k)[(bh * epi.n_tok + t) * epi.dh + d] = op16<OT>::cvt_sat(v);
            else {
              ((bf16_t*)epi.vn)[(bh * epi.n_tok + t) * epi.dh + d] = op16<OT>::cvt_sat(v);
            }
          }
        }
      }
    }
}


// ------------------------------------------------------------------------------------------------------
// Large-M variant for the two eval-path projections: 256x128x64 tiles, 8 waves (4 x 2, 64x64 each = two waves per
// SIMD), THREE LDS stages (144 KB, one workgroup per CU) with the prefetch two K tiles ahead: the loop waits with a
// COUNTED vmcnt (the newest tile stays in flight across the barrier) and uses a raw s_barrier, so the LDS-DMA of tile
// kt+2 and kt+1 overlap the MFMAs of tile kt (cdna_hip_programming.md "Pipelining across barriers", 3-buffer span).
// One barrier per K tile: passing barrier(kt) proves every wave finished computing tile kt-1, whose buffer is the one
// stage(kt+2) overwrites.  Same lane-linear LDS image + source/read XOR swizzle as the 128x128 kernel.
// ------------------------------------------------------------------------------------------------------
#define GBM 256
#define GSTAGE 49152          // A 32 KB + B 16 KB
__global__ __launch_bounds__(512, 2) void gemm_bf16_nt_big_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3][A 32 KB | B 16 KB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;                      // 4 x 2
  const int ntn = N / BN;
  const int ntm = (M + GBM - 1) / GBM;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * GBM, n0 = (tile % ntn) * BN;

  const int sr = lane >> 3, scp = lane & 7;
  // A: 32 wave-instructions per stage (4 per wave), B: 16 (2 per wave)
  const bf16_t* a_src[4];
  const bf16_t* b_src[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave * 4 + i) * 8 + sr;                 // 0..255
    const int c = scp ^ ((r >> 1) & 7);
    int ar = m0 + r; if (ar > M - 1) ar = M - 1;
    a_src[i] = A + (size_t)ar * lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + sr;                 // 0..127
    const int c = scp ^ ((r >> 1) & 7);
    b_src[i] = B + (size_t)(n0 + r) * ldb + c * 8;
  }
  auto stage = [&](int buf, int kt) {
    char* la = smem + buf * GSTAGE;
    char* lb = la + 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + (size_t)kt * BK),
                                       (__attribute__((address_space(3))) void*)(la + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + (size_t)kt * BK),
                                       (__attribute__((address_space(3))) void*)(lb + (wave * 2 + i) * 1024), 16, 0, 0);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](int buf) {
    const char* la = smem + buf * GSTAGE;
    const char* lb = la + 32768;
    // all 16 fragments of the K tile are requested up front (64 VGPRs): the MFMAs of k-step 0 run under the LDS
    // latency of k-step 1's fragments instead of each group of 8 MFMAs waiting for its own reads
    bf16x8 af[2][4], bfr[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = wm * 64 + i * 16 + fr;
        af[ks][i] = *(const bf16x8*)(la + ra * 128 + (((ks * 4 + fq) ^ ((ra >> 1) & 7)) << 4));
        const int rb = wn * 64 + i * 16 + fr;
        bfr[ks][i] = *(const bf16x8*)(lb + rb * 128 + (((ks * 4 + fq) ^ ((rb >> 1) & 7)) << 4));
      }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  const int nk = K / BK;
  stage(0, 0);
  if (nk > 1) stage(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt landed (6 LDS-DMA per wave per tile; leave the newer tile in flight), then meet
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");                      // no LDS access may be scheduled above the barrier
    // (staggering waves 4-7 - prefetch after their MFMAs - measured 8 % slower here: 828 vs 905 TFLOP/s)
    if (kt + 2 < nk) stage((kt + 2) % 3, kt + 2);
    compute(kt % 3);
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fr;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * 64 + i * 16 + fq * 4 + e;
        if (m < M) C[(size_t)m * ldc + n] = acc[i][j][e] + bv;
      }
    }
}

// ------------------------------------------------------------------------------------------
// exact fp32 GEMM: 128x128x16 tiles, v_mfma_f32_16x16x4_f32.  Each lane reads 4 consecutive k
// (one ds_read_b128) and feeds element j to the j-th MFMA; A and B use the same k permutation,
// so the sum is over the same 16 k values.
// ------------------------------------------------------------------------------------------
#define FK 16
#define FLD 20
template <int DUMMY>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(
    const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
    float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc) {
  __shared__ __attribute__((aligned(16))) float sa[BM * FLD];
  __shared__ __attribute__((aligned(16))) float sb[BN * FLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (N + BN - 1) / BN;
  const int ntm = (M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  for (int k0 = 0; k0 < K; k0 += FK) {
    float4 ra[2], rb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;          // 512 float4 per operand tile
      const int r = idx >> 2, c4 = idx & 3;
      int ar = m0 + r; if (ar > M - 1) ar = M - 1;
      int br = n0 + r; if (br > N - 1) br = N - 1;
      ra[i] = *(const float4*)(A + (size_t)ar * lda + k0 + c4 * 4);
      rb[i] = *(const float4*)(B + (size_t)br * ldb + k0 + c4 * 4);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      const int r = idx >> 2, c4 = idx & 3;
      *(float4*)(sa + r * FLD + c4 * 4) = ra[i];
      *(float4*)(sb + r * FLD + c4 * 4) = rb[i];
    }
    __syncthreads();
    float4 af[4], bfv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i] = *(const float4*)(sa + (wm * 64 + i * 16 + fr) * FLD + fq * 4);
      bfv[i] = *(const float4*)(sb + (wn * 64 + i * 16 + fr) * FLD + fq * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bfv[j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bfv[j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bfv[j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bfv[j].w, acc[i][j], 0, 0, 0);
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fr;
      if (n < N) {
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + wm * 64 + i * 16 + fq * 4 + e;
          if (m < M) C[(size_t)m * ldc + n] = acc[i][j][e] + bv;
        }
      }
    }
}

void launch_gemm_bf16_experimental(int variant, const void* A, int lda, const void* B, int ldb, const float* bias, float* C,
                                   int ldc, int M, int N, int K, hipStream_t s);
// bf16: requires N % 128 == 0, K % 64 == 0 (checked by the caller); M arbitrary (>0)
void launch_gemm_bf16_nt_epi(const void* A, int lda, const void* B, int ldb, const float* bias, float* C, int ldc, int M,
                             int N, int K, GemmEpi epi, hipStream_t s) {
  if (M <= 0) return;
  // whole-chip shapes: the 256x256 ping-pong kernel carries the same epilogues with 16-byte accesses (gemm_pp.hip)
  static const bool no_pp = prego_tune_env("PREGO_GEMM_NO_PINGPONG") != nullptr;
  if (M >= 4096 && !no_pp) {
    void* cdst = epi.mode == EPI_STORE_BF16 ? epi.out_b : (void*)C;
    if (launch_gemm_bf16_pingpong_epi(A, lda, B, ldb, bias, cdst, ldc, M, N, K, epi, s) == 0) return;
  }
  const int ntm = (M + BM - 1) / BM, ntn = N / BN;
  if (epi.f16) {                 // IEEE fp16 operands / 16-bit outputs
#define GL16(E)                                                                                                  \
  do {                                                                                                           \
    static DeviceOnce once;                                                                                      \
    once.run([] { (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<float, E, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); }); \
    gemm_bf16_nt_kernel<float, E, f16_t><<<ntm * ntn, 256, 65536, s>>>((const bf16_t*)A, (const bf16_t*)B, bias, C, M, N, K, lda, ldb, ldc, epi); \
  } while (0)
    switch (epi.mode) {
      case EPI_STORE: GL16(EPI_STORE); break;
      case EPI_RESIDUAL: GL16(EPI_RESIDUAL); break;
      case EPI_GELU_BF16: GL16(EPI_GELU_BF16); break;
      case EPI_STORE_BF16: GL16(EPI_STORE_BF16); break;
      case EPI_TOKENS: GL16(EPI_TOKENS); break;
      default: GL16(EPI_QKV); break;
    }
#undef GL16
    return;
  }
#define GL(E)                                                                                                   \
  do {                                                                                                          \
    static DeviceOnce once;                                                                                     \
    once.run([] { (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_kernel<float, E>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); }); \
    gemm_bf16_nt_kernel<float, E><<<ntm * ntn, 256, 65536, s>>>((const bf16_t*)A, (const bf16_t*)B, bias, C, M, N, K, lda, ldb, ldc, epi); \
  } while (0)
  switch (epi.mode) {
    case EPI_STORE: GL(EPI_STORE); break;
    case EPI_RESIDUAL: GL(EPI_RESIDUAL); break;
    case EPI_GELU_BF16: GL(EPI_GELU_BF16); break;
    case EPI_STORE_BF16: GL(EPI_STORE_BF16); break;
    case EPI_TOKENS: GL(EPI_TOKENS); break;
    default: GL(EPI_QKV); break;
  }
#undef GL
}
void launch_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, const float* bias, float* C, int ldc, int M,
                         int N, int K, hipStream_t s, bool f16, bool train_splitk) {
  if (f16) {                             // fp16 operands: ping-pong kernel for whole-chip shapes, the 128 x 128 kernel below
    GemmEpi e16{};
    e16.mode = EPI_STORE; e16.f16 = 1;
    launch_gemm_bf16_nt_epi(A, lda, B, ldb, bias, C, ldc, M, N, K, e16, s);
    return;
  }
  static const bool no_big = prego_tune_env("PREGO_GEMM_NO_BIG") != nullptr;
  static const bool no_splitk = prego_tune_env("PREGO_GEMM_NO_SPLITK") != nullptr;
  // at most one 128 x 128 tile per CU (the training step's projections: rows = 16 x 128): eight waves per workgroup, two K groups
  // (gemm_tn.hip) - 54 -> ~35 us where the 256 x 128 kernel left half of the CUs without a tile
  if (train_splitk && !no_splitk && ((M + 127) / 128) * ((N + 127) / 128) <= 256 && K >= 1024 && K % 64 == 0 &&
      launch_gemm_bf16_tn(false, false, A, lda, B, ldb, bias, C, ldc, M, N, K, K, nullptr, s) == 0)
    return;
  if (M >= 4096 && N % 256 == 0 && !no_big) {   // 256x256 tiles: ping-pong 8-phase schedule with 16-byte stores (gemm_pp.hip)
    static const bool no_pp = prego_tune_env("PREGO_GEMM_NO_PINGPONG") != nullptr;      // A/B knob: the previous production kernel
    if (no_pp || launch_gemm_bf16_pingpong(A, lda, B, ldb, bias, C, ldc, M, N, K, s) != 0)
      launch_gemm_bf16_experimental(9, A, lda, B, ldb, bias, C, ldc, M, N, K, s);
    return;
  }
  if (M >= 2048 && !no_big) {            // enough 256-row tiles to fill the chip
    static DeviceOnce once;
    once.run([] { (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * GSTAGE); });
    const int ntm = (M + GBM - 1) / GBM, ntn = N / BN;
    gemm_bf16_nt_big_kernel<<<ntm * ntn, 512, 3 * GSTAGE, s>>>((const bf16_t*)A, (const bf16_t*)B, bias, C, M, N, K, lda, ldb, ldc);
    return;
  }
  GemmEpi epi{};
  epi.mode = EPI_STORE;
  launch_gemm_bf16_nt_epi(A, lda, B, ldb, bias, C, ldc, M, N, K, epi, s);
}

void launch_gemm_bf16_variant(int variant, const void* A, int lda, const void* B, int ldb, const float* bias, float* C, int ldc,
                              int M, int N, int K, hipStream_t s) {
  if (variant == 0) {
    GemmEpi epi{}; epi.mode = EPI_STORE;
    launch_gemm_bf16_nt_epi(A, lda, B, ldb, bias, C, ldc, M, N, K, epi, s);
  } else if (variant == 1) {
    (void)hipFuncSetAttribute((const void*)gemm_bf16_nt_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * GSTAGE);
    const int ntm = (M + GBM - 1) / GBM, ntn = N / BN;
    gemm_bf16_nt_big_kernel<<<ntm * ntn, 512, 3 * GSTAGE, s>>>((const bf16_t*)A, (const bf16_t*)B, bias, C, M, N, K, lda, ldb, ldc);
  } else if (variant == 30) {        // the eight-wave split-K workgroup of gemm_tn.hip on nn.Linear's form (<= 256 tiles)
    (void)launch_gemm_bf16_tn(false, false, A, lda, B, ldb, bias, C, ldc, M, N, K, K, nullptr, s);
  } else {
    launch_gemm_bf16_experimental(variant, A, lda, B, ldb, bias, C, ldc, M, N, K, s);
  }
}

// fp32: K % 16 == 0; M, N arbitrary
void launch_gemm_f32_nt(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc, int M,
                        int N, int K, hipStream_t s) {
  if (M <= 0) return;
  const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
  gemm_f32_nt_kernel<0><<<ntm * ntn, 256, 0, s>>>(A, B, bias, C, M, N, K, lda, ldb, ldc);
}
