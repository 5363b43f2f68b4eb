// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Wave = 64 lanes, 4 SIMDs per CU, MFMA 16x16x32 bf16 / 16x16x4 f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(3))) unsigned u32x3;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef unsigned short bf16_t;                               // raw bf16 bits in memory

#define PREGO_WAVE 64
#define AUX_NT 2            // non-temporal: streams through (bypasses) the per-CU L1, L2-served
#define AUX_SC1 16          // buffer/global aux bit: system-coherent level 1 = bypass the per-CU L1

// fp32 -> bf16, round to nearest even.  A plain cast lowers to v_cvt_pk_bf16_f32 on gfx950
// (NaN-safe, MI355X_MICROARCH.md "Correctness boundaries").
__device__ __forceinline__ bf16_t f2bf(float x) {
  __bf16 b = (__bf16)x;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
  return __builtin_bit_cast(float, (unsigned)b << 16);
}
// two floats -> packed bf16x2 as ONE v_cvt_pk_bf16_f32 (the scalar form is two converts + an or_sdwa); same RNE rounding
typedef __bf16 bf16x2_v_ __attribute__((ext_vector_type(2)));
typedef float f32x2_v_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_v_){lo, hi}, bf16x2_v_));
}

// ---- 16-bit MFMA operand types ---------------------------------------------------------------------------------------
// bf16 (8 mantissa bits, fp32 range) or IEEE fp16 (11 mantissa bits, |x| <= 65504): the matrix pipe runs both at the same rate
// and every kernel accumulates in fp32, so fp16 operands cut the operand rounding error 8x at no cost in time (profiles/
// precision_study_r03.json: bf16 operands flip 9 argmaxes above a 1e-3 margin on the fixtures, fp16 operands none).  Memory is raw
// 16-bit words either way (bf16_t pointers); the tag type OT only selects conversions and the MFMA opcode.
typedef _Float16 f16_t;                                       // fp16 operand tag (distinct from bf16_t = unsigned short)
typedef _Float16 f16x8_v_ __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_v_ __attribute__((ext_vector_type(2)));
#define PREGO_F16_MAX 65504.0f
template <typename OT> struct op16;
template <> struct op16<bf16_t> {
  static constexpr bool is_f16 = false;
  static __device__ __forceinline__ unsigned pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
  static __device__ __forceinline__ unsigned pack2_sat(float lo, float hi) { return pack_bf16x2(lo, hi); }   // fp32 range: nothing to saturate
  static __device__ __forceinline__ bf16_t cvt(float x) { return f2bf(x); }
  static __device__ __forceinline__ bf16_t cvt_sat(float x) { return f2bf(x); }
  static __device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
  static __device__ __forceinline__ float up(bf16_t b) { return bf2f(b); }
  static __device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct op16<f16_t> {
  static constexpr bool is_f16 = true;
  static __device__ __forceinline__ unsigned pack2(float lo, float hi) {        // one v_cvt_pk_f16_f32 (RNE); caller guarantees the range
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_v_){lo, hi}, f16x2_v_));
  }
  static __device__ __forceinline__ unsigned pack2_sat(float lo, float hi) {    // saturate instead of overflowing to inf
    return pack2(__builtin_amdgcn_fmed3f(lo, -PREGO_F16_MAX, PREGO_F16_MAX), __builtin_amdgcn_fmed3f(hi, -PREGO_F16_MAX, PREGO_F16_MAX));
  }
  static __device__ __forceinline__ bf16_t cvt(float x) { return __builtin_bit_cast(bf16_t, (_Float16)x); }
  static __device__ __forceinline__ bf16_t cvt_sat(float x) { return cvt(__builtin_amdgcn_fmed3f(x, -PREGO_F16_MAX, PREGO_F16_MAX)); }
  static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(f16x2_v_, w)[0]; }
  static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(f16x2_v_, w)[1]; }
  static __device__ __forceinline__ float up(bf16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
  static __device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_v_, a), __builtin_bit_cast(f16x8_v_, b), c, 0, 0, 0);
  }
};
// ---- split operands ("fp16x2", PREGO_F16X2): a value v travels as TWO fp16 numbers, hi = fp16(v) and lo = fp16(v - hi), i.e. ~22
// mantissa bits, and a product a.b is taken as a_hi.b_lo + a_lo.b_hi + a_hi.b_hi on the fp16 matrix pipe with fp32 accumulation
// (the dropped a_lo.b_lo term is 2^-22 relative: fp32 rounding noise).  Three 16-bit MFMA products = 3/16 of the cost of the
// exact-fp32 MFMA at the same error class: the argmax-identical mode of the north star (rnn.py:58-70) without the 8.6x of
// compute_dtype = 'fp32'.  Weights are pre-scaled by a power of two (exact) so that their lo halves stay in fp16's normal range
// (|w| ~ 1e-2 would put w_lo ~ 2^-19 among the subnormals: 18 instead of 22 bits; the fp64-referenced emulation moves from
// 1.1e-5 to 1.7e-6 worst probability error with the scale, scripts/precision_study.py); the epilogue multiplies the scale out.
// Memory layout of a split row of n elements: [n hi | n lo] (raw fp16 bits), same bytes as the fp32 row.
struct x2_t { unsigned v; };                                  // tag type only (sizeof 4 = bytes per split element)
template <typename T> struct is_x2 { static constexpr bool value = false; };
template <> struct is_x2<x2_t> { static constexpr bool value = true; };
// two floats -> (hi pair, lo pair), both as packed fp16x2 words; hi saturates at +-65504
__device__ __forceinline__ void x2_split2(float a, float b, unsigned& hi, unsigned& lo) {
  hi = op16<f16_t>::pack2_sat(a, b);
  const float ra = a - op16<f16_t>::lo(hi), rb = b - op16<f16_t>::hi(hi);
  lo = op16<f16_t>::pack2_sat(ra, rb);
}

// operand tag of an output element type: float stays float, 16-bit types name themselves
template <typename T> struct is_op16 { static constexpr bool value = sizeof(T) == 2; };

// streaming (read-once) 16-byte load
__device__ __forceinline__ float4 nt_load4(const float* p) {
  const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
  return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// v_exp_f32 + v_rcp_f32 (1 ulp each): the recurrence runs one wave per SIMD, every VALU instruction is ~4 cycles of
// the step's critical path, so no IEEE division sequences here
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh(x) = 2 sigmoid(2x) - 1: abs error ~1e-7, saturates cleanly (exp overflow -> rcp(inf) = 0)
__device__ __forceinline__ float tanhf_(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

// exact-erf GELU (nn.GELU default, Transformer.py:40)
__device__ __forceinline__ float gelu_erf_(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// stateless dropout mask: keep iff hash(seed, element index) >= p * 2^32 (same mask in forward and backward)
__device__ __forceinline__ unsigned mix32_(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (unsigned)x;
}
__device__ __forceinline__ bool dropout_keep_(unsigned long long seed, size_t idx, unsigned thresh) {
  return mix32_(seed * 0x9E3779B97F4A7C15ULL + idx) >= thresh;
}

// Packed time-major row layout (the cuDNN/PackedSequence idea, rebuilt for this path):
// clips are sorted by length, descending; at time t the first nact[t] sorted clips are alive;
// row(t, i) = rowoff[t] + i.  Everything between the pack kernel and the head kernel is
// indexed by packed row, so a chunk of time steps is a contiguous row range.
// Continuous batching: the units that are alive/sorted are SLOTS, not clips.  A slot runs a list of clips back to
// back (longest-processing-time packing on the host), so the number of sequential steps is max(longest clip,
// frames / slots) instead of the longest clip with most slots idle.  Slot i's clips are the segments
// seg_off[i] .. seg_off[i+1]-1: clip seg_clip[k] occupies steps [seg_start[k], seg_start[k+1]) of that slot.
// With one clip per slot this degenerates to the plain length-sorted packing.
#ifndef PREGO_HAVE_SLOTPLAN
#define PREGO_HAVE_SLOTPLAN 1
struct SlotPlan {
  const int* rowoff;      // [s_max + 1] prefix sum of nact
  const int* nact;        // [s_max]  live slots per step
  const int* seg_off;     // [n_slots + 1]
  const int* seg_clip;    // [n_clips] caller's clip index
  const int* seg_start;   // [n_clips] first step of the segment
  const int* blk_step;    // [ceil(rows / 32)] step of packed row 32 b (row -> step without a binary search: head kernel)
  int s_max;
  int n_slots;
};
#endif

// largest t with rowoff[t] <= row  (row < rowoff[t_max])
__device__ __forceinline__ int plan_time_of_row(const int* __restrict__ rowoff, int t_max, int row) {
  int lo = 0, hi = t_max;            // invariant: rowoff[lo] <= row < rowoff[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (rowoff[mid] <= row) lo = mid; else hi = mid;
  }
  return lo;
}

// packed row -> (caller's clip, frame inside the clip)
__device__ __forceinline__ void plan_clip_of_row(const SlotPlan& p, int row, int& clip, int& frame) {
  const int s = plan_time_of_row(p.rowoff, p.s_max, row);
  const int slot = row - p.rowoff[s];
  int k = p.seg_off[slot];
  const int kend = p.seg_off[slot + 1];
  while (k + 1 < kend && p.seg_start[k + 1] <= s) ++k;      // a handful of segments per slot
  clip = p.seg_clip[k];
  frame = s - p.seg_start[k];
}
