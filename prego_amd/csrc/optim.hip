// torch.optim.AdamW (step_recognition/main.py:62-67: lr 1e-4, weight_decay 0.05, default betas / eps, amsgrad off) as ONE
// fused multi-tensor launch: decoupled weight decay, both moment updates, bias corrections and the parameter update in a single
// pass over (p, g, m, v), and - for a model handle - the refreshed MFMA-operand copy of the parameter (bf16 or fp32) written
// from the same registers, so that the training step needs no per-step weight re-ingest (prego_miniroad_set_weights re-reads all
// 17.9 M parameters and re-converts them: four more passes over the weights).
//   p *= 1 - lr wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)            (torch/optim/adamw.py, single-tensor form)
// The tensor table travels BY VALUE in the kernel arguments (no host staging buffer, nothing to keep alive).
#include "common.h"
#include "kernels.h"

#define ADAM_MAX_TENSORS 24
#define ADAM_ELEMS_PER_BLOCK (256 * 4 * 4)      // 256 lanes x float4 x 4

struct AdamDesc {
  float* p; const float* g; float* m; float* v;
  void* copy;                 // nullable: operand copy of the updated parameter
  long long n;
  int block0;                 // first block of this tensor
  short copy_bf16;
  short vec;                  // 16-byte accesses allowed: n % 4 == 0 and every base pointer 16-byte aligned (checked on the host:
                              // the generic ABI takes arbitrary pointers, e.g. views at odd storage offsets)
};
struct AdamTable { AdamDesc d[ADAM_MAX_TENSORS]; int n; };

// guard (nullable): a device word; while it is non-zero the launch changes nothing.  prego_miniroad_adamw_step passes the handle's
// timeout word: a step whose recurrence / BPTT gave up (garbage gradients) then leaves parameters, moments and operand copies as they
// were, whether or not the host has looked at the word yet - the training loop needs no synchronisation in front of optimizer.step()
__global__ __launch_bounds__(256) void adamw_kernel(AdamTable tab, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                    float bc2_sqrt, unsigned* __restrict__ guard, const float* __restrict__ peer) {
  if (guard && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  // peer (nullable): the ranks' timeout flags summed by the gradient all-reduce (prego_miniroad_guard_publish).  Non-zero (or NaN) = some
  // rank's step gave up: every rank skips the step and raises its own word, so that every rank's check() reports it at the same step
  if (peer && !(*peer == 0.0f)) {
    if (guard && blockIdx.x == 0 && threadIdx.x == 0) atomicCAS(guard, 0u, 0x200u);
    return;
  }
  int ti = 0;
  for (int i = 1; i < tab.n; ++i) if ((int)blockIdx.x >= tab.d[i].block0) ti = i;      // <= 24 scalar compares
  const AdamDesc d = tab.d[ti];
  const long long base = (long long)(blockIdx.x - d.block0) * ADAM_ELEMS_PER_BLOCK;
  const float decay = 1.0f - lr * wd, step_size = lr / bc1, omb1 = 1.0f - b1, omb2 = 1.0f - b2;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long i = base + ((long long)u * 256 + threadIdx.x) * 4;
    if (i >= d.n) break;
    float pv[4], gv[4], mv[4], vv[4];
    const bool full = d.vec != 0 && i + 4 <= d.n;                  // 16-byte path (size a multiple of 4 floats, aligned bases)
    if (full) {
      const float4 a = *(const float4*)(d.p + i), b = *(const float4*)(d.g + i), c = *(const float4*)(d.m + i), e = *(const float4*)(d.v + i);
      pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w; gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
      mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w; vv[0] = e.x; vv[1] = e.y; vv[2] = e.z; vv[3] = e.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool ok = i + k < d.n;
        pv[k] = ok ? d.p[i + k] : 0.f; gv[k] = ok ? d.g[i + k] : 0.f; mv[k] = ok ? d.m[i + k] : 0.f; vv[k] = ok ? d.v[i + k] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pv[k] *= decay;
      mv[k] = b1 * mv[k] + omb1 * gv[k];
      vv[k] = b2 * vv[k] + omb2 * gv[k] * gv[k];
      pv[k] -= step_size * (mv[k] / (sqrtf(vv[k]) / bc2_sqrt + eps));
    }
    if (full) {
      *(float4*)(d.p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *(float4*)(d.m + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *(float4*)(d.v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      if (d.copy) {
        if (d.copy_bf16) { uint2 w; w.x = pack_bf16x2(pv[0], pv[1]); w.y = pack_bf16x2(pv[2], pv[3]); *(uint2*)((bf16_t*)d.copy + i) = w; }
        else *(float4*)((float*)d.copy + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (i + k < d.n) {
          d.p[i + k] = pv[k]; d.m[i + k] = mv[k]; d.v[i + k] = vv[k];
          if (d.copy) { if (d.copy_bf16) ((bf16_t*)d.copy)[i + k] = f2bf(pv[k]); else ((float*)d.copy)[i + k] = pv[k]; }
        }
    }
  }
}

// copies[i] nullable; copy_bf16: element type of every non-NULL copy.  Returns 0, or -1 on a bad argument.
int launch_adamw(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 void* const* copies, const long long* numel, bool copy_bf16, long long step, float lr, float b1, float b2, float eps,
                 float wd, hipStream_t s, unsigned* guard, const float* peer) {
  if (n_tensors <= 0 || step <= 0) return -1;
  const float bc1 = 1.0f - (float)pow((double)b1, (double)step);
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)step));
  for (int t0 = 0; t0 < n_tensors; t0 += ADAM_MAX_TENSORS) {
    AdamTable tab;
    tab.n = n_tensors - t0 < ADAM_MAX_TENSORS ? n_tensors - t0 : ADAM_MAX_TENSORS;
    int blocks = 0;
    for (int i = 0; i < tab.n; ++i) {
      const int k = t0 + i;
      if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k] || numel[k] <= 0) return -1;
      void* cp = copies ? copies[k] : nullptr;
      const uintptr_t bits = (uintptr_t)params[k] | (uintptr_t)grads[k] | (uintptr_t)exp_avg[k] | (uintptr_t)exp_avg_sq[k] | (uintptr_t)cp;
      const short vec = ((bits & 15) == 0 && (numel[k] & 3) == 0) ? 1 : 0;
      if (bits & 3) return -1;                                   // not even float-aligned
      tab.d[i] = AdamDesc{params[k], grads[k], exp_avg[k], exp_avg_sq[k], cp, numel[k], blocks, (short)(copy_bf16 ? 1 : 0), vec};
      blocks += (int)((numel[k] + ADAM_ELEMS_PER_BLOCK - 1) / ADAM_ELEMS_PER_BLOCK);
    }
    adamw_kernel<<<blocks, 256, 0, s>>>(tab, lr, b1, b2, eps, wd, bc1, bc2_sqrt, guard, peer);
  }
  return 0;
}

__global__ void guard_publish_kernel(const unsigned* __restrict__ abort_word, float* __restrict__ dst) {
  dst[0] = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1.0f : 0.0f;
}
void launch_guard_publish(const unsigned* abort_word, float* dst, hipStream_t s) { guard_publish_kernel<<<1, 1, 0, s>>>(abort_word, dst); }
