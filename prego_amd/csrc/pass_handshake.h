// Device side of the split pass's start handshake (kernels.h: PassHandshake).  Included by ff_pass.hip and gru_recurrence.hip.
#pragma once
#include "common.h"
#include "kernels.h"

// one lane per workgroup calls these
__device__ __forceinline__ void hs_decide(const PassHandshake& h, unsigned state) {
  unsigned expected = 0u;
  if (__hip_atomic_compare_exchange_strong(h.word, &expected, state, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
    if (h.host) __hip_atomic_store(h.host, (h.seq << 2) | state, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// the decision (PREGO_HS_GO / PREGO_HS_FAIL); waits `ticks` of s_memrealtime at most, then votes FAIL and takes whatever was decided
__device__ __forceinline__ unsigned hs_wait(const PassHandshake& h, unsigned ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const unsigned v = __hip_atomic_load(h.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v) return v;
    if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)ticks) hs_decide(h, PREGO_HS_FAIL);
    else __builtin_amdgcn_s_sleep(16);
  }
}

