// Persistent GRU recurrence for MiniROAD (nn.GRU(2048,1024,1,batch_first) at model/rnn/rnn.py:38,61):
//   gh = h W_hh^T + b_hh;  r = s(gi_r+gh_r);  z = s(gi_z+gh_z);  n = tanh(gi_n + r*gh_n);  h' = (1-z) n + z h
// The input projection gi (with b_ih, and b_hh for the r/z rows, already added) comes from the GEMM.
//
// MI355X design.  T sequential steps, each a [clips x H] x [H x 3H] product, is latency bound, so:
//  * W_hh never leaves the register file.  The 256 CUs are split into G independent groups of P
//    workgroups (bf16: P = 32, G = 8; fp32: P = 64, G = 4).  A workgroup owns 16*UT hidden units
//    (all three gates) and holds its 3*16*UT x H weight slice as MFMA A-fragments in VGPRs: wave q
//    (one wave per SIMD, 512-VGPR budget) keeps the K-quarter [q*H/4, (q+1)*H/4) = 192 VGPRs.
//  * Clips are independent (h0 = 0 per clip, rnn.py:49,60): sorted clip i runs on group i % G,
//    slot i / G, so every group advances its own <= 16*NCT clips and groups never talk.
//  * Per step a workgroup needs the whole h_{t-1} of its group: an all-gather inside the group through
//    a double-buffered exchange buffer in global memory, published with write-through (sc1) stores +
//    one flag per producer, consumed with sc1 loads straight into MFMA B-fragments
//    (cdna_hip_programming.md Guideline 16, form R1; every load of handed-off bytes is an sc1 load).
//    Wave q only waits for the producers of its own K-quarter.
//  * fp32 state: h lives in registers of the lane that owns (unit, clip); only the MFMA operand copy
//    is rounded to bf16.
//  * every spin is bounded; a timeout raises an abort word that ends the launch (no hung GPU).
#include "common.h"
#include "kernels.h"


#define SPIN_LIMIT (1u << 21)

template <typename WT, int HID, int UT, int NCT>
__global__ __launch_bounds__(256, 1) void gru_recurrence_kernel(GruArgs a) {
  constexpr bool BF = (sizeof(WT) == 2);
  constexpr int UNITS = 16 * UT;              // hidden units owned by this workgroup
  constexpr int P = HID / UNITS;              // workgroups per group
  constexpr int KQ = HID / 4;                 // K range per wave
  constexpr int PPW = KQ / UNITS;             // producers a wave depends on
  constexpr int CPG = 16 * NCT;               // clip slots per group
  constexpr int NT = UT * NCT;                // 16x16 output tiles per gate
  constexpr int OWN_T = NT >= 4 ? NT / 4 : 1; // gate-phase tiles per wave
  constexpr int OWN_R = NT >= 4 ? 4 : NT;     // accumulator registers per owned tile
  constexpr int NKS = BF ? KQ / 32 : KQ / 16; // MFMA k-steps (bf16: 32 deep; f32: 4x4 deep)
  static_assert(PPW <= 32, "poll lanes");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = (f32x4*)smem;                  // [4 waves][3 gates][NT][64 lanes]

  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x % a.G;             // group (blocks b, b+8 share an XCD: a group stays on few XCDs)
  const int w = blockIdx.x / a.G;             // member of the group
  if (g >= a.n_clips) return;                 // group without clips
  const int l15 = lane & 15, l4 = lane >> 4;

  // ---- resident weights -------------------------------------------------------------------
  bf16x8 wb[BF ? 3 : 1][BF ? UT : 1][BF ? NKS : 1];
  float wf[BF ? 1 : 3][BF ? 1 : UT][BF ? 1 : NKS][4];
#pragma unroll
  for (int gate = 0; gate < 3; ++gate)
#pragma unroll
    for (int ut = 0; ut < UT; ++ut)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const size_t row = (size_t)gate * HID + w * UNITS + ut * 16 + l15;
        if constexpr (BF) {
          const int k = q * KQ + ks * 32 + 8 * l4;
          wb[gate][ut][ks] = *(const bf16x8*)((const bf16_t*)a.whh + row * HID + k);
        } else {
          const int k = q * KQ + ks * 16 + 4 * l4;
          const float4 v = *(const float4*)((const float*)a.whh + row * HID + k);
          wf[gate][ut][ks][0] = v.x; wf[gate][ut][ks][1] = v.y; wf[gate][ut][ks][2] = v.z; wf[gate][ut][ks][3] = v.w;
        }
      }

  // ---- gate-phase ownership ---------------------------------------------------------------
  int own_tile[OWN_T];
  int own_r0;
  if constexpr (NT >= 4) {
#pragma unroll
    for (int i = 0; i < OWN_T; ++i) own_tile[i] = q + 4 * i;
    own_r0 = 0;
  } else if constexpr (NT == 2) {
    own_tile[0] = q & 1; own_r0 = (q >> 1) * 2;
  } else {
    own_tile[0] = 0; own_r0 = q;
  }
  float hreg[OWN_T][OWN_R];
  float bhn[OWN_T][OWN_R];
  int sidx[OWN_T];   // sorted clip index of my clip per owned tile (or >= n_clips)
  int ucol[OWN_T];   // first global hidden unit of my registers
  int slot[OWN_T];
#pragma unroll
  for (int i = 0; i < OWN_T; ++i) {
    const int ut = own_tile[i] / NCT, ct = own_tile[i] % NCT;
    slot[i] = ct * 16 + l15;
    sidx[i] = slot[i] * a.G + g;
    ucol[i] = w * UNITS + ut * 16 + l4 * 4 + own_r0;
#pragma unroll
    for (int e = 0; e < OWN_R; ++e) {
      hreg[i][e] = (sidx[i] < a.n_clips) ? a.h_state[(size_t)sidx[i] * HID + ucol[i] + e] : 0.f;
      bhn[i][e] = a.b_hn[ucol[i] + e];
    }
  }

  // exchange buffer of my group: [2][CPG][HID] WT
  char* hx_base = (char*)a.hx + (size_t)g * 2 * CPG * HID * sizeof(WT);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)hx_base, 0, 2 * CPG * HID * (int)sizeof(WT), 0x00020000);
  unsigned* gflags = a.flags + g * P;

  auto publish = [&](int buf, unsigned epoch, bool all_slots) {
    // write my h (MFMA-operand precision) into hx[buf], write-through
#pragma unroll
    for (int i = 0; i < OWN_T; ++i) {
      if (all_slots || sidx[i] < a.n_clips) {
        const int off = ((buf * CPG + slot[i]) * HID + ucol[i]) * (int)sizeof(WT);
        if constexpr (BF) {
          if constexpr (OWN_R == 4) {
            u32x2 v = {pack_bf16x2(hreg[i][0], hreg[i][1]), pack_bf16x2(hreg[i][2], hreg[i][3])};
            __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, AUX_SC1);
          } else if constexpr (OWN_R == 2) {
            __builtin_amdgcn_raw_buffer_store_b32(pack_bf16x2(hreg[i][0], hreg[i][1]), rs, off, 0, AUX_SC1);
          } else {
            __builtin_amdgcn_raw_buffer_store_b16(f2bf(hreg[i][0]), rs, off, 0, AUX_SC1);
          }
        } else {
          if constexpr (OWN_R == 4) {
            u32x4 v = {__float_as_uint(hreg[i][0]), __float_as_uint(hreg[i][1]), __float_as_uint(hreg[i][2]), __float_as_uint(hreg[i][3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, AUX_SC1);
          } else if constexpr (OWN_R == 2) {
            u32x2 v = {__float_as_uint(hreg[i][0]), __float_as_uint(hreg[i][1])};
            __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, AUX_SC1);
          } else {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hreg[i][0]), rs, off, 0, AUX_SC1);
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains (R1)
    __syncthreads();
    if (tid == 0) __hip_atomic_store(gflags + w, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // prologue: h_{t0-1} goes to buffer 1, epoch 1
  publish(1, 1u, true);

  const int nsteps = a.t1 - a.t0;
  for (int tl = 0; tl < nsteps; ++tl) {
    const int t = a.t0 + tl;
    const int na = a.nact[t];
    const int rbase = a.rowoff[t] - a.row_base;

    // (1) prefetch gi for my elements (independent of h)
    float gir[OWN_T][3][OWN_R];
#pragma unroll
    for (int i = 0; i < OWN_T; ++i) {
      const bool act = sidx[i] < na;
#pragma unroll
      for (int gate = 0; gate < 3; ++gate) {
        const float* p = a.gi + (size_t)(rbase + (act ? sidx[i] : 0)) * (3 * HID) + gate * HID + ucol[i];
        if constexpr (OWN_R == 4) {
          const float4 v = act ? nt_load4(p) : make_float4(0, 0, 0, 0);
          gir[i][gate][0] = v.x; gir[i][gate][1] = v.y; gir[i][gate][2] = v.z; gir[i][gate][3] = v.w;
        } else {
#pragma unroll
          for (int e = 0; e < OWN_R; ++e) gir[i][gate][e] = act ? p[e] : 0.f;
        }
      }
    }

    // (2) wait for the producers of my K-quarter: flags >= tl + 1
    {
      const unsigned need = (unsigned)tl + 1u;
      unsigned spins = 0;
      for (;;) {
        unsigned f = need;
        if (lane < PPW) f = __hip_atomic_load(gflags + q * PPW + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (lane == PPW) f = __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0xFFFFFFFFu : need;
        const bool aborted = __any(lane == PPW && f == 0xFFFFFFFFu);
        if (aborted) return;
        if (__all(f >= need)) break;
        if (++spins > SPIN_LIMIT) {
          if (lane == 0) __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }

    // (3) h_{t-1} fragments (sc1 loads, straight to registers) and the MFMA K-quarter
    const int rbuf = (tl + 1) & 1;
    f32x4 acc[3][UT][NCT];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
      for (int ut = 0; ut < UT; ++ut)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[gate][ut][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      u32x4 hb[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const int k = BF ? (q * KQ + ks * 32 + 8 * l4) : (q * KQ + ks * 16 + 4 * l4);
        const int off = ((rbuf * CPG + ct * 16 + l15) * HID + k) * (int)sizeof(WT);
        hb[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX_SC1);
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if constexpr (BF) {
          const bf16x8 bfrag = __builtin_bit_cast(bf16x8, hb[ks]);
#pragma unroll
          for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
              acc[gate][ut][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[gate][ut][ks], bfrag, acc[gate][ut][ct], 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float bj = __uint_as_float(hb[ks][j]);
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
#pragma unroll
              for (int ut = 0; ut < UT; ++ut)
                acc[gate][ut][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[gate][ut][ks][j], bj, acc[gate][ut][ct], 0, 0, 0);
          }
        }
      }
    }

    // (4) cross-wave (K-quarter) reduction through LDS
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
      for (int ut = 0; ut < UT; ++ut)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
          red[((q * 3 + gate) * NT + ut * NCT + ct) * 64 + lane] = acc[gate][ut][ct];
    __syncthreads();

    // (5) gates and state update for the elements this lane owns
#pragma unroll
    for (int i = 0; i < OWN_T; ++i) {
      float gh[3][4];
#pragma unroll
      for (int gate = 0; gate < 3; ++gate) {
        f32x4 s = red[((0 * 3 + gate) * NT + own_tile[i]) * 64 + lane];
#pragma unroll
        for (int qq = 1; qq < 4; ++qq) s += red[((qq * 3 + gate) * NT + own_tile[i]) * 64 + lane];
        gh[gate][0] = s[0]; gh[gate][1] = s[1]; gh[gate][2] = s[2]; gh[gate][3] = s[3];
      }
      if (sidx[i] < na) {
#pragma unroll
        for (int e = 0; e < OWN_R; ++e) {
          const int re = own_r0 + e;   // register index inside the tile (compile-time for NT>=4)
          float ghr, ghz, ghn;
          if constexpr (NT >= 4) { ghr = gh[0][e]; ghz = gh[1][e]; ghn = gh[2][e]; }
          else {
            ghr = re == 0 ? gh[0][0] : re == 1 ? gh[0][1] : re == 2 ? gh[0][2] : gh[0][3];
            ghz = re == 0 ? gh[1][0] : re == 1 ? gh[1][1] : re == 2 ? gh[1][2] : gh[1][3];
            ghn = re == 0 ? gh[2][0] : re == 1 ? gh[2][1] : re == 2 ? gh[2][2] : gh[2][3];
          }
          const float r = sigmoidf_(gir[i][0][e] + ghr);
          const float z = sigmoidf_(gir[i][1][e] + ghz);
          const float n = tanhf_(gir[i][2][e] + r * (ghn + bhn[i][e]));
          hreg[i][e] = (1.0f - z) * n + z * hreg[i][e];
        }
      }
    }

    // (6) publish h_t for the group (also the barrier that protects `red` for the next step)
    publish(tl & 1, (unsigned)tl + 2u, false);

    // (7) outputs, off the critical path
#pragma unroll
    for (int i = 0; i < OWN_T; ++i) {
      if (sidx[i] < na) {
        const size_t o = (size_t)(rbase + sidx[i]) * HID + ucol[i];
        if (a.h_relu_out) {
          if constexpr (BF) {
            bf16_t* p = (bf16_t*)a.h_relu_out + o;
            if constexpr (OWN_R == 4) {
              uint2 v; v.x = pack_bf16x2(fmaxf(hreg[i][0], 0.f), fmaxf(hreg[i][1], 0.f));
              v.y = pack_bf16x2(fmaxf(hreg[i][2], 0.f), fmaxf(hreg[i][3], 0.f));
              *(uint2*)p = v;
            } else {
#pragma unroll
              for (int e = 0; e < OWN_R; ++e) p[e] = f2bf(fmaxf(hreg[i][e], 0.f));
            }
          } else {
            float* p = (float*)a.h_relu_out + o;
#pragma unroll
            for (int e = 0; e < OWN_R; ++e) p[e] = fmaxf(hreg[i][e], 0.f);
          }
        }
        if (a.h_raw_out) {
#pragma unroll
          for (int e = 0; e < OWN_R; ++e) a.h_raw_out[o + e] = hreg[i][e];
        }
      }
    }
  }

  // final state back to h_state (streaming / next chunk)
#pragma unroll
  for (int i = 0; i < OWN_T; ++i)
    if (sidx[i] < a.n_clips) {
#pragma unroll
      for (int e = 0; e < OWN_R; ++e) a.h_state[(size_t)sidx[i] * HID + ucol[i] + e] = hreg[i][e];
    }
}

// Returns 0 on success, -1 for unsupported (hid, nct).  flags must hold G*P words and is zeroed here.
int launch_gru_recurrence(bool bf16, int hid, int nct, GruArgs a, hipStream_t s) {
  if (hid != 1024) return -1;
  const int P = bf16 ? 32 : 64;
  (void)hipMemsetAsync(a.flags, 0, (size_t)a.G * P * sizeof(unsigned), s);
  const int grid = a.G * P;
#define LAUNCH(WT, UT, NCT)                                                                            \
  do {                                                                                                 \
    const size_t lds = (size_t)4 * 3 * (UT * NCT) * 64 * 16;                                           \
    (void)hipFuncSetAttribute((const void*)gru_recurrence_kernel<WT, 1024, UT, NCT>,                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    gru_recurrence_kernel<WT, 1024, UT, NCT><<<grid, 256, lds, s>>>(a);                                \
  } while (0)
  if (bf16) {
    if (nct == 1) LAUNCH(bf16_t, 2, 1);
    else if (nct == 2) LAUNCH(bf16_t, 2, 2);
    else if (nct == 4) LAUNCH(bf16_t, 2, 4);
    else return -1;
  } else {
    if (nct == 1) LAUNCH(float, 1, 1);
    else if (nct == 2) LAUNCH(float, 1, 2);
    else if (nct == 4) LAUNCH(float, 1, 4);
    else return -1;
  }
#undef LAUNCH
  return 0;
}
