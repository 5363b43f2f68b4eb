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
//    a double-buffered exchange buffer in global memory.  THE DATA IS THE FLAG: |h| <= 1 for a GRU state,
//    so the top exponent bit of every bf16 (bit 14) / fp32 (bit 30) element is free and carries a one-bit
//    epoch tag ((step >> 1) & 1; with two buffers a stale element always shows the other value).  Producers
//    write-through (sc1) and never wait; consumers load the MFMA B-fragments with sc1 loads straight to
//    registers and simply re-load a fragment until every element in it shows the expected tag.  No flag, no
//    fence, no drain, no producer-side barrier, and no reliance on store ordering or granule atomicity: each
//    2/4-byte element validates itself.  Wave q only ever waits for the 8 producers of its own K-quarter.
//  * Where the data travels.  sc1 traffic is served by the fabric at ~10 B/clk/CU (measured: the gather of
//    32 KB per CU per step cost 3000 cycles), the XCD's own L2 is 5x faster.  So every launch first VERIFIES
//    placement: each workgroup reads HW_REG_XCC_ID, draws a ticket on its XCD's counter, and all meet at a
//    bounded rendezvous; if every XCD got exactly P workgroups, group := XCD and the hand-off uses plain stores
//    (the dirty line stays in the XCD's L2) + L1-bypassing `nt` loads (L2-served) - coherent because one XCD's
//    CUs share one L2.  Otherwise (other placement, fewer CUs, the 64-workgroup fp32 groups) it falls back to
//    sc1 stores + sc1 loads.  Either way the element tags decide validity, so correctness never depends on
//    placement - only speed does; stale-looking `nt` data escalates to sc1 loads after a few retries.
//  * fp32 state: h lives in registers of the lane that owns (unit, clip); only the MFMA operand copy is
//    rounded to bf16 (and saturated below 2.0 so that the tag bit stays free even for a hostile h0).
//  * gi for step t+1 is prefetched during step t; clip tiles whose clips have all ended are skipped.
//  * every spin is bounded; a timeout raises an abort word that ends the launch (no hung GPU).
#include "common.h"
#include "kernels.h"

#define SPIN_LIMIT (1u << 22)

template <typename WT, int HID, int UT, int NCT>
__global__ __launch_bounds__(256, 1) void gru_recurrence_kernel(GruArgs a) {
  constexpr bool BF = (sizeof(WT) == 2);
  constexpr int UNITS = 16 * UT;              // hidden units owned by this workgroup
  constexpr int KQ = HID / 4;                 // K range per wave
  constexpr int NT = UT * NCT;                // 16x16 output tiles per gate
  constexpr int OWN_T = NT >= 4 ? NT / 4 : 1; // gate-phase tiles per wave
  constexpr int OWN_R = NT >= 4 ? 4 : NT;     // accumulator registers per owned tile
  constexpr int NKS = BF ? KQ / 32 : KQ / 16; // fragments per clip tile (bf16: 32 k each; f32: 16 k each)
  constexpr int TG = (BF && NCT == 2) ? 2 : 1;   // clip tiles gathered together (register budget: otherwise tile by tile)
  constexpr int NRED = (NT <= 4) ? 2 : 1;     // LDS reduction buffers (two when they fit: one barrier per step)
  constexpr unsigned TAGM = BF ? 0x40004000u : 0x40000000u;
  constexpr int SLOTS = 64;                   // clip slots per group in the exchange buffer (max 16*NCT)
  // Exchange layout = MFMA B-fragment order: [k-step of 32 (bf16) / 16 (f32)][clip tile 0..3][lane 0..63][16 B], so a
  // consumer's fragment load is ONE contiguous 1 KiB (rows at a 2 KiB stride all fell on the same L2 channel and
  // ran at 10 B/clk/CU).  Element (clip slot c, hidden unit k): fragment (k / KF, c / 16), lane ((k % KF) / EPL) * 16
  // + c % 16, byte (k % EPL) * sizeof(WT), with KF = k per fragment, EPL = elements per lane.
  constexpr int KF = BF ? 32 : 16;
  constexpr int EPL = BF ? 8 : 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = (f32x4*)smem;                  // [NRED][4 waves][3 gates][NT][64 lanes]
  constexpr int RED_STRIDE = 4 * 3 * NT * 64;

  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- placement rendezvous: group := XCD when every XCD holds exactly P workgroups ------------
  constexpr int P = HID / UNITS;
  __shared__ int s_place[4];
  if (tid == 0) {
    int gg = blockIdx.x % a.G, ww = blockIdx.x / a.G, loc = 0;
    if (a.sync != nullptr && a.G == 8 && gridDim.x == 8 * P) {
      const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;          // HW_REG_XCC_ID[2:0]
      const unsigned ticket = __hip_atomic_fetch_add(a.sync + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(a.sync + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      bool ok = true;
      while (__hip_atomic_load(a.sync + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        if (++spins > SPIN_LIMIT) { ok = false; break; }
        __builtin_amdgcn_s_sleep(4);
      }
      if (!ok) { __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); loc = -1; }
      else {
        loc = 1;
        for (int i = 0; i < 8; ++i)
          if (__hip_atomic_load(a.sync + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)P) loc = 0;
        if (loc) { gg = xcc; ww = (int)ticket; }
      }
    }
    s_place[0] = gg; s_place[1] = ww; s_place[2] = loc;
  }
  __syncthreads();
  const int g = __builtin_amdgcn_readfirstlane(s_place[0]);   // group
  const int w = __builtin_amdgcn_readfirstlane(s_place[1]);   // member of the group
  const int place = __builtin_amdgcn_readfirstlane(s_place[2]);
  if (place < 0) return;                      // rendezvous timed out (abort word set)
  const bool local = place == 1;              // whole group on one XCD, verified
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
  int sidx[OWN_T];    // sorted clip index of my clip per owned tile (may be >= n_clips)
  int ucol[OWN_T];    // first global hidden unit of my registers
  int slot[OWN_T];
  int tfirst[OWN_T];  // sorted index of the first slot of the owned clip tile: tile in use at t iff tfirst < nact[t]
#pragma unroll
  for (int i = 0; i < OWN_T; ++i) {
    const int ut = own_tile[i] / NCT, ct = own_tile[i] % NCT;
    slot[i] = ct * 16 + l15;
    sidx[i] = slot[i] * a.G + g;
    tfirst[i] = ct * 16 * a.G + g;
    ucol[i] = w * UNITS + ut * 16 + l4 * 4 + own_r0;
#pragma unroll
    for (int e = 0; e < OWN_R; ++e) {
      hreg[i][e] = (sidx[i] < a.n_clips) ? a.h_state[(size_t)sidx[i] * HID + ucol[i] + e] : 0.f;
      bhn[i][e] = a.b_hn[ucol[i] + e];
    }
  }

  // exchange buffers: [2][G][SLOTS][HID] WT ; one descriptor per group, buffer index in the offset
  const int buf_stride = a.G * SLOTS * HID * (int)sizeof(WT);
  char* hx_base = (char*)a.hx + (size_t)g * SLOTS * HID * sizeof(WT);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)hx_base, 0, buf_stride + SLOTS * HID * (int)sizeof(WT), 0x00020000);

  // tagged MFMA-operand copy of one state element
  auto tag_bf = [](float x, unsigned tag) -> unsigned {
    x = fminf(fmaxf(x, -1.9921875f), 1.9921875f);
    return ((unsigned)f2bf(x) & 0xBFFFu) | (tag << 14);
  };
  auto tag_f32 = [](float x, unsigned tag) -> unsigned {
    x = fminf(fmaxf(x, -1.9999998f), 1.9999998f);
    return (__float_as_uint(x) & 0xBFFFFFFFu) | (tag << 30);
  };
  // publish my slice of h for the consumers of the next time step (tiles still in use then); no waiting
  auto publish = [&](int buf, unsigned tag, int na_next) {
#pragma unroll
    for (int i = 0; i < OWN_T; ++i) {
      if (tfirst[i] < na_next) {
        const int off = buf * buf_stride + ((ucol[i] / KF) * 4 + (slot[i] >> 4)) * 1024 + ((((ucol[i] % KF) / EPL) << 4) + (slot[i] & 15)) * 16 +
                        (ucol[i] % EPL) * (int)sizeof(WT);
        if constexpr (BF) {
          if constexpr (OWN_R == 4) {
            u32x2 v = {tag_bf(hreg[i][0], tag) | (tag_bf(hreg[i][1], tag) << 16), tag_bf(hreg[i][2], tag) | (tag_bf(hreg[i][3], tag) << 16)};
            { if (local) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 0); else __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, AUX_SC1); }
          } else if constexpr (OWN_R == 2) {
            { if (local) __builtin_amdgcn_raw_buffer_store_b32(tag_bf(hreg[i][0], tag) | (tag_bf(hreg[i][1], tag) << 16), rs, off, 0, 0); else __builtin_amdgcn_raw_buffer_store_b32(tag_bf(hreg[i][0], tag) | (tag_bf(hreg[i][1], tag) << 16), rs, off, 0, AUX_SC1); }
          } else {
            { if (local) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)tag_bf(hreg[i][0], tag), rs, off, 0, 0); else __builtin_amdgcn_raw_buffer_store_b16((unsigned short)tag_bf(hreg[i][0], tag), rs, off, 0, AUX_SC1); }
          }
        } else {
          if constexpr (OWN_R == 4) {
            u32x4 v = {tag_f32(hreg[i][0], tag), tag_f32(hreg[i][1], tag), tag_f32(hreg[i][2], tag), tag_f32(hreg[i][3], tag)};
            { if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0); else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, AUX_SC1); }
          } else if constexpr (OWN_R == 2) {
            u32x2 v = {tag_f32(hreg[i][0], tag), tag_f32(hreg[i][1], tag)};
            { if (local) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 0); else __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, AUX_SC1); }
          } else {
            { if (local) __builtin_amdgcn_raw_buffer_store_b32(tag_f32(hreg[i][0], tag), rs, off, 0, 0); else __builtin_amdgcn_raw_buffer_store_b32(tag_f32(hreg[i][0], tag), rs, off, 0, AUX_SC1); }
          }
        }
      }
    }
  };

  // gi loads are UNCONDITIONAL (inactive lanes read row 0 of the step, which always exists, and ignore it): a
  // "load or zero" select makes hipcc branch around every load and wait vmcnt(0) right behind it.
  auto load_gi = [&](float (&dst)[OWN_T][3][OWN_R], int na, int rbase) {
#pragma unroll
    for (int i = 0; i < OWN_T; ++i) {
      const int r = rbase + (sidx[i] < na ? sidx[i] : 0);
#pragma unroll
      for (int gate = 0; gate < 3; ++gate) {
        const float* p = a.gi + (size_t)r * (3 * HID) + gate * HID + ucol[i];
        if constexpr (OWN_R == 4) {
          const float4 v = nt_load4(p);
          dst[i][gate][0] = v.x; dst[i][gate][1] = v.y; dst[i][gate][2] = v.z; dst[i][gate][3] = v.w;
        } else if constexpr (OWN_R == 2) {
          const float2 v = *(const float2*)p;
          dst[i][gate][0] = v.x; dst[i][gate][1] = v.y;
        } else {
          dst[i][gate][0] = p[0];
        }
      }
    }
  };

  const bool stamp = a.stamps != nullptr && blockIdx.x == 0 && q == 0;
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long st_t = 0;
#define STAMP(i) do { if (stamp) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_t; st_t = n_; } } while (0)
  // the plan tables are read-only for the whole launch: constant address space = scalar (s_load) path, so the
  // look-ahead never touches the vector-memory queue (a vector load here drags a vmcnt wait through every step)
  typedef const __attribute__((address_space(4))) int* cint_p;
  cint_p nact_c = (cint_p)a.nact;
  cint_p rowoff_c = (cint_p)a.rowoff;
  const int nsteps = a.t1 - a.t0;
  // prologue: h_{t0-1} -> buffer 1 with the tag of step "-1" (= 1); gi of the first step
  // plan scalars one step ahead of their use (s_load latency off the critical path)
  int na_c = nact_c[a.t0], rb_c = rowoff_c[a.t0] - a.row_base;         // step tl
  int na_n = nsteps > 1 ? nact_c[a.t0 + 1] : 0, rb_n = nsteps > 1 ? rowoff_c[a.t0 + 1] - a.row_base : 0;   // step tl+1
  publish(1, 1u, na_c);
  float giA[OWN_T][3][OWN_R], giB[OWN_T][3][OWN_R];                     // ping-pong: no register copies
  load_gi(giA, na_c, rb_c);

  // one time step; gir = gi of this step (loaded a step ago), gin = where the next step's gi lands
  auto step = [&](const int tl, float (&gir)[OWN_T][3][OWN_R], float (&gin)[OWN_T][3][OWN_R]) -> bool {
    const int t = a.t0 + tl;
    const int na = na_c;
    const int rbase = rb_c;
    const bool more = tl + 1 < nsteps;
    const int t2 = (tl + 2 < nsteps) ? t + 2 : t;                       // look-ahead index (clamped)
    const int na_2 = nact_c[t2], rb_2 = rowoff_c[t2] - a.row_base;
    const int rbuf = (tl + 1) & 1;
    const unsigned etag = (unsigned)(((tl - 1) >> 1) & 1);      // tag of the step that produced h_{t-1}
    const unsigned eword = etag ? TAGM : 0u;

    if (stamp) st_t = __builtin_amdgcn_s_memtime();
    f32x4 acc[3][UT][NCT];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
      for (int ut = 0; ut < UT; ++ut)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[gate][ut][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- (1) gather h_{t-1} (data-is-the-flag) and multiply -------------------------------
#pragma unroll
    for (int c0 = 0; c0 < NCT; c0 += TG) {
      if (c0 * 16 * a.G + g < na) {                             // else: this and all later clip tiles are finished
        u32x4 hb[TG][NKS];
        unsigned pending = 0;
#pragma unroll
        for (int c = 0; c < TG; ++c)
          if ((c0 + c) * 16 * a.G + g < na) pending |= ((1u << NKS) - 1u) << (c * NKS);
        // Gather: load every fragment, re-load the ones that still show the old tag until all are valid; then
        // multiply in a FIXED order (bit-reproducible fp32 sums; consuming fragments in arrival order measured 7 %
        // faster per step but makes the summation order, and so the last bits, depend on timing).
        unsigned spins = 0;
        for (;;) {
#pragma unroll
          for (int c = 0; c < TG; ++c)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
              if ((pending >> (c * NKS + ks)) & 1u) {
                const int off = rbuf * buf_stride + ((q * NKS + ks) * 4 + (c0 + c)) * 1024 + lane * 16;
                hb[c][ks] = (local && spins < 6u) ? __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX_NT)
                                                  : __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX_SC1);
              }
          // fast check: every element of every requested fragment carries the expected tag?
          unsigned bad = 0;
#pragma unroll
          for (int c = 0; c < TG; ++c)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
              if ((pending >> (c * NKS + ks)) & 1u)
                bad |= (hb[c][ks][0] ^ eword) | (hb[c][ks][1] ^ eword) | (hb[c][ks][2] ^ eword) | (hb[c][ks][3] ^ eword);
          if (__all((bad & TAGM) == 0u)) break;
          // slow path: find out which fragments are still stale, re-load only those
#pragma unroll
          for (int c = 0; c < TG; ++c)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
              if ((pending >> (c * NKS + ks)) & 1u) {
                const unsigned b = ((hb[c][ks][0] ^ eword) | (hb[c][ks][1] ^ eword) | (hb[c][ks][2] ^ eword) | (hb[c][ks][3] ^ eword)) & TAGM;
                if (__all(b == 0u)) pending &= ~(1u << (c * NKS + ks));
              }
          if (pending == 0u) break;
          if (++spins > SPIN_LIMIT) {
            if (lane == 0) __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
          }
          if ((spins & 255u) == 0u) {
            if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
          }
        }
        if (stamp) st_acc[5] += spins;
        STAMP(0);
#pragma unroll
        for (int c = 0; c < TG; ++c) {
          if ((c0 + c) * 16 * a.G + g < na) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
              u32x4 v = hb[c][ks];
              if (etag) { v[0] &= ~TAGM; v[1] &= ~TAGM; v[2] &= ~TAGM; v[3] &= ~TAGM; }
              if constexpr (BF) {
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                  for (int ut = 0; ut < UT; ++ut)
                    acc[gate][ut][c0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[gate][ut][ks], bfrag, acc[gate][ut][c0 + c], 0, 0, 0);
              } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                  const float bj = __uint_as_float(v[jj]);
#pragma unroll
                  for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                    for (int ut = 0; ut < UT; ++ut)
                      acc[gate][ut][c0 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[gate][ut][ks][jj], bj, acc[gate][ut][c0 + c], 0, 0, 0);
                }
              }
            }
          }
        }
      }
    }
    STAMP(1);

    // ---- (1b) the gather's vmcnt(0) has just retired every older vector-memory op, including the loads of this
    // step's gi (issued one step ago).  Pin that fact for the compiler (it would otherwise put a vmcnt(0) in front of
    // the first use of the loop-carried registers, i.e. behind the prefetch issued next), then prefetch gi(t+1):
    // it has a whole step to land and is already old when the next gather waits.
#pragma unroll
    for (int i = 0; i < OWN_T; ++i)
#pragma unroll
      for (int gate = 0; gate < 3; ++gate)
#pragma unroll
        for (int e = 0; e < OWN_R; ++e) asm volatile("" : "+v"(gir[i][gate][e]));
    if (more) load_gi(gin, na_n, rb_n);

    // ---- (2) cross-wave (K-quarter) reduction through LDS ---------------------------------
    f32x4* redw = red + (NRED == 2 ? (tl & 1) * RED_STRIDE : 0);
    if constexpr (NRED == 1) __syncthreads();      // previous step's readers are done
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
      for (int ut = 0; ut < UT; ++ut)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
          if (ct * 16 * a.G + g < na) redw[((q * 3 + gate) * NT + ut * NCT + ct) * 64 + lane] = acc[gate][ut][ct];
    __syncthreads();
    STAMP(2);

    // ---- (3) gates + state update ----------------------------------------------------------
#pragma unroll
    for (int i = 0; i < OWN_T; ++i) {
      if (tfirst[i] < na) {
        float gh[3][4];
        f32x4 part[3][4];
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) part[gate][qq] = redw[((qq * 3 + gate) * NT + own_tile[i]) * 64 + lane];
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) {
          const f32x4 s = (part[gate][0] + part[gate][1]) + (part[gate][2] + part[gate][3]);
          gh[gate][0] = s[0]; gh[gate][1] = s[1]; gh[gate][2] = s[2]; gh[gate][3] = s[3];
        }
        if (sidx[i] < na) {
#pragma unroll
          for (int e = 0; e < OWN_R; ++e) {
            const int re = own_r0 + e;
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
            if (a.keep_r) {                                      // training: gate activations for BPTT
              const size_t ko = (size_t)(rbase + sidx[i]) * HID + ucol[i] + e;
              a.keep_r[ko] = r; a.keep_z[ko] = z; a.keep_n[ko] = n; a.keep_ghn[ko] = ghn + bhn[i][e];
            }
          }
        }
      }
    }

    // ---- (4) publish h_t for step t+1 (fire and forget) -----------------------------------
    if (more) publish(tl & 1, (unsigned)((tl >> 1) & 1), na_n);
    STAMP(3);

    // ---- (5) outputs ----------------------------------------------------------------------
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
    na_c = na_n; rb_c = rb_n; na_n = na_2; rb_n = rb_2;
    STAMP(4);
    return true;
  };
  for (int tl = 0; tl < nsteps; tl += 2) {
    if (!step(tl, giA, giB)) return;
    if (tl + 1 < nsteps && !step(tl + 1, giB, giA)) return;
  }
  if (stamp && lane == 0) {
    for (int i = 0; i < 6; ++i) a.stamps[i] += st_acc[i];
    a.stamps[6] += (unsigned long long)nsteps;
  }
#undef STAMP

  // final state back to h_state (streaming / next chunk)
#pragma unroll
  for (int i = 0; i < OWN_T; ++i)
    if (sidx[i] < a.n_clips) {
#pragma unroll
      for (int e = 0; e < OWN_R; ++e) a.h_state[(size_t)sidx[i] * HID + ucol[i] + e] = hreg[i][e];
    }
}

// Returns 0 on success, -1 for unsupported (hid, nct).
// a.hx must hold [2][G][64][hid] elements; both buffers are re-armed here (stream ordered):
// buffer 0 := tag 1 everywhere (first expected tag there is 0), buffer 1 := tag 0 (first expected tag is 1).
int launch_gru_recurrence(bool bf16, int hid, int nct, GruArgs a, hipStream_t s) {
  if (hid != 1024) return -1;
  const int P = bf16 ? 32 : 64;
  const size_t es = bf16 ? 2 : 4;
  const size_t buf_bytes = (size_t)a.G * 64 * hid * es;
  if (bf16) (void)hipMemsetD16Async((hipDeviceptr_t)a.hx, 0x4000, buf_bytes / 2, s);
  else (void)hipMemsetD32Async((hipDeviceptr_t)a.hx, 0x40000000, buf_bytes / 4, s);
  (void)hipMemsetAsync((char*)a.hx + buf_bytes, 0, buf_bytes, s);
  if (a.sync) (void)hipMemsetAsync(a.sync, 0, 16 * sizeof(unsigned), s);
  const int grid = a.G * P;
#define LAUNCH(WT, UT, NCT)                                                                            \
  do {                                                                                                 \
    const size_t lds = (size_t)((UT * NCT) <= 4 ? 2 : 1) * 4 * 3 * (UT * NCT) * 64 * 16;               \
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
