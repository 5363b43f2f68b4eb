// Persistent GRU recurrence for MiniROAD (nn.GRU(2048,1024,1,batch_first) at model/rnn/rnn.py:38,61):
//   gh = h W_hh^T + b_hh;  r = s(gi_r+gh_r);  z = s(gi_z+gh_z);  n = tanh(gi_n + r*gh_n);  h' = (1-z) n + z h
// The input projection gi (with b_ih, and b_hh for the r/z rows, already added) comes from the GEMM.
//
// MI355X design.  T sequential steps, each a [clips x H] x [H x 3H] product, is latency bound, so:
//  * W_hh never leaves the register file.  The 256 CUs are split into G independent groups of P
//    workgroups (bf16: P = 32, G = 8; fp32: P = 64, G = 4).  A workgroup owns 16*UT hidden units
//    (all three gates) and holds its 3*16*UT x H weight slice as MFMA A-fragments in VGPRs: wave q
//    (one wave per SIMD, 512-VGPR budget) keeps the K-quarter [q*H/4, (q+1)*H/4) = 192 VGPRs.
//  * Clips are independent (h0 = 0 per clip, rnn.py:49,60): the sorted slots are dealt to the groups 16 at a
//    time (layer ct = slots [ct*16G, (ct+1)*16G), group g takes [g*16, g*16+16) of each layer), so every group
//    advances its own <= 16*NCT slots and groups never talk.  Slots are sorted by load, so the groups finish one
//    after the other (the BASELINE workload: group 7 after 12.6 k of the 34 k steps, group 1 after 21.4 k) and a
//    finished group's workgroups EXIT, which hands their CUs to whatever else is queued on the device.
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
//  * A group's clips are processed in tiles of 16 (the MFMA N dimension), ONE TILE AT A TIME inside a step:
//    gather(tile) -> MFMA -> LDS reduce -> gates -> publish(tile).  Tile c's inputs were published during the
//    previous step, a whole (NCT-1)-tile pipeline ago, so with two or more live tiles the cross-CU latency of one
//    tile hides under the arithmetic of the others (measured 4.4 -> ~3 us per step for two tiles).
//  * gi for step t+1 is prefetched during step t; clip tiles whose clips have all ended are skipped.
//  * every spin is bounded; a timeout raises an abort word that ends the launch (no hung GPU).
#include "common.h"
#include "kernels.h"
#include "pass_handshake.h"
#include <cstdlib>

#define SPIN_LIMIT (1u << 22)
#ifndef GRU_NSEG
#define GRU_NSEG 2                           // segments of the gather validated and multiplied one after the other (A/B at the end of round 2, per launch: 1: 1.434, 2: 1.413, 4: 1.434, 8: 1.588 ms)
#endif
#define GRU_MAX_TILES 8                      // clip tiles per group the exchange buffer is laid out for (128 slots)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// operand tag of the weight type: the 16-bit types name themselves, fp32 weights never reach the 16-bit conversions
template <typename WT> struct gru_ot { typedef bf16_t type; };
template <> struct gru_ot<f16_t> { typedef f16_t type; };

// TRAIN: also store the gate activations / raw state BPTT needs (a.keep_*, a.h_raw_out); the inference instantiation
// carries none of that code or its registers.
// GI16: the input projection rows (a.gi) are bf16 (inference path with bf16 intermediates)
// PASS: the recurrence of a WHOLE pass in one launch on XCDs 0 .. a.Gd - 1 (split pass, ff_pass.hip): gi is a ring indexed by absolute
// packed row that the feed-forward kernel fills WHILE this launch runs (sc1 loads: the producer is on another XCD), a.gi_cnt[c] says
// when chunk c is complete, a.rec_cnt[c] tells the producer when its ring slot is free again.  Same step arithmetic.
template <typename WT, int HID, int UT, int NCT, bool TRAIN, bool GI16 = false, bool PASS = false>
__global__ __launch_bounds__(256, 1) void gru_recurrence_kernel(GruArgs a) {
  static_assert(!PASS || (GI16 && !TRAIN && NCT == 1 && sizeof(WT) == 2), "pass mode: 16-bit operands and GI, one clip tile, inference");
  constexpr bool BF = (sizeof(WT) == 2);          // 16-bit operands: bf16_t or f16_t (same layout, same tag bit 14: |h| < 2 in both)
  typedef typename gru_ot<WT>::type OT;
  constexpr int UNITS = 16 * UT;              // hidden units owned by this workgroup
  constexpr int KQ = HID / 4;                 // K range per wave
  constexpr int OWN_R = UT == 2 ? 2 : 1;      // accumulator registers a lane owns per clip tile (4 waves share UT tiles)
  constexpr int NKS = BF ? KQ / 32 : KQ / 16; // fragments per clip tile (bf16: 32 k each; f32: 16 k each)
  constexpr unsigned TAGM = BF ? 0x40004000u : 0x40000000u;
  // Exchange layout = MFMA B-fragment order: [k-step of 32 (bf16) / 16 (f32)][clip tile 0..7][lane 0..63][16 B], so a
  // consumer's fragment load is ONE contiguous 1 KiB (rows at a 2 KiB stride all fell on the same L2 channel and
  // ran at 10 B/clk/CU).  Element (clip slot c, hidden unit k): fragment (k / KF, c / 16), lane ((k % KF) / EPL) * 16
  // + c % 16, byte (k % EPL) * sizeof(WT), with KF = k per fragment, EPL = elements per lane.
  constexpr int KF = BF ? 32 : 16;
  constexpr int EPL = BF ? 8 : 4;
  constexpr int GROUP_BYTES = (HID / KF) * GRU_MAX_TILES * 1024;     // one buffer of one group

  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = (f32x4*)smem;                  // [2 parities][4 waves][3 gates][UT][64 lanes]
  constexpr int RED_STRIDE = 4 * 3 * UT * 64;

  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef GRU_PRIO
  // the recurrence is the latency-critical path of a pass: its waves win every issue arbitration against co-resident
  // feed-forward waves (one extra wave per SIMD fits beside the 312 registers of this kernel)
  __builtin_amdgcn_s_setprio(GRU_PRIO);
#endif
  // ---- placement rendezvous: group := XCD when every XCD holds exactly P workgroups ------------
  constexpr int P = HID / UNITS;
  __shared__ int s_place[4];
  if (tid == 0) {
    int gg = blockIdx.x % a.G, ww = blockIdx.x / a.G, loc = 0;
    // word 20 of the rendezvous array survives the per-launch re-arm: 1 = an earlier full-width launch of this handle verified that
    // every XCD receives exactly P workgroups (group := XCD).  Only then may a launch be compacted; otherwise it runs full width
    const bool compact_ok = a.sync != nullptr && a.G == 8 && gridDim.x == 8 * P && a.Gd > 0 && a.Gd < a.G &&
                            __hip_atomic_load(a.sync + 20, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u;
    s_place[3] = compact_ok ? 1 : 0;
    if constexpr (PASS) {
      // pass launch: shares the device with the feed-forward kernel (never full width), and starts with the handshake of kernels.h:
      // no workgroup proceeds before the leader (XCD 0's first ticket) has seen exactly-P-workgroups on each of the Gd recurrence XCDs
      // and a feed-forward workgroup on each of the others; every wait is bounded and a FAIL ends both launches before they wrote anything
      if (!compact_ok) { hs_decide(a.hs, PREGO_HS_FAIL); loc = -1; }
      else {
        const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
        if (xcc >= a.Gd) loc = -2;
        else {
          const unsigned ticket = __hip_atomic_fetch_add(a.sync + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ticket >= (unsigned)P) hs_decide(a.hs, PREGO_HS_FAIL);            // more than P workgroups on one XCD: not the verified placement
          else if (xcc == 0 && ticket == 0u) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
              bool all = true;
              for (int x = 0; x < 8; ++x) {
                const unsigned v = __hip_atomic_load(x < a.Gd ? a.sync + x : a.hs.ff_here + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                all = all && v >= (x < a.Gd ? (unsigned)P : 1u);
              }
              if (all) { hs_decide(a.hs, PREGO_HS_GO); break; }
              if (__hip_atomic_load(a.hs.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
              if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)a.hs.ticks_lead) { hs_decide(a.hs, PREGO_HS_FAIL); break; }
              __builtin_amdgcn_s_sleep(16);
            }
          }
          if (hs_wait(a.hs, a.hs.ticks_all) == PREGO_HS_GO && ticket < (unsigned)P) { loc = 1; gg = xcc; ww = (int)ticket; }
          else loc = -1;
        }
      }
    } else
    if (compact_ok) {
      // compacted launch (a.Gd groups; probe of DESIGN 5c): the rendezvous is per XCD - a workgroup on an XCD without a group leaves
      // at once and is not waited for (another kernel may hold those CUs for the whole launch); the others wait for the 32 of
      // their own XCD only.  An XCD below Gd that does not collect exactly P workgroups ends the launch with the abort word.
      const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
      if (xcc >= a.Gd) loc = -2;
      else {
        const unsigned ticket = __hip_atomic_fetch_add(a.sync + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        bool ok = true;
        while (__hip_atomic_load(a.sync + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)P) {
          if (++spins > SPIN_LIMIT) { ok = false; break; }
          __builtin_amdgcn_s_sleep(4);
        }
        if (!ok || ticket >= (unsigned)P) { __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); loc = -1; }
        else { loc = 1; gg = xcc; ww = (int)ticket; }
      }
    } else if (a.sync != nullptr && a.G == 8 && gridDim.x == 8 * P) {
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
        if (blockIdx.x == 0) __hip_atomic_store(a.sync + 20, loc ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    s_place[0] = gg; s_place[1] = ww; s_place[2] = loc;
  }
  __syncthreads();
  const int g = __builtin_amdgcn_readfirstlane(s_place[0]);   // group
  const int w = __builtin_amdgcn_readfirstlane(s_place[1]);   // member of the group
  const int place = __builtin_amdgcn_readfirstlane(s_place[2]);
  const bool compacted = __builtin_amdgcn_readfirstlane(s_place[3]) != 0;
  if (place < 0) return;                      // rendezvous timed out (abort word set)
  const bool local = place == 1;              // whole group on one XCD, verified
  // Slots are dealt to the groups INTERLEAVED: slot s of a layer of 16 G slots goes to group s % G, column s / G of its tile.
  // The slots are sorted by load, so every group gets a cross-section of the loads and its live columns thin out over the
  // whole run (instead of group 0 holding the 16 longest slots to the end): the per-step gather below skips dead columns, so
  // the 1 MB-per-step L2 burst of an XCD - what bounds the hand-off - shrinks with the number of live clips.
  // a.Gd (0 = a.G): groups the slots are dealt to in this launch.  Fewer than a.G packs the live slots into the first Gd groups
  // (XCDs 0 .. Gd - 1 under the verified placement): the other XCDs' workgroups leave at once (overlap probe)
  const int gd = compacted ? a.Gd : a.G;
  if (g >= gd) return;
  const int l15 = lane & 15, l4 = lane >> 4;
  // pass mode: chunks this wave knows complete (`have`) / has reported consumed (`sig`); both wave-uniform
  int have = 0, sig = 0;
  auto pass_signal_upto = [&](int c_end) {     // caller: every vector-memory op that touched chunks < c_end has retired
    if constexpr (PASS) {
      while (sig < c_end) {
        if (lane == 0) __hip_atomic_fetch_add(a.rec_cnt + sig, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++sig;
      }
    }
  };
  if (g >= a.n_clips) { pass_signal_upto(a.n_chunks); return; }                 // group without slots (its most loaded slot is slot g)
  {                                           // group whose slots have all ended before this launch (nact never grows)
    typedef const __attribute__((address_space(4))) int* cint_p0;
    if (g >= ((cint_p0)a.nact)[a.t0]) { pass_signal_upto(a.n_chunks); return; }
  }
  // every chunk up to the one that holds packed row `last_row` is complete (false: aborted)
  auto pass_wait_row = [&](int last_row) -> bool {
    if constexpr (PASS) {
      const int c_need = last_row >> a.chunk_shift;
      while (have <= c_need) {
        const unsigned need = (unsigned)(have == a.n_chunks - 1 ? a.units_last : a.units_per_chunk);
        unsigned spins = 0;
        while (__hip_atomic_load(a.gi_cnt + have, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
          if (++spins > (1u << 21)) {
            if (lane == 0) __hip_atomic_store(a.abort_word, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
          }
          if ((spins & 63u) == 0u && __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
          __builtin_amdgcn_s_sleep(8);
        }
        ++have;
      }
    }
    return true;
  };

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

  // ---- gate-phase ownership inside a clip tile: the UT accumulator tiles x 4 registers are split over the 4 waves
  const int own_ut = UT == 2 ? (q & 1) : 0;
  // (round 6: letting the owned register pair alternate with bit 3 of the lane makes the gate phase's 8-byte LDS reads conflict-free -
  // and changes nothing: 94.6 against 94.6 ms, profiles/r06_ab_ownswz.log.  The counted bank conflicts are not on the step's critical path)
  const int own_r0 = UT == 2 ? (q >> 1) * 2 : q;
  const int ucol = w * UNITS + own_ut * 16 + l4 * 4 + own_r0;       // first hidden unit of this lane's registers
  float hreg[NCT][OWN_R];
  float bhn[OWN_R];
  int sidx[NCT];      // sorted clip index of my clip in tile ct (may be >= n_clips)
  int tfirst[NCT];    // sorted index of the tile's first slot: tile in use at t iff tfirst < nact[t]
#pragma unroll
  for (int e = 0; e < OWN_R; ++e) bhn[e] = a.b_hn[ucol + e];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    sidx[ct] = ct * 16 * gd + l15 * gd + g;
    tfirst[ct] = ct * 16 * gd + g;
#pragma unroll
    for (int e = 0; e < OWN_R; ++e) hreg[ct][e] = (sidx[ct] < a.n_clips) ? a.h_state[(size_t)sidx[ct] * HID + ucol + e] : 0.f;
  }
  // continuous batching: a slot runs several clips back to back; h restarts from 0 where the next clip begins
  int nstart[NCT], segp[NCT], sege[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    nstart[ct] = 0x7fffffff; segp[ct] = 0; sege[ct] = 0;
    if (a.seg_start != nullptr && sidx[ct] < a.n_clips) {
      int k = a.seg_off[sidx[ct]];
      sege[ct] = a.seg_off[sidx[ct] + 1];
      while (k < sege[ct] && a.seg_start[k] < a.t0) ++k;
      if (k < sege[ct] && a.seg_start[k] == a.t0 && a.t0 > 0) {        // a clip starts on this launch's first step
#pragma unroll
        for (int e = 0; e < OWN_R; ++e) hreg[ct][e] = 0.f;
      }
      while (k < sege[ct] && a.seg_start[k] <= a.t0) ++k;
      segp[ct] = k;
      nstart[ct] = k < sege[ct] ? a.seg_start[k] : 0x7fffffff;
    }
  }

  // exchange buffers: [2][G][GROUP_BYTES]; one descriptor per group, buffer index in the offset
  const int buf_stride = a.G * GROUP_BYTES;
  char* hx_base = (char*)a.hx + (size_t)g * GROUP_BYTES;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)hx_base, 0, buf_stride + GROUP_BYTES, 0x00020000);

  // tagged MFMA-operand copy of one state element
  auto tag_bf = [](float x, unsigned tag) -> unsigned {
    x = fminf(fmaxf(x, -1.9921875f), 1.9921875f);
    return ((unsigned)f2bf(x) & 0xBFFFu) | (tag << 14);
  };
  auto tag_f32 = [](float x, unsigned tag) -> unsigned {
    x = fminf(fmaxf(x, -1.9999998f), 1.9999998f);
    return (__float_as_uint(x) & 0xBFFFFFFFu) | (tag << 30);
  };
  // publish this lane's slice of tile ct; no waiting
  auto publish = [&](int ct, int buf, unsigned tag, bool zero) {
    const int off = buf * buf_stride + ((ucol / KF) * GRU_MAX_TILES + ct) * 1024 + ((((ucol % KF) / EPL) << 4) + l15) * 16 +
                    (ucol % EPL) * (int)sizeof(WT);
    if constexpr (BF && OWN_R == 2) {
      f32x2 hv = {zero ? 0.f : hreg[ct][0], zero ? 0.f : hreg[ct][1]};
      hv[0] = __builtin_amdgcn_fmed3f(hv[0], -1.9921875f, 1.9921875f);
      hv[1] = __builtin_amdgcn_fmed3f(hv[1], -1.9921875f, 1.9921875f);
      const unsigned v = (op16<OT>::pack2(hv[0], hv[1]) & 0xBFFFBFFFu) | (tag ? 0x40004000u : 0u);
      if (local) __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, AUX_SC1);
    } else if constexpr (BF) {                 // one 16-bit element per lane (UT = 1: hidden sizes whose weight slice leaves room for one row tile)
      const float hv = __builtin_amdgcn_fmed3f(zero ? 0.f : hreg[ct][0], -1.9921875f, 1.9921875f);
      const unsigned short v = (unsigned short)(((unsigned)op16<OT>::cvt(hv) & 0xBFFFu) | (tag ? 0x4000u : 0u));
      if (local) __builtin_amdgcn_raw_buffer_store_b16(v, rs, off, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b16(v, rs, off, 0, AUX_SC1);
    } else {
      const unsigned v = tag_f32(zero ? 0.f : hreg[ct][0], tag);
      if (local) __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 0);
      else __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, AUX_SC1);
    }
  };
  // gi loads are UNCONDITIONAL (inactive lanes read row 0 of the step, which always exists, and ignore it): a
  // "load or zero" select makes hipcc branch around every load and wait vmcnt(0) right behind it.
  __amdgpu_buffer_rsrc_t rs_gi = rs;
  if constexpr (PASS) rs_gi = __builtin_amdgcn_make_buffer_rsrc((void*)a.gi, 0, (a.gi_row_mask + 1u) * (unsigned)(3 * HID * 2), 0x00020000);
  auto load_gi = [&](float (&dst)[3][OWN_R], int ct, int na, int rbase) {
    const int r = rbase + (sidx[ct] < na ? sidx[ct] : 0);
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) {
      if constexpr (PASS) {                    // (handled above: one 12-byte load for the three gates)
      } else
      if constexpr (GI16) {
        const bf16_t* p = (const bf16_t*)a.gi + (size_t)r * (3 * HID) + gate * HID + ucol;
        // RAW bits only: unpacking here would make hipcc wait for the prefetch right behind its issue; the gate phase unpacks
        if constexpr (OWN_R == 2) dst[gate][0] = __uint_as_float(*(const unsigned*)p);
        else dst[gate][0] = __uint_as_float((unsigned)p[0]);
      } else {
        const float* p = (const float*)a.gi + (size_t)r * (3 * HID) + gate * HID + ucol;
        if constexpr (OWN_R == 2) {
          const float2 v = *(const float2*)p;
          dst[gate][0] = v.x; dst[gate][1] = v.y;
        } else {
          dst[gate][0] = p[0];
        }
      }
    }
  };

  // pass mode: ring slot of the absolute row; the row was written on another XCD: sc1.  The feed-forward launch of a split pass projects
  // with W_ih's rows permuted to (unit pair, gate, unit % 2) order (miniroad.cpp: w_ih_perm), so this lane's r / z / n pairs are 12 adjacent
  // bytes: ONE vector-memory instruction per step instead of three (round 6; they queue in front of the next step's gather).  The three
  // dwords stay ONE register triple from the load to the gate phase: copied into separate registers they cost an s_waitcnt vmcnt(0) right
  // behind the load (measured: +0.35 us per step)
  static_assert(!PASS || OWN_R == 2, "pass mode: a lane owns a pair of units");
  auto load_gi_pass = [&](int ct, int na, int rbase) -> u32x3 {
    const int r = rbase + (sidx[ct] < na ? sidx[ct] : 0);
    const unsigned off = ((unsigned)r & a.gi_row_mask) * (unsigned)(3 * HID * 2) + (unsigned)((ucol >> 1) * 12);
    return __builtin_amdgcn_raw_buffer_load_b96(rs_gi, (int)off, 0, AUX_SC1);
  };
  // (round 6, measured: compiling the stamps OUT of the pass instantiation - six wave-uniform branches per step less - makes the step 6 %
  // SLOWER, 2.26 against 2.13 us on one box, and so does folding `local` into a constant there: the branches cut the step into the basic
  // blocks its phases are, and hipcc's scheduler does worse with the freedom of one big block.  profiles/r06_ab_nostamp.log)
  const bool stamp = a.stamps != nullptr && (PASS ? (g == 0 && w == 0) : blockIdx.x == 0) && q == 0;   // pass mode: workgroup 0 may sit on an XCD without a group
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long st_t = 0;
#define STAMP(i) do { if (stamp) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_t; st_t = n_; } } while (0)
  // the plan tables are read-only for the whole launch: constant address space = scalar (s_load) path, so the
  // look-ahead never touches the vector-memory queue (a vector load here drags a vmcnt wait through every step)
  typedef const __attribute__((address_space(4))) int* cint_p;
  cint_p nact_c = (cint_p)a.nact;
  cint_p rowoff_c = (cint_p)a.rowoff;
  const int nsteps = a.t1 - a.t0;
  // plan scalars one step ahead of their use (s_load latency off the critical path)
  // (rowoff is carried RAW and row_base subtracted where it is used: a subtraction next to the look-ahead s_load makes
  // hipcc wait for that load at the top of every step)
  int na_c = nact_c[a.t0], rb_c = rowoff_c[a.t0];                      // step tl
  int na_n = nsteps > 1 ? nact_c[a.t0 + 1] : 0, rb_n = nsteps > 1 ? rowoff_c[a.t0 + 1] : 0;   // step tl+1
  // prologue: h_{t0-1} -> buffer 1 with the tag of step "-1" (= 1); gi of the first step
  float giA[NCT][3][OWN_R], giB[NCT][3][OWN_R];                         // ping-pong: no register copies
  u32x3 gvA[NCT], gvB[NCT];                                             // pass mode: the same as one register triple per tile
  if constexpr (PASS) { if (!pass_wait_row(rowoff_c[a.t0 + 1] - 1)) return; }      // the first step's rows
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
    if (tfirst[ct] < na_c) {
      publish(ct, 1, 1u, false);
      if constexpr (PASS) gvA[ct] = load_gi_pass(ct, na_c, rb_c - a.row_base);
      else load_gi(giA[ct], ct, na_c, rb_c - a.row_base);
    }
  int parity = 0;

  // one time step; gir = gi of this step (loaded a step ago), gin = where the next step's gi lands
  auto step = [&](const int tl, float (&gir)[NCT][3][OWN_R], float (&gin)[NCT][3][OWN_R], u32x3 (&gvr)[NCT], u32x3 (&gvn)[NCT]) -> bool {
    const int t = a.t0 + tl;
    const int na = na_c;
    const int rbase = rb_c - a.row_base;
    const bool more = tl + 1 < nsteps;
    const int t2 = (tl + 2 < nsteps) ? t + 2 : t;                       // look-ahead index (clamped)
    const int na_2 = nact_c[t2], rb_2 = rowoff_c[t2];
    const int re_n = PASS ? rowoff_c[more ? t + 2 : t + 1] : 0;         // pass mode: end of step t + 1's rows
    const int rbuf = (tl + 1) & 1;
    const unsigned etag = (unsigned)(((tl - 1) >> 1) & 1);              // tag of the step that produced h_{t-1}
    const unsigned eword = etag ? TAGM : 0u;
    const unsigned untag = etag ? ~TAGM : 0xFFFFFFFFu;
    if (stamp) st_t = __builtin_amdgcn_s_memtime();

#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      if (tfirst[ct] < na) {                                            // else: this and all later clip tiles are finished
        // ---- (1) gather h_{t-1} of tile ct (data-is-the-flag): load every fragment, re-load the ones that still show
        // the old tag until all are valid; then multiply in a FIXED order (bit-reproducible fp32 sums).
        // Every round (re)loads ALL fragments and validates them with one AND- or OR-reduction over the tag bits: no
        // per-fragment bookkeeping.  (Tracking stale fragments individually made hipcc carry the fragment registers
        // through the spin loop in AGPRs - 165 v_accvgpr moves and ~100 scalar branches per step - for a retry that
        // happens 0.3 times per step and costs one L2-served 8 KB re-read.)
        // The fragments are consumed in NSEG segments: segment s is validated and multiplied while the later segments are
        // still on their way (loads retire in order: vmcnt counts down), so most of the gather's transfer - 32 KB per CU
        // and step at 64 B/clk - runs under the MFMAs.  A stale segment (its producers are late) re-loads itself AND
        // every later segment until it is valid; the summation order stays fixed (ks ascending) whatever happens.
        constexpr int NSEG = GRU_NSEG, SEGK = NKS / NSEG;
        f32x4 acc[3][UT];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        u32x4 hb[NKS];
        unsigned spins = 0;
        // a lane's 16 bytes of a fragment belong to ONE clip column (lane & 15): columns whose slot has ended are not fetched
        // (an offset past the buffer's num_records returns zeros without a memory access) and never count as stale
        const bool col_live = sidx[ct] < na;
        const int lane_off = col_live ? lane * 16 : 0x7FFF0000;
        if (local) {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks)
            hb[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, rbuf * buf_stride + ((q * NKS + ks) * GRU_MAX_TILES + ct) * 1024 + lane_off, 0, AUX_NT);
        } else {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks)
            hb[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, rbuf * buf_stride + ((q * NKS + ks) * GRU_MAX_TILES + ct) * 1024 + lane_off, 0, AUX_SC1);
        }
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
          auto seg_stale = [&]() -> bool {
            unsigned badv = 0u;
#pragma unroll
            for (int ks = sg * SEGK; ks < (sg + 1) * SEGK; ++ks)
            {
              // the xor that tests the tag also REMOVES it (a fresh element carries exactly `eword`'s bits there): the fragment is
              // un-tagged in place and goes to the MFMA as it is - no separate mask pass (32 VALU per step until round 5)
              hb[ks][0] ^= eword; hb[ks][1] ^= eword; hb[ks][2] ^= eword; hb[ks][3] ^= eword;      // (skipping it on tag-0 steps behind a branch: 1 % slower)
              badv |= (hb[ks][0] | hb[ks][1]) | (hb[ks][2] | hb[ks][3]);            // tag bit survives iff stale
            }
            return !__all(!col_live || (badv & TAGM) == 0u);
          };
          if (seg_stale()) {
            for (;;) {
              if (++spins > SPIN_LIMIT) {
                if (lane == 0) __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
              }
              if ((spins & 255u) == 0u) {
                if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
              }
              if (local && spins < 6u) {
#pragma unroll
                for (int ks = sg * SEGK; ks < NKS; ++ks)
                  hb[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, rbuf * buf_stride + ((q * NKS + ks) * GRU_MAX_TILES + ct) * 1024 + lane_off, 0, AUX_NT);
              } else {
#pragma unroll
                for (int ks = sg * SEGK; ks < NKS; ++ks)
                  hb[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, rbuf * buf_stride + ((q * NKS + ks) * GRU_MAX_TILES + ct) * 1024 + lane_off, 0, AUX_SC1);
              }
              if (!seg_stale()) break;
            }
          }
          if (sg == 0) STAMP(1);                   // stamps only: top of step -> first segment valid (the hand-off latency)
#pragma unroll
          for (int ks = sg * SEGK; ks < (sg + 1) * SEGK; ++ks) {
            const u32x4 v = hb[ks];               // un-tagged by the staleness test above (dead columns: zeros ^ eword, products nobody reads)
            if constexpr (BF) {
              const bf16x8 bfrag = __builtin_bit_cast(bf16x8, v);
#pragma unroll
              for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                for (int ut = 0; ut < UT; ++ut)
                  acc[gate][ut] = op16<OT>::mfma(wb[gate][ut][ks], bfrag, ks == 0 ? zero4 : acc[gate][ut]);
            } else {
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                const float bj = __uint_as_float(v[jj]);
#pragma unroll
                for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                  for (int ut = 0; ut < UT; ++ut)
                    acc[gate][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[gate][ut][ks][jj], bj, (ks == 0 && jj == 0) ? zero4 : acc[gate][ut], 0, 0, 0);
              }
            }
          }
        }
        if (stamp) st_acc[5] += spins;
        STAMP(0);
        // (issuing the next tile's gather here, under this tile's reduction and gate phase, measured SLOWER: hipcc
        // answers the loop-carried loads with vmcnt(0) waits that drag the prefetch's latency into the gate phase;
        // it needs asm-issued loads with hand-counted waits - next round)
        // ---- (1b) the gather's vmcnt(0) has just retired every older vector-memory op, including the loads of this
        // step's gi (issued one step ago).  Pin that fact for the compiler (it would otherwise put a vmcnt(0) in front
        // of the first use of the loop-carried registers, i.e. behind the prefetch issued next), then prefetch gi(t+1)
        if constexpr (PASS) asm volatile("" : "+v"(gvr[ct]));
        else {
#pragma unroll
          for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int e = 0; e < OWN_R; ++e) asm volatile("" : "+v"(gir[ct][gate][e]));
        }
        if constexpr (PASS) {
          // rows below rowoff[t] belong to finished steps and nothing of them is in flight (the vmcnt(0) above): report their chunks;
          // then make sure the chunk(s) of step t + 1's rows are complete before the prefetch below reads them
          const int c_done = rb_c >> a.chunk_shift;
          if (sig < c_done) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pass_signal_upto(c_done); }
          if (more) { if (!pass_wait_row(re_n - 1)) return false; }
        }
        if (more && tfirst[ct] < na_n) {
          if constexpr (PASS) gvn[ct] = load_gi_pass(ct, na_n, rb_n - a.row_base);
          else load_gi(gin[ct], ct, na_n, rb_n - a.row_base);
        }

        // ---- (2) cross-wave (K-quarter) reduction through LDS, double buffered: one barrier per tile
        // (round 5: rocprofv3 counts SQ_LDS_BANK_CONFLICT = 1/3 of this kernel's active LDS cycles - the gate phase reads 8 bytes per lane
        // at a 16-byte stride.  Writing every accumulator as two 8-byte planes makes those reads conflict-free and was measured SLOWER:
        // 2.014 against 1.984 us per step in the split pass, equal in the chunked pass, same device, scripts/probes/ab_lib.sh - twelve
        // ds_write_b64 instead of six b128 in front of the barrier cost more than the conflicts behind it)
        f32x4* redw = red + parity * RED_STRIDE;
        parity ^= 1;
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
          for (int ut = 0; ut < UT; ++ut) redw[((q * 3 + gate) * UT + ut) * 64 + lane] = acc[gate][ut];
        __syncthreads();
        STAMP(2);

        // ---- (3) gates + state update for the registers this lane owns
        {
          if constexpr (OWN_R == 2) {
            // this lane owns registers {own_r0, own_r0+1} of its tile: 8-byte LDS reads at immediate offsets, and the two
            // elements go through the gate math as one f32x2 (v_pk_add/mul/fma_f32): the step runs one wave per SIMD,
            // so every VALU instruction is ~4 cycles of its critical path
            const f32x2* rp = (const f32x2*)((const float*)&redw[own_ut * 64 + lane] + own_r0);
            f32x2 part[3][4];
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
#pragma unroll
              for (int qq = 0; qq < 4; ++qq) part[gate][qq] = rp[((qq * 3 + gate) * UT) * 64 * 2];
            // all twelve reads in flight before the first add (left alone, hipcc recycles registers and serialises the
            // LDS round trips: read 2, wait, add, read 2, wait, ...)
            asm volatile("" : "+v"(part[0][0]), "+v"(part[0][1]), "+v"(part[0][2]), "+v"(part[0][3]), "+v"(part[1][0]), "+v"(part[1][1]),
                              "+v"(part[1][2]), "+v"(part[1][3]), "+v"(part[2][0]), "+v"(part[2][1]), "+v"(part[2][2]), "+v"(part[2][3]));
            f32x2 gh[3];
#pragma unroll
            for (int gate = 0; gate < 3; ++gate) gh[gate] = (part[gate][0] + part[gate][1]) + (part[gate][2] + part[gate][3]);
            if (sidx[ct] < na) {
              const f32x2 one = {1.f, 1.f};
              f32x2 gr, gz, gn;
              if constexpr (GI16) {          // one dword = the bf16 pair of my two units
                const unsigned ur = PASS ? gvr[ct][0] : __float_as_uint(gir[ct][0][0]), uz = PASS ? gvr[ct][1] : __float_as_uint(gir[ct][1][0]),
                               un = PASS ? gvr[ct][2] : __float_as_uint(gir[ct][2][0]);
                gr = (f32x2){op16<OT>::lo(ur), op16<OT>::hi(ur)};
                gz = (f32x2){op16<OT>::lo(uz), op16<OT>::hi(uz)};
                gn = (f32x2){op16<OT>::lo(un), op16<OT>::hi(un)};
              } else {
                gr = (f32x2){gir[ct][0][0], gir[ct][0][1]}; gz = (f32x2){gir[ct][1][0], gir[ct][1][1]}; gn = (f32x2){gir[ct][2][0], gir[ct][2][1]};
              }
              const f32x2 hp = {hreg[ct][0], hreg[ct][1]}, bh = {bhn[0], bhn[1]};
              const f32x2 xr = gr + gh[0], xz = gz + gh[1];
              f32x2 r, z, n;
              r[0] = sigmoidf_(xr[0]); r[1] = sigmoidf_(xr[1]);
              z[0] = sigmoidf_(xz[0]); z[1] = sigmoidf_(xz[1]);
              const f32x2 ghn = gh[2] + bh;
              const f32x2 xn = gn + r * ghn;
              n[0] = tanhf_(xn[0]); n[1] = tanhf_(xn[1]);
              const f32x2 hn = (one - z) * n + z * hp;
              hreg[ct][0] = hn[0]; hreg[ct][1] = hn[1];
              if constexpr (TRAIN) {
                if (a.keep_r) {                                      // training: gate activations for BPTT
                  const size_t ko = (size_t)(rbase + sidx[ct]) * HID + ucol;
                  *(f32x2*)(a.keep_r + ko) = r; *(f32x2*)(a.keep_z + ko) = z;
                  *(f32x2*)(a.keep_n + ko) = n; *(f32x2*)(a.keep_ghn + ko) = ghn;
                }
              }
            }
          } else {
            float gh[3][OWN_R];
            float part[3][4];
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
#pragma unroll
              for (int qq = 0; qq < 4; ++qq)
                part[gate][qq] = *((const float*)&redw[((qq * 3 + gate) * UT + own_ut) * 64 + lane] + own_r0);
#pragma unroll
            for (int gate = 0; gate < 3; ++gate) gh[gate][0] = (part[gate][0] + part[gate][1]) + (part[gate][2] + part[gate][3]);
            if (sidx[ct] < na) {
#pragma unroll
              for (int e = 0; e < OWN_R; ++e) {
                const float g_r = GI16 ? op16<OT>::lo(__float_as_uint(gir[ct][0][e])) : gir[ct][0][e];
                const float g_z = GI16 ? op16<OT>::lo(__float_as_uint(gir[ct][1][e])) : gir[ct][1][e];
                const float g_n = GI16 ? op16<OT>::lo(__float_as_uint(gir[ct][2][e])) : gir[ct][2][e];
                const float r = sigmoidf_(g_r + gh[0][e]);
                const float z = sigmoidf_(g_z + gh[1][e]);
                const float ghn = gh[2][e] + bhn[e];
                const float n = tanhf_(g_n + r * ghn);
                hreg[ct][e] = (1.0f - z) * n + z * hreg[ct][e];
                if constexpr (TRAIN) {
                  if (a.keep_r) {
                    const size_t ko = (size_t)(rbase + sidx[ct]) * HID + ucol + e;
                    a.keep_r[ko] = r; a.keep_z[ko] = z; a.keep_n[ko] = n; a.keep_ghn[ko] = ghn;
                  }
                }
              }
            }
          }
        }
        // ---- (4) publish h_t of this tile for step t+1 (fire and forget)
        const bool restart = (t + 1 == nstart[ct]);          // the slot's next clip begins at step t+1: it sees h = 0
        if (more && sidx[ct] < na_n) publish(ct, tl & 1, (unsigned)((tl >> 1) & 1), restart);   // dead columns are never read
        STAMP(3);
        // ---- (5) outputs
        if (sidx[ct] < na) {
          const size_t o = (size_t)(rbase + sidx[ct]) * HID + ucol;
          if (a.h_relu_out) {
            // a.out_floor: 0 = relu(h_t) (rnn.py:62, the classifier's operand); -inf = h_t itself (the next layer's input of a stacked GRU)
            if constexpr (BF) {
              bf16_t* p = (bf16_t*)a.h_relu_out + o;
              if constexpr (OWN_R == 2) *(unsigned*)p = op16<OT>::pack2(fmaxf(hreg[ct][0], a.out_floor), fmaxf(hreg[ct][1], a.out_floor));
              else p[0] = op16<OT>::cvt(fmaxf(hreg[ct][0], a.out_floor));
            } else {
              float* p = (float*)a.h_relu_out + o;
#pragma unroll
              for (int e = 0; e < OWN_R; ++e) p[e] = fmaxf(hreg[ct][e], a.out_floor);
            }
          }
          if constexpr (TRAIN) {
            if (a.h_raw_out) {
              if constexpr (OWN_R == 2) *(f32x2*)(a.h_raw_out + o) = (f32x2){hreg[ct][0], hreg[ct][1]};       // neighbours: one 8-byte store
              else a.h_raw_out[o] = hreg[ct][0];
            }
          }
        }
        if (restart) {
#pragma unroll
          for (int e = 0; e < OWN_R; ++e) hreg[ct][e] = 0.f;
          ++segp[ct];
          nstart[ct] = segp[ct] < sege[ct] ? a.seg_start[segp[ct]] : 0x7fffffff;
          asm volatile("" : "+v"(nstart[ct]));      // take the (rare) load's wait here, not as a vmcnt(0) in every step
        }
        STAMP(4);
      }
    }
    na_c = na_n; rb_c = rb_n; na_n = na_2; rb_n = rb_2;
    return true;
  };
  const unsigned long long rt0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0ull;       // 100 MHz: with the cycle sums below = this XCD's shader clock
  for (int tl = 0; tl < nsteps; tl += 2) {
    if (tfirst[0] >= na_c) break;             // every slot of this group has ended: leave, free the CU
    if (!step(tl, giA, giB, gvA, gvB)) return;
    if (tl + 1 < nsteps && !step(tl + 1, giB, giA, gvB, gvA)) return;
  }
  if (stamp && lane == 0) {
    for (int i = 0; i < 6; ++i) a.stamps[i] += st_acc[i];
    a.stamps[6] += (unsigned long long)nsteps;
    a.stamps[7] += __builtin_amdgcn_s_memrealtime() - rt0;
  }
#undef STAMP
  if constexpr (PASS) {                       // this group is done with every remaining chunk
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pass_signal_upto(a.n_chunks);
  }

  // final state back to h_state (streaming / next chunk)
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
    if (sidx[ct] < a.n_clips) {
#pragma unroll
      for (int e = 0; e < OWN_R; ++e) a.h_state[(size_t)sidx[ct] * HID + ucol + e] = hreg[ct][e];
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Multi-tile steps, software-pipelined (round 3).  With two or more live clip tiles per group the kernel above runs the
// tiles of a step strictly one after the other (measured 2.0 / 4.0 / 8.6 us per step for 1 / 2 / 4 tiles): every tile pays its
// own gather - ~1 000 cycles of L2 latency and transfer for 32 KB per CU, although its data was published a whole tile-slot
// earlier and is simply sitting in the XCD's L2.  Here the gather of the NEXT tile-slot (tile ct + 1 of this step, or tile 0
// of the next step after the last live tile) is issued right behind this tile's MFMAs as LDS-DMA (global_load_lds_dwordx4: L2 ->
// LDS, lane-linear 1 KiB per fragment, wave-private region, no destination registers the compiler could copy before the data has
// landed - a first version with asm loads into VGPRs was miscompiled exactly that way: hipcc parked the "loaded" registers in
// AGPRs right behind the issue), and consumed one slot later: a hand-counted `s_waitcnt vmcnt(5)`, then eight ds_read_b128 of the
// wave's own image.  Exactly five vector-memory instructions
// are issued between the two points on every path - the three gi prefetch loads, the publish store and the relu(h) store, all
// made unconditional (lanes / tiles that must not act use an offset past the buffer's num_records: no memory access, but the
// instruction is issued and counted; a wait count that is too SMALL only waits longer, one that is too large would read
// registers before their data has landed).  The compiler's own waits stay correct: it counts fewer outstanding operations
// than there are, i.e. it can only over-wait.  Same arithmetic, same summation order as the kernel above: bit-identical
// results (the chunking / continuous-batching / clip-alone tests run both).  bf16 / fp16 operands, inference only.
// A step with ONE live tile cannot prefetch (its h is not produced yet) and runs the classic sequence; launches whose live
// slots fit one tile per group use the kernel above (launch_gru_recurrence picks per launch).
// ------------------------------------------------------------------------------------------------------------------------
#define GRU_MT_DMA(SRC, LDSADDR, POLICY) \
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off " POLICY : : "v"(SRC), "s"(LDSADDR) : "memory", "m0")

template <typename WT, int HID, int NCT, bool SPEC>
__global__ __launch_bounds__(256, 1) void gru_recurrence_mt_kernel(GruArgs a) {
  static_assert(sizeof(WT) == 2 && NCT >= 2, "multi-tile kernel: 16-bit operands, two or more clip tiles");
  typedef typename gru_ot<WT>::type OT;
  constexpr int UT = 2, UNITS = 16 * UT, KQ = HID / 4, NKS = KQ / 32, KF = 32, EPL = 8;
  constexpr unsigned TAGM = 0x40004000u;
  constexpr int GROUP_BYTES = (HID / KF) * GRU_MAX_TILES * 1024;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = (f32x4*)smem;                  // [2 parities][4 waves][3 gates][UT][64 lanes]
  constexpr int RED_STRIDE = 4 * 3 * UT * 64;
  constexpr int RED_BYTES = 2 * RED_STRIDE * 16;
  char* gsm = smem + RED_BYTES;               // gather images [2 buffers][4 waves][NKS fragments][64 lanes][16 B]
  const unsigned gsm_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)gsm;
  constexpr int GATHER_BYTES = 2 * 4 * NKS * 1024;
  // gi rows (16-bit pairs of my two units) also arrive by LDS-DMA: [2 step parities][NCT][3 gates][256 threads] dwords.  With
  // no compiler-visible vector load left in the loop, hipcc inserts no vmcnt wait of its own (its conservative vmcnt(0) in
  // front of the loop-carried gi registers would wait for the gather prefetch too: measured in the first build of this kernel)
  unsigned* gism = (unsigned*)(gsm + GATHER_BYTES);
  const unsigned gism_lds = gsm_lds + (unsigned)GATHER_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- placement rendezvous (as in gru_recurrence_kernel)
  constexpr int P = HID / UNITS;
  __shared__ int s_place[4];
  if (tid == 0) {
    int gg = blockIdx.x % a.G, ww = blockIdx.x / a.G, loc = 0;
    if (a.sync != nullptr && a.G == 8 && gridDim.x == 8 * P) {
      const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
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
  const int g = __builtin_amdgcn_readfirstlane(s_place[0]);
  const int w = __builtin_amdgcn_readfirstlane(s_place[1]);
  const int place = __builtin_amdgcn_readfirstlane(s_place[2]);
  if (place < 0) return;
  const bool local = place == 1;
  if (g >= a.n_clips) return;
  {
    typedef const __attribute__((address_space(4))) int* cint_p0;
    if (g >= ((cint_p0)a.nact)[a.t0]) return;
  }
  const int l15 = lane & 15, l4 = lane >> 4;

  // ---- resident weights
  bf16x8 wb[3][UT][NKS];
#pragma unroll
  for (int gate = 0; gate < 3; ++gate)
#pragma unroll
    for (int ut = 0; ut < UT; ++ut)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const size_t row = (size_t)gate * HID + w * UNITS + ut * 16 + l15;
        wb[gate][ut][ks] = *(const bf16x8*)((const bf16_t*)a.whh + row * HID + q * KQ + ks * 32 + 8 * l4);
      }
  const int own_ut = q & 1, own_r0 = (q >> 1) * 2;
  const int ucol = w * UNITS + own_ut * 16 + l4 * 4 + own_r0;
  float hreg[NCT][2], bhn[2];
  int sidx[NCT], tfirst[NCT];
  bhn[0] = a.b_hn[ucol]; bhn[1] = a.b_hn[ucol + 1];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    sidx[ct] = ct * 16 * a.G + l15 * a.G + g;
    tfirst[ct] = ct * 16 * a.G + g;
#pragma unroll
    for (int e = 0; e < 2; ++e) hreg[ct][e] = (sidx[ct] < a.n_clips) ? a.h_state[(size_t)sidx[ct] * HID + ucol + e] : 0.f;
  }
  int nstart[NCT], segp[NCT], sege[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    nstart[ct] = 0x7fffffff; segp[ct] = 0; sege[ct] = 0;
    if (a.seg_start != nullptr && sidx[ct] < a.n_clips) {
      int k = a.seg_off[sidx[ct]];
      sege[ct] = a.seg_off[sidx[ct] + 1];
      while (k < sege[ct] && a.seg_start[k] < a.t0) ++k;
      if (k < sege[ct] && a.seg_start[k] == a.t0 && a.t0 > 0) { hreg[ct][0] = 0.f; hreg[ct][1] = 0.f; }
      while (k < sege[ct] && a.seg_start[k] <= a.t0) ++k;
      segp[ct] = k;
      nstart[ct] = k < sege[ct] ? a.seg_start[k] : 0x7fffffff;
    }
  }

  // ---- exchange buffers; relu(h) output through a buffer resource (dead lanes: offset past num_records, instruction still issued)
  const int buf_stride = a.G * GROUP_BYTES;
  char* hx_base = (char*)a.hx + (size_t)g * GROUP_BYTES;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)hx_base, 0, buf_stride + GROUP_BYTES, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_hr = __builtin_amdgcn_make_buffer_rsrc(a.h_relu_out, 0, 0x7FFFFFF0, 0x00020000);
  constexpr int OOR = 0x7FFFFFF8;                 // >= num_records of both resources
  auto publish = [&](int ct, int buf, unsigned tag, bool zero, bool cond) {
    const int off = buf * buf_stride + ((ucol / KF) * GRU_MAX_TILES + ct) * 1024 + ((((ucol % KF) / EPL) << 4) + l15) * 16 + (ucol % EPL) * 2;
    f32x2 hv = {zero ? 0.f : hreg[ct][0], zero ? 0.f : hreg[ct][1]};
    hv[0] = __builtin_amdgcn_fmed3f(hv[0], -1.9921875f, 1.9921875f);
    hv[1] = __builtin_amdgcn_fmed3f(hv[1], -1.9921875f, 1.9921875f);
    const unsigned v = (op16<OT>::pack2(hv[0], hv[1]) & 0xBFFFBFFFu) | (tag ? 0x40004000u : 0u);
    if (local) __builtin_amdgcn_raw_buffer_store_b32(v, rs, cond ? off : OOR, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(v, rs, cond ? off : OOR, 0, AUX_SC1);
  };
  // gi of one step for tile ct -> LDS (three 4-byte DMA per wave: counted instructions 1-3 of a slot); lanes whose slot has ended read
  // row 0 of the step, which always exists, and ignore it
  auto gi_dma = [&](int par, int ct, int na, int rbase) {
    const int r = rbase + (sidx[ct] < na ? sidx[ct] : 0);
    const bf16_t* src = (const bf16_t*)a.gi + (size_t)r * (3 * HID) + ucol;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(gism_lds + (unsigned)(((par * NCT + ct) * 3) * 1024 + q * 256)));
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(src + gate * HID), "s"(dst + (unsigned)gate * 1024u) : "memory", "m0");
  };
  // gather issue: the 8 fragments of tile ct from exchange buffer rbuf into LDS image `img` (0 / 1) of this wave; dead columns are
  // not fetched (their lanes are masked off: the image keeps whatever it held, the columns are never used)
  auto issue = [&](int img, int rbuf, int ct, bool col_live, bool sc1) {
    const char* src = hx_base + (size_t)rbuf * buf_stride + (size_t)((q * NKS) * GRU_MAX_TILES + ct) * 1024 + lane * 16;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(gsm_lds + (unsigned)((img * 4 + q) * NKS) * 1024u));
    if (col_live) {
      if (!sc1) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) GRU_MT_DMA(src + (size_t)ks * GRU_MAX_TILES * 1024, dst + (unsigned)ks * 1024u, "nt");
      } else {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) GRU_MT_DMA(src + (size_t)ks * GRU_MAX_TILES * 1024, dst + (unsigned)ks * 1024u, "sc1");
      }
    }
  };
  auto fetch = [&](u32x4 (&hbr)[NKS], int img) {           // the wave's own image -> fragment registers
    const u32x4* p = (const u32x4*)(gsm + (size_t)((img * 4 + q) * NKS) * 1024) + lane;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) hbr[ks] = p[ks * 64];
  };

  typedef const __attribute__((address_space(4))) int* cint_p;
  cint_p nact_c = (cint_p)a.nact;
  cint_p rowoff_c = (cint_p)a.rowoff;
  const int nsteps = a.t1 - a.t0;
  int na_c = nact_c[a.t0], rb_c = rowoff_c[a.t0];
  int na_n = nsteps > 1 ? nact_c[a.t0 + 1] : 0, rb_n = nsteps > 1 ? rowoff_c[a.t0 + 1] : 0;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
    if (tfirst[ct] < na_c) {
      publish(ct, 1, 1u, false, true);
      gi_dma(0, ct, na_c, rb_c - a.row_base);      // step tl = 0 reads parity 0
    }
  int parity = 0;
  bool pre0 = false;                              // tile 0 of the step about to run was prefetched into image 0
  // debug stamps (PREGO_GRU_STAMPS=1), workgroup 0 / wave 0, cycle sums per tile-slot phase: 0 gather wait, 1 LDS fetch + MFMA with the tag
  // check and the next slot's DMA inside, 2 = COUNT of tiles redone on stale data, 3 gi DMA issue, 4 LDS reduction write + barrier, 5 gates, 6 stores; [7] = tile-slots; retries in [5]'s slot of the classic layout is not kept
  const bool stamp = a.stamps != nullptr && blockIdx.x == 0 && q == 0;
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_t = 0;
#ifdef GRU_MT_STAMPS      // diagnostic build only (scripts/probes/mt_stamps.py): the stamp branches cut every phase into its own basic block
#define MSTAMP(i) do { if (stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_t; st_t = n_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define MSTAMP(i) do { } while (0)
#endif

  auto step = [&](const int tl) -> bool {
    const int t = a.t0 + tl;
    const int na = na_c;
    const int rbase = rb_c - a.row_base;
    const bool more = tl + 1 < nsteps;
    const int t2 = (tl + 2 < nsteps) ? t + 2 : t;
    const int na_2 = nact_c[t2], rb_2 = rowoff_c[t2];
    const int rbuf = (tl + 1) & 1;
    const unsigned etag = (unsigned)(((tl - 1) >> 1) & 1);
    const unsigned eword = etag ? TAGM : 0u;
    const unsigned untag = etag ? ~TAGM : 0xFFFFFFFFu;
    // what the NEXT step's tile-0 gather expects (issued at the end of this step when two or more tiles are alive)
    const int rbuf_n = tl & 1;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      if (tfirst[ct] < na) {
        u32x4 HB[NKS];
        const bool col_live = sidx[ct] < na;
        // ---- (1) this tile's h_{t-1}: prefetched one slot ago (hand-counted wait), or fetched now (first step of the launch, or
        // the previous step had a single live tile)
        const bool prefetched = ct > 0 || pre0;
#ifdef GRU_MT_STAMPS
        if (stamp) { st_t = __builtin_amdgcn_s_memtime(); st_acc[7] += 1; }
#endif
        if (prefetched) {
          asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
          issue(ct & 1, rbuf, ct, col_live, !local);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        MSTAMP(0);
        fetch(HB, ct & 1);
        // ---- where the next tile-slot's gather comes from (decided before the multiply: its eight DMA go out BETWEEN the MFMAs)
        const bool next_is_tile = ct + 1 < NCT && tfirst[ct + 1 < NCT ? ct + 1 : ct] < na;
        int nx_img = 0, nx_rbuf = rbuf, nx_ct = 0;
        bool nx_any = false, nx_live = false;
        if (ct + 1 < NCT && next_is_tile) {
          nx_any = true; nx_img = (ct + 1) & 1; nx_ct = ct + 1; nx_live = sidx[ct + 1 < NCT ? ct + 1 : ct] < na;
        } else if (ct > 0) {                         // last live tile of this step, and not the only one: tile 0 of step t + 1
          pre0 = more && tfirst[0] < na_n;
          nx_any = pre0; nx_img = 0; nx_rbuf = rbuf_n; nx_ct = 0; nx_live = sidx[0] < na_n;
        } else {
          pre0 = false;                              // a single live tile: its next h does not exist yet
        }
        const char* nx_src = hx_base + (size_t)nx_rbuf * buf_stride + (size_t)((q * NKS) * GRU_MAX_TILES + nx_ct) * 1024 + lane * 16;
        const unsigned nx_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(gsm_lds + (unsigned)((nx_img * 4 + q) * NKS) * 1024u));
        const bool nx_issue = nx_any && nx_live;
        // ---- tag check + multiply in a fixed order.  With four tiles per group a prefetched image is two or more slots old and
        // practically never stale (10 of 2 048 slots in the stamped run): the check (64 VALU operations) then rides in the MFMAs'
        // issue shadow and a stale tile is fetched again and REDONE; with two tiles the image is one slot old and stale half of the
        // time (measured), so the check comes first there
        f32x4 acc[3][UT];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        auto refetch = [&]() -> bool {                   // until every fragment shows the expected tag; false = timed out
          unsigned spins = 0;
          for (;;) {
            if (++spins > SPIN_LIMIT) {
              if (lane == 0) __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              return false;
            }
            if ((spins & 255u) == 0u) {
              if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
            }
            issue(ct & 1, rbuf, ct, col_live, !(local && spins < 6u));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fetch(HB, ct & 1);
            unsigned b2 = 0u;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
              b2 |= ((HB[ks][0] ^ eword) | (HB[ks][1] ^ eword)) | ((HB[ks][2] ^ eword) | (HB[ks][3] ^ eword));
            if (__all(!col_live || (b2 & TAGM) == 0u)) return true;
          }
        };
        unsigned badv = 0u;
        if constexpr (!SPEC) {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks)
            badv |= ((HB[ks][0] ^ eword) | (HB[ks][1] ^ eword)) | ((HB[ks][2] ^ eword) | (HB[ks][3] ^ eword));
          if (!__all(!col_live || (badv & TAGM) == 0u)) {
            if (!refetch()) return false;
          }
          badv = 0u;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          u32x4 v = HB[ks];
          // the xor that tests the tag also removes it (round 6, as in the classic kernel): one VALU per register instead of two
          v[0] ^= eword; v[1] ^= eword; v[2] ^= eword; v[3] ^= eword;
          if constexpr (SPEC) badv |= (v[0] | v[1]) | (v[2] | v[3]);
          const bf16x8 bfrag = __builtin_bit_cast(bf16x8, v);
#pragma unroll
          for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) acc[gate][ut] = op16<OT>::mfma(wb[gate][ut][ks], bfrag, ks == 0 ? zero4 : acc[gate][ut]);
        }
        if constexpr (SPEC) {
          if (!__all(!col_live || (badv & TAGM) == 0u)) {
            if (!refetch()) return false;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {           // the same products, same order, on the valid fragments
              u32x4 v = HB[ks];
              v[0] ^= eword; v[1] ^= eword; v[2] ^= eword; v[3] ^= eword;          // valid fragments: the xor is the un-tagging
              const bf16x8 bfrag = __builtin_bit_cast(bf16x8, v);
#pragma unroll
              for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                for (int ut = 0; ut < UT; ++ut) acc[gate][ut] = op16<OT>::mfma(wb[gate][ut][ks], bfrag, ks == 0 ? zero4 : acc[gate][ut]);
            }
          }
        }
        MSTAMP(1);
        // ---- (2) the next tile-slot's gather goes out now, under this tile's reduction and gate phase (behind the MFMAs: a
        // vector-memory instruction that waits for queue space blocks the wave's whole instruction stream, MFMAs included -
        // issuing the eight DMA BETWEEN the MFMAs measured slower, 8.66 against 7.62 us per four-tile step)
        if (nx_issue) {
          if (local) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) GRU_MT_DMA(nx_src + (size_t)ks * GRU_MAX_TILES * 1024, nx_dst + (unsigned)ks * 1024u, "nt");
          } else {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) GRU_MT_DMA(nx_src + (size_t)ks * GRU_MAX_TILES * 1024, nx_dst + (unsigned)ks * 1024u, "sc1");
          }
        }
        // ---- (3) gi of step t + 1 for this tile (always issued: counted instructions 1-3; rows clamp to a live one)
        gi_dma((tl + 1) & 1, ct, more ? na_n : na, (more ? rb_n : rb_c) - a.row_base);
        MSTAMP(3);
        // ---- (4) K-quarter reduction through LDS
        f32x4* redw = red + parity * RED_STRIDE;
        parity ^= 1;
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
          for (int ut = 0; ut < UT; ++ut) redw[((q * 3 + gate) * UT + ut) * 64 + lane] = acc[gate][ut];
        __syncthreads();
        MSTAMP(4);
        // ---- (5) gates + state update
        {
          const f32x2* rp = (const f32x2*)((const float*)&redw[own_ut * 64 + lane] + own_r0);
          f32x2 part[3][4];
#pragma unroll
          for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) part[gate][qq] = rp[((qq * 3 + gate) * UT) * 64 * 2];
          asm volatile("" : "+v"(part[0][0]), "+v"(part[0][1]), "+v"(part[0][2]), "+v"(part[0][3]), "+v"(part[1][0]), "+v"(part[1][1]),
                            "+v"(part[1][2]), "+v"(part[1][3]), "+v"(part[2][0]), "+v"(part[2][1]), "+v"(part[2][2]), "+v"(part[2][3]));
          f32x2 gh[3];
#pragma unroll
          for (int gate = 0; gate < 3; ++gate) gh[gate] = (part[gate][0] + part[gate][1]) + (part[gate][2] + part[gate][3]);
          if (col_live) {
            const f32x2 one = {1.f, 1.f};
            f32x2 gr, gz, gn;
            const unsigned* gl = gism + ((tl & 1) * NCT + ct) * 3 * 256 + tid;      // landed before this slot's gather wait
            const unsigned ur = gl[0], uz = gl[256], un = gl[512];
            gr = (f32x2){op16<OT>::lo(ur), op16<OT>::hi(ur)};
            gz = (f32x2){op16<OT>::lo(uz), op16<OT>::hi(uz)};
            gn = (f32x2){op16<OT>::lo(un), op16<OT>::hi(un)};
            const f32x2 hp = {hreg[ct][0], hreg[ct][1]}, bh = {bhn[0], bhn[1]};
            const f32x2 xr = gr + gh[0], xz = gz + gh[1];
            f32x2 r, z, n;
            r[0] = sigmoidf_(xr[0]); r[1] = sigmoidf_(xr[1]);
            z[0] = sigmoidf_(xz[0]); z[1] = sigmoidf_(xz[1]);
            const f32x2 ghn = gh[2] + bh;
            const f32x2 xn = gn + r * ghn;
            n[0] = tanhf_(xn[0]); n[1] = tanhf_(xn[1]);
            const f32x2 hn = (one - z) * n + z * hp;
            hreg[ct][0] = hn[0]; hreg[ct][1] = hn[1];
          }
        }
        MSTAMP(5);
        // ---- (6) publish h_t (counted instruction 4) and relu(h_t) (counted instruction 5): always issued
        const bool restart = (t + 1 == nstart[ct]);
        publish(ct, tl & 1, (unsigned)((tl >> 1) & 1), restart, more && sidx[ct] < na_n);
        {
          const unsigned hv = op16<OT>::pack2(fmaxf(hreg[ct][0], a.out_floor), fmaxf(hreg[ct][1], a.out_floor));
          const long long o = ((long long)(rbase + sidx[ct]) * HID + ucol) * 2;
          __builtin_amdgcn_raw_buffer_store_b32(hv, rs_hr, (col_live && a.h_relu_out != nullptr) ? (int)o : OOR, 0, 0);
        }
        if (restart) {
          hreg[ct][0] = 0.f; hreg[ct][1] = 0.f;
          ++segp[ct];
          nstart[ct] = segp[ct] < sege[ct] ? a.seg_start[segp[ct]] : 0x7fffffff;
          asm volatile("" : "+v"(nstart[ct]));
        }
        MSTAMP(6);
      }
    }
    na_c = na_n; rb_c = rb_n; na_n = na_2; rb_n = rb_2;
    return true;
  };
  for (int tl = 0; tl < nsteps; ++tl) {
    if (tfirst[0] >= na_c) break;
    if (!step(tl)) return;
  }
  if (stamp && lane == 0) {
    for (int i = 0; i < 8; ++i) a.stamps[i] += st_acc[i];
  }
#undef MSTAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // a prefetch issued by the last step may still be in flight
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
    if (sidx[ct] < a.n_clips) {
      a.h_state[(size_t)sidx[ct] * HID + ucol] = hreg[ct][0];
      a.h_state[(size_t)sidx[ct] * HID + ucol + 1] = hreg[ct][1];
    }
}

// (Round 5, measured and removed: a BATCHED two-tile step - both tiles of a pair through each phase together: 16 fragment loads in
// flight, 96 MFMAs behind one wait chain, one LDS reduction + barrier, the two gate chains interleaved in one basic block; bit-identical
// to this file's kernels.  7.62 us per four-tile step against the pipelined kernel's 7.25 on the same device (synth512: 27.7 vs 28.3 M
// frames/s): what a tile costs is moving its 32 KB through the CU's vector-memory path in the wave's own time (~850 cycles) and the serial
// reduce / gate / store phases, not the L2 round trip the batching amortises - and 411-469 registers put the fragments in AGPRs.)
// Returns 0 on success, -1 for unsupported (hid, nct).
// a.hx must hold [2][G][GROUP_BYTES] bytes (gru_hx_bytes); both buffers are re-armed here (stream ordered):
// buffer 0 := tag 1 everywhere (first expected tag there is 0), buffer 1 := tag 0 (first expected tag is 1).
size_t gru_hx_bytes(bool bf16, int hid, int G) {
  return (size_t)2 * G * (bf16 ? hid / 32 : hid / 16) * GRU_MAX_TILES * 1024;
}
int gru_max_tiles() { return 4; }   // kernels are instantiated for 1, 2 and 4 live tiles (8 spills registers)

// re-arm both exchange buffers and the rendezvous words in ONE launch (three hipMemsetAsync calls per recurrence launch were
// 0.7 ms per pass of fill kernels plus their launch gaps): buffer 0 := tag 1 everywhere, buffer 1 := 0, sync[0..15] := 0
__global__ void gru_arm_kernel(unsigned* __restrict__ hx, size_t words_per_buf, unsigned pattern, unsigned* __restrict__ sync) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = i; k < words_per_buf / 4; k += stride) {
    ((uint4*)hx)[k] = make_uint4(pattern, pattern, pattern, pattern);
    ((uint4*)(hx + words_per_buf))[k] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (sync != nullptr && i < 16) sync[i] = 0u;
}

GruArm gru_arm_desc(bool bf16, int hid, int G, void* hx, unsigned* sync) {
  return GruArm{(unsigned*)hx, (unsigned long long)(gru_hx_bytes(bf16, hid, G) / 2 / 4), bf16 ? 0x40004000u : 0x40000000u, sync};
}

// The recurrence of a whole split pass: XCDs 0 .. a.Gd - 1 (needs the verified placement of an earlier full-width launch: rendezvous
// word 20), one clip tile per group, GI ring + chunk counters of GruArgs' pass fields.  -1: unsupported.
// The exchange buffers / rendezvous words must have been armed by launch_gru_arm EARLIER in the stream, before the feed-forward launch of
// the pass is released: an ordinary kernel queued between the two persistent launches would need workgroups on CUs the feed-forward
// kernel fills completely (2 x 256 registers per SIMD) and would hold the recurrence up until the feed-forward has finished - which
// waits for the recurrence.
void launch_gru_arm(bool bf16, int hid, int G, void* hx, unsigned* sync, hipStream_t s) {
  gru_arm_kernel<<<256, 256, 0, s>>>((unsigned*)hx, gru_hx_bytes(bf16, hid, G) / 2 / 4, bf16 ? 0x40004000u : 0x40000000u, sync);
}
int launch_gru_recurrence_pass(int hid, GruArgs a, hipStream_t s) {
  if (hid != 1024 || a.G != 8 || a.Gd < 1 || a.Gd >= 8 || !a.gi_bf16 || !a.gi_cnt || !a.rec_cnt || a.sync == nullptr) return -1;
  const size_t lds = (size_t)2 * 4 * 3 * 2 * 64 * 16;
  if (a.f16) gru_recurrence_kernel<f16_t, 1024, 2, 1, false, true, true><<<256, 256, lds, s>>>(a);
  else gru_recurrence_kernel<bf16_t, 1024, 2, 1, false, true, true><<<256, 256, lds, s>>>(a);
  return 0;
}

// other hidden sizes (rnn.py:31: any cfg['hidden_dim']): the classic kernel, inference instantiations only.  What decides is the
// weight slice a workgroup keeps in registers (3 gates x 16 UT rows x H / 4 per wave): 16-bit operands H = 512 (UT = 2: 96 registers) and
// H = 2048 (UT = 1: 192, 128 workgroups per group = two groups of four XCDs, sc1 hand-off); fp32 operands H = 512 (UT = 1: 96)
template <int HID>
static int launch_gru_recurrence_other(bool bf16, int nct, const GruArgs& a, int grid, hipStream_t s) {
#define LAUNCH_O(WT, UT, NCT)                                                                                    \
  do {                                                                                                             \
    const size_t lds = (size_t)2 * 4 * 3 * UT * 64 * 16;                                                           \
    if (a.gi_bf16) gru_recurrence_kernel<WT, HID, UT, NCT, false, true><<<grid, 256, lds, s>>>(a);                \
    else gru_recurrence_kernel<WT, HID, UT, NCT, false><<<grid, 256, lds, s>>>(a);                                \
  } while (0)
  constexpr int UTB = HID <= 1024 ? 2 : 1;
  // training (round 6: rnn.py:31-38 takes any hidden_dim, and so does trainer/train.py): the instantiation that keeps the gate
  // activations / raw state for BPTT (fp32 GI; bf16 or - H = 512 - exact-fp32 operands)
  if (a.keep_r != nullptr || a.h_raw_out != nullptr) {
#define LAUNCH_T(WT, UT, NCT)                                                                                    \
    do {                                                                                                           \
      const size_t lds = (size_t)2 * 4 * 3 * UT * 64 * 16;                                                         \
      gru_recurrence_kernel<WT, HID, UT, NCT, true><<<grid, 256, lds, s>>>(a);                                    \
    } while (0)
    if (a.gi_bf16 || a.f16) return -1;
    if (bf16) { if (nct == 1) LAUNCH_T(bf16_t, UTB, 1); else if (nct == 2) LAUNCH_T(bf16_t, UTB, 2); else if (nct <= 4) LAUNCH_T(bf16_t, UTB, 4); else return -1; }
    else {
      if constexpr (HID > 512) return -1;
      else { if (nct == 1) LAUNCH_T(float, 1, 1); else if (nct == 2) LAUNCH_T(float, 1, 2); else if (nct <= 4) LAUNCH_T(float, 1, 4); else return -1; }
    }
#undef LAUNCH_T
    return 0;
  }
  if (bf16) {
    if (!a.gi_bf16) return -1;                 // 16-bit operands run with 16-bit GI (inference)
    if (a.f16) { if (nct == 1) LAUNCH_O(f16_t, UTB, 1); else if (nct == 2) LAUNCH_O(f16_t, UTB, 2); else if (nct <= 4) LAUNCH_O(f16_t, UTB, 4); else return -1; }
    else { if (nct == 1) LAUNCH_O(bf16_t, UTB, 1); else if (nct == 2) LAUNCH_O(bf16_t, UTB, 2); else if (nct <= 4) LAUNCH_O(bf16_t, UTB, 4); else return -1; }
  } else {
    if constexpr (HID > 512) return -1;
    else { if (a.gi_bf16) return -1; if (nct == 1) LAUNCH_O(float, 1, 1); else if (nct == 2) LAUNCH_O(float, 1, 2); else if (nct <= 4) LAUNCH_O(float, 1, 4); else return -1; }
  }
#undef LAUNCH_O
  return 0;
}
// workgroups per recurrence group for (operand type, hidden size): H / (16 UT)
int gru_group_size(bool bf16, int hid) { return hid / (bf16 && hid <= 1024 ? 32 : 16); }
bool gru_hidden_supported(bool bf16, int hid) { return hid == 1024 || hid == 512 || (bf16 && hid == 2048); }

int launch_gru_recurrence(bool bf16, int hid, int nct, GruArgs a, hipStream_t s) {
  if (hid != 1024) {
    if (!gru_hidden_supported(bf16, hid)) return -1;
    if (!a.armed) gru_arm_kernel<<<256, 256, 0, s>>>((unsigned*)a.hx, gru_hx_bytes(bf16, hid, a.G) / 2 / 4, bf16 ? 0x40004000u : 0x40000000u, a.sync);
    const int grid = a.G * gru_group_size(bf16, hid);
    return hid == 512 ? launch_gru_recurrence_other<512>(bf16, nct, a, grid, s) : launch_gru_recurrence_other<2048>(bf16, nct, a, grid, s);
  }
  const int P = bf16 ? 32 : 64;
  const size_t buf_bytes = gru_hx_bytes(bf16, hid, a.G) / 2;
  if (!a.armed) gru_arm_kernel<<<256, 256, 0, s>>>((unsigned*)a.hx, buf_bytes / 4, bf16 ? 0x40004000u : 0x40000000u, a.sync);
  const int grid = a.G * P;
  const bool train = a.keep_r != nullptr || a.h_raw_out != nullptr;
  // two or more clip tiles per group, inference: the software-pipelined multi-tile kernel (PREGO_GRU_NO_MT=1: the classic kernel, A/B)
  const bool no_mt = a.no_mt != 0;
#ifdef GRU_MT_STAMPS
  const bool stamps_ok = true;
#else
  const bool stamps_ok = a.stamps == nullptr;       // PREGO_GRU_STAMPS=1 on a production build: the classic kernel carries the stamps
#endif
  // the pipelined kernel stores relu(h) through a buffer resource (32-bit byte offsets): launches over more than 2^31 bytes of rows
  // (a caller-chosen chunk of a million rows) stay on the classic kernel and its 64-bit addressing
  const bool rows_ok = a.rows > 0 && (long long)a.rows * hid * 2 < (1ll << 31) - 65536;
  if (bf16 && a.gi_bf16 && nct >= 2 && nct <= 4 && !train && !no_mt && stamps_ok && rows_ok) {
    static const int spec_env = prego_tune_env("PREGO_GRU_MT_SPEC") ? atoi(prego_tune_env("PREGO_GRU_MT_SPEC")) : -1;     // A/B: tag check inside (1) / before (0) the multiply
    const bool spec = spec_env >= 0 ? spec_env != 0 : nct >= 3;
    // reduction buffers + two gather images per wave + the gi rows of two steps
    const size_t lds = (size_t)2 * 4 * 3 * 2 * 64 * 16 + (size_t)2 * 4 * 8 * 1024 + (size_t)2 * (nct == 2 ? 2 : 4) * 3 * 1024;
#define LAUNCH_MT(WT, NCT)                                                                                   \
    do {                                                                                                       \
      static DeviceOnce once;                                                                                  \
      once.run([&] {                                                                                           \
        (void)hipFuncSetAttribute((const void*)gru_recurrence_mt_kernel<WT, 1024, NCT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
        (void)hipFuncSetAttribute((const void*)gru_recurrence_mt_kernel<WT, 1024, NCT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      });                                                                                                      \
      if (spec) gru_recurrence_mt_kernel<WT, 1024, NCT, true><<<grid, 256, lds, s>>>(a);                       \
      else gru_recurrence_mt_kernel<WT, 1024, NCT, false><<<grid, 256, lds, s>>>(a);                           \
    } while (0)
    if (a.f16) { if (nct == 2) LAUNCH_MT(f16_t, 2); else LAUNCH_MT(f16_t, 4); }
    else { if (nct == 2) LAUNCH_MT(bf16_t, 2); else LAUNCH_MT(bf16_t, 4); }
#undef LAUNCH_MT
    return 0;
  }
#define LAUNCH(WT, UT, NCT)                                                                            \
  do {                                                                                                 \
    const size_t lds = (size_t)2 * 4 * 3 * UT * 64 * 16;                                               \
    if (train) gru_recurrence_kernel<WT, 1024, UT, NCT, true><<<grid, 256, lds, s>>>(a);              \
    else if (a.gi_bf16) gru_recurrence_kernel<WT, 1024, UT, NCT, false, true><<<grid, 256, lds, s>>>(a); \
    else gru_recurrence_kernel<WT, 1024, UT, NCT, false><<<grid, 256, lds, s>>>(a);                    \
  } while (0)
  if (bf16 && a.f16) {
    if (train) return -1;                    // fp16 operands: inference only (training runs on bf16 / fp32 handles)
    if (nct == 1) LAUNCH(f16_t, 2, 1);
    else if (nct == 2) LAUNCH(f16_t, 2, 2);
    else if (nct <= 4) LAUNCH(f16_t, 2, 4);
    else return -1;
  } else if (bf16) {
    if (nct == 1) LAUNCH(bf16_t, 2, 1);
    else if (nct == 2) LAUNCH(bf16_t, 2, 2);
    else if (nct <= 4) LAUNCH(bf16_t, 2, 4);
    else return -1;
  } else {
    if (nct == 1) LAUNCH(float, 1, 1);
    else if (nct == 2) LAUNCH(float, 1, 2);
    else if (nct <= 4) LAUNCH(float, 1, 4);
    else return -1;
  }
#undef LAUNCH
  return 0;
}
