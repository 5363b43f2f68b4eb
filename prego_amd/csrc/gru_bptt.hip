// Persistent reverse-time recurrence of BPTT through nn.GRU's cell (model/rnn/rnn.py:61; loss.backward(), trainer/train.py:23).
// Replaces the step-by-step loop "elementwise kernel + 16-row GEMM" (two dependent launches per time step, 8 workgroups
// in the GEMM: 27 us per step on a 128-frame window) by ONE launch that keeps W_hh in registers, like the forward
// recurrence (gru_recurrence.hip):
//   for t = T-1 .. 0, for the clips alive at t (sorted prefix) and every hidden unit j:
//     dh      = dHout[row(t,b)][j] + (clip alive at t+1 ? dh_{t+1} z_{t+1} + (dGH_{t+1} W_hh)[b][j] : 0)
//     dn = dh (1-z); dz = dh (h_{t-1} - n); dpre_n = dn (1-n^2); dpre_r = dpre_n ghn r (1-r); dpre_z = dz z (1-z)
//     dGI[row] = [dpre_r | dpre_z | dpre_n]      dGH[row] = [dpre_r | dpre_z | dpre_n r]      (fp32 + operand-dtype copies)
// Decomposition = the forward kernel's: G groups x P workgroups, a workgroup owns 16*UT output units j (columns of dh) and
// holds its slice of W_hh^T ([j][3H], K = 3H = 3072) as MFMA A-fragments in VGPRs, wave q the K-quarter q (192 VGPRs).
// Per step a workgroup needs the whole dGH_{t+1} of its group's clips (3 x the forward's h): an all-gather through a
// FIVE-buffered exchange buffer in fragment-major layout.  Gradients are not bounded, so the forward's tag-bit-in-the-data validity
// trick does not apply as it stands; since round 5 the data still IS its own flag: a word of all ones (a bf16 pair / an fp32 value no
// publish ever writes: such a NaN is stored with its lowest bit cleared) means "not yet published".  A consumer re-loads its
// fragments until no live word is all ones; a producer, at the TOP of step t, resets what it published at step t+3 (buffer
// (t+3) % 5, which it will write again at step t-2).  Why five buffers and three steps: (WAR) at the top of its step t a workgroup
// has gathered everyone's step t+2 data, so everyone has finished the gate math of step t+2 and with it the gather of step t+3 -
// nobody reads that buffer's old contents any more; (RAW) a consumer polls for step t-2 data once it has finished step t-2, which
// took everyone's step t-1 data, and a workgroup's step t-1 (and step t) publish was issued after its step-t reset had been
// acknowledged (the wave waited for loads it issued behind the reset stores: vmcnt retires in order) - so stale step t+3 data can
// never be taken for step t-2 data.  (Rounds 3-4: one EPOCH WORD per producer behind a drain of the publish stores + a barrier, and
// a poll of the group's epoch words before the gather - two more dependent L2 round trips per step: 3.05 us per step, now 2.2.)  As in the forward kernel the
// launch first VERIFIES placement (every workgroup of a group reports its XCC id): a group that sits on one XCD hands off
// with plain stores + L1-bypassing nt loads through that XCD's L2, any other placement with sc1 stores + sc1 loads (the
// agent-scope release/acquire FENCES this replaced wrote back and invalidated the whole L2 every step: 30 us per step).
// Every spin is bounded (abort word).
// Training batches are small (16 windows x 128 frames in the reference, one clip tile of one group), so this kernel is
// written for low launch count first: the 128-step loop drops from 3.5 ms to < 0.5 ms (train step 6.2 -> 2.7 ms).
#include "common.h"
#include "kernels.h"

#define BPTT_SPIN_LIMIT (1u << 22)
#define BPTT_MAX_TILES 4
#define BPTT_BUFS 5            // exchange buffers (header comment)

template <typename WT, int HID, int UT, int NCT>
__global__ __launch_bounds__(256, 1) void gru_bptt_kernel(BpttArgs a) {
  constexpr bool BF = (sizeof(WT) == 2);
  constexpr int UNITS = 16 * UT;              // output units (columns of dh) owned by this workgroup
  constexpr int K3 = 3 * HID;                 // contraction length
  constexpr int KQ = K3 / 4;                  // K range per wave
  constexpr int KF = BF ? 32 : 16;            // k per fragment
  constexpr int EPL = BF ? 8 : 4;             // elements per lane and fragment
  constexpr int NKS = KQ / KF;                // fragments per wave and clip tile
  constexpr int NFR = K3 / KF;                // fragments of one clip tile's dGH
  constexpr int OWN_R = UT == 2 ? 2 : 1;
  constexpr int GROUP_BYTES = NFR * BPTT_MAX_TILES * 1024;
  constexpr int P = HID / UNITS;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = (f32x4*)smem;                  // [2 parities][4 waves][UT][64 lanes]
  constexpr int RED_STRIDE = 4 * UT * 64;

  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x % a.G, w = blockIdx.x / a.G;
  // clips are dealt to the groups interleaved (clip s of a layer of 16 G -> group s % G, column s / G), as in the forward
  // recurrence: a train.py batch of 16 windows then runs as 2 columns on each of the 8 XCDs instead of 16 columns on one, and
  // the gather below fetches live columns only, so an XCD's L2 serves 1/8 of the 96 KB-per-CU dGH burst
  if (g >= a.n_clips) return;                 // group without clips
  const int l15 = lane & 15, l4 = lane >> 4;
  // ---- placement: does the whole group sit on one XCD? (sync[16+g] = XCC id mask, sync[32+g] = arrivals) ----------------
  __shared__ int s_local;
  if (tid == 0) {
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;          // HW_REG_XCC_ID[2:0]
    __hip_atomic_fetch_or(a.sync + 16 + g, 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(a.sync + 32 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    int loc = 1;
    while (__hip_atomic_load(a.sync + 32 + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)P) {
      if (++spins > BPTT_SPIN_LIMIT) { __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); loc = -1; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    if (loc > 0) {
      const unsigned m = __hip_atomic_load(a.sync + 16 + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      loc = (m & (m - 1)) == 0u ? 1 : 0;
    }
    s_local = loc;
  }
  __syncthreads();
  if (s_local < 0) return;
  const bool local = s_local == 1 && !a.force_sc1;

  // ---- resident weights: W_hh^T rows (unit j) x this wave's K-quarter -------------------------------------------------
  bf16x8 wb[BF ? UT : 1][BF ? NKS : 1];
  float wf[BF ? 1 : UT][BF ? 1 : NKS][4];
#pragma unroll
  for (int ut = 0; ut < UT; ++ut)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const size_t row = (size_t)(w * UNITS + ut * 16 + l15);
      const int k = q * KQ + ks * KF + EPL * l4;
      if constexpr (BF) {
        wb[ut][ks] = *(const bf16x8*)((const bf16_t*)a.whhT + row * K3 + k);
      } else {
        const float4 v = *(const float4*)((const float*)a.whhT + row * K3 + k);
        wf[ut][ks][0] = v.x; wf[ut][ks][1] = v.y; wf[ut][ks][2] = v.z; wf[ut][ks][3] = v.w;
      }
    }

  // ---- ownership in the elementwise phase (as in the forward kernel) ---------------------------------------------------
  const int own_ut = UT == 2 ? (q & 1) : 0;
  const int own_r0 = UT == 2 ? (q >> 1) * 2 : q;
  const int ucol = w * UNITS + own_ut * 16 + l4 * 4 + own_r0;
  int sidx[NCT], tfirst[NCT];
  float carry[NCT][OWN_R];                    // dh_{t+1} * z_{t+1} of my (unit, clip) pairs
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    sidx[ct] = ct * 16 * a.G + l15 * a.G + g;
    tfirst[ct] = ct * 16 * a.G + g;
#pragma unroll
    for (int e = 0; e < OWN_R; ++e) carry[ct][e] = 0.f;
  }
  const int buf_stride = a.G * GROUP_BYTES;
  char* hx_base = (char*)a.hx + (size_t)g * GROUP_BYTES;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)hx_base, 0, (BPTT_BUFS - 1) * buf_stride + GROUP_BYTES, 0x00020000);
  // byte offset of my element (clip l15 of tile ct, k = gate * H + my unit) inside a buffer
  auto elem_off = [&](int ct, int gate) -> int {
    const int k = gate * HID + ucol;
    return ((k / KF) * BPTT_MAX_TILES + ct) * 1024 + ((((k % KF) / EPL) << 4) + l15) * 16 + (k % EPL) * (int)sizeof(WT);
  };

  // plan tables through the scalar path (constant address space), one step ahead: vector loads here put a vmcnt(0) and an
  // L2 round trip at the top of every step
  typedef const __attribute__((address_space(4))) int* cint_p;
  cint_p nact_c = (cint_p)a.nact;
  cint_p rowoff_c = (cint_p)a.rowoff;
  int na_cur = nact_c[a.t_max - 1], na_prev = 0;                           // nact[t], nact[t+1]
  int ro_cur = rowoff_c[a.t_max - 1];                                       // rowoff[t]
  int na_nx = a.t_max > 1 ? nact_c[a.t_max - 2] : 0, ro_nx = a.t_max > 1 ? rowoff_c[a.t_max - 2] : 0;   // step t-1
  for (int t = a.t_max - 1; t >= 0; --t) {
    const int na = na_cur;
    const int na_next = na_prev;
    const int row_t = ro_cur;
    const int row_tm1 = ro_nx;                                              // rowoff[t-1] (0 at t == 0: unused)
    const int t2 = t >= 2 ? t - 2 : 0;
    const int na_2 = nact_c[t2], ro_2 = rowoff_c[t2];                       // look-ahead for step t-2
    na_prev = na_cur; na_cur = na_nx; ro_cur = ro_nx; na_nx = na_2; ro_nx = ro_2;
    if (tfirst[0] >= na) continue;            // nothing of this group alive yet (clips are sorted longest first)
    const int pbuf = t % BPTT_BUFS;           // where this step publishes
    const int rbuf = (t + 1) % BPTT_BUFS;     // where step t+1 published
    // ---- reset what I published at step t+3 (header comment): FIRST in the step, so that waiting for any later load of this wave
    // proves these stores acknowledged before this step's publish is issued
    if (t + 3 < a.t_max) {
      const int zbuf = (t + 3) % BPTT_BUFS;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        if (sidx[ct] < na) {
#pragma unroll
          for (int gate = 0; gate < 3; ++gate) {
            if (local) __builtin_amdgcn_raw_buffer_store_b32(0xFFFFFFFFu, rs, zbuf * buf_stride + elem_off(ct, gate), 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(0xFFFFFFFFu, rs, zbuf * buf_stride + elem_off(ct, gate), 0, AUX_SC1);
          }
        }
      }
    }

    // ---- this step's inputs (HBM) go out BEFORE the wait for the other workgroups: their latency runs under the spin,
    // the gather and the MFMAs instead of in front of the gate math
    float in_dh[NCT][OWN_R], in_r[NCT][OWN_R], in_z[NCT][OWN_R], in_n[NCT][OWN_R], in_g[NCT][OWN_R], in_hp[NCT][OWN_R];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int sl = sidx[ct] < na ? sidx[ct] : 0;             // inactive lanes read row 0 of the step (exists) and ignore it
      const size_t eo = (size_t)(row_t + sl) * HID + ucol;
      const size_t ep = (size_t)(row_tm1 + sl) * HID + ucol;   // t == 0: row 0 of step 0, ignored
      if constexpr (OWN_R == 2) {
        // a lane's two units are neighbours (ucol is even): ONE 8-byte load per array instead of two scalar ones - six vector-memory
        // instructions per tile and step instead of twelve in front of the gather (round 6)
        const float2 v0 = *(const float2*)(a.dHout + eo), v1 = *(const float2*)(a.R + eo), v2 = *(const float2*)(a.Z + eo);
        const float2 v3 = *(const float2*)(a.N + eo), v4 = *(const float2*)(a.GHN + eo), v5 = *(const float2*)(a.Hraw + ep);
        in_dh[ct][0] = v0.x; in_dh[ct][1] = v0.y; in_r[ct][0] = v1.x; in_r[ct][1] = v1.y; in_z[ct][0] = v2.x; in_z[ct][1] = v2.y;
        in_n[ct][0] = v3.x; in_n[ct][1] = v3.y; in_g[ct][0] = v4.x; in_g[ct][1] = v4.y; in_hp[ct][0] = v5.x; in_hp[ct][1] = v5.y;
      } else {
#pragma unroll
        for (int e = 0; e < OWN_R; ++e) {
          in_dh[ct][e] = a.dHout[eo + e]; in_r[ct][e] = a.R[eo + e]; in_z[ct][e] = a.Z[eo + e];
          in_n[ct][e] = a.N[eo + e]; in_g[ct][e] = a.GHN[eo + e]; in_hp[ct][e] = a.Hraw[ep + e];
        }
      }
    }
    float dpr[NCT][OWN_R], dpz[NCT][OWN_R], dpn[NCT][OWN_R], dpnr[NCT][OWN_R];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      if (tfirst[ct] >= na) continue;
      // ---- (1) dhpart = dGH_{t+1} . W_hh for my units: gather + MFMA + cross-wave reduction -------------------------------
      float dhp[OWN_R];
#pragma unroll
      for (int e = 0; e < OWN_R; ++e) dhp[e] = 0.f;
      if (tfirst[ct] < na_next) {
        f32x4 acc[UT];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        // all of the wave's fragments in flight first (one L2 round trip, not NKS of them), then the products
        u32x4 hb[NKS];
        // a lane's 16 bytes belong to ONE clip column (lane & 15): columns that were not alive at step t+1 published nothing
        // and are not fetched (an offset past num_records returns zeros without a memory access)
        const int lane_off = sidx[ct] < na_next ? lane * 16 : 0x7FFF0000;
        // the data is its own flag: (re)load ALL of the wave's fragments until no live word is all ones (a round = one L2 round trip
        // with every fragment in flight; no per-fragment bookkeeping - the forward kernel's lesson)
        const bool col_live = sidx[ct] < na_next;
        unsigned spins = 0;
        for (;;) {
          if (local) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
              hb[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, rbuf * buf_stride + ((q * NKS + ks) * BPTT_MAX_TILES + ct) * 1024 + lane_off, 0, AUX_NT);
          } else {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
              hb[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, rbuf * buf_stride + ((q * NKS + ks) * BPTT_MAX_TILES + ct) * 1024 + lane_off, 0, AUX_SC1);
          }
          unsigned mx = 0u;
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            const unsigned m01 = hb[ks][0] > hb[ks][1] ? hb[ks][0] : hb[ks][1];
            const unsigned m23 = hb[ks][2] > hb[ks][3] ? hb[ks][2] : hb[ks][3];
            const unsigned m = m01 > m23 ? m01 : m23;
            mx = mx > m ? mx : m;
          }
          if (__all(!col_live || mx != 0xFFFFFFFFu)) break;
          if (++spins > BPTT_SPIN_LIMIT / 64) { if (lane == 0) __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          if ((spins & 63u) == 0u && __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const u32x4 v = hb[ks];
          if constexpr (BF) {
            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, v);
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
              acc[ut] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[ut][ks], bfrag, ks == 0 ? zero4 : acc[ut], 0, 0, 0);
          } else {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const float bj = __uint_as_float(v[jj]);
#pragma unroll
              for (int ut = 0; ut < UT; ++ut)
                acc[ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ut][ks][jj], bj, (ks == 0 && jj == 0) ? zero4 : acc[ut], 0, 0, 0);
            }
          }
        }
        f32x4* redw = red + ((t + ct) & 1) * RED_STRIDE;
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) redw[(q * UT + ut) * 64 + lane] = acc[ut];
        __syncthreads();
        if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;      // a gather gave up (here or elsewhere)
#pragma unroll
        for (int e = 0; e < OWN_R; ++e) {
          float p[4];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) p[qq] = *((const float*)&redw[(qq * UT + own_ut) * 64 + lane] + own_r0 + e);
          dhp[e] = (p[0] + p[1]) + (p[2] + p[3]);
        }
      }
      // ---- (2) gate gradients of my (unit, clip) pairs ----------------------------------------------------------------------
      if (sidx[ct] < na) {
        const bool cont = sidx[ct] < na_next;
#pragma unroll
        for (int e = 0; e < OWN_R; ++e) {
          float dh = in_dh[ct][e];
          if (cont) dh += carry[ct][e] + dhp[e];
          const float r = in_r[ct][e], z = in_z[ct][e], n = in_n[ct][e], ghn = in_g[ct][e];
          const float hprev = t > 0 ? in_hp[ct][e] : 0.f;
          const float dn = dh * (1.f - z);
          const float dz = dh * (hprev - n);
          dpn[ct][e] = dn * (1.f - n * n);
          dpr[ct][e] = dpn[ct][e] * ghn * r * (1.f - r);
          dpz[ct][e] = dz * z * (1.f - z);
          dpnr[ct][e] = dpn[ct][e] * r;
          carry[ct][e] = dh * z;
        }
        // ---- (3) publish dGH_t of this tile for step t-1: element (clip l15, k = gate*H + unit) ----------------------------
        if (t > 0) {
#pragma unroll
          for (int gate = 0; gate < 3; ++gate) {
            const int off = pbuf * buf_stride + elem_off(ct, gate);
            const float v0 = gate == 0 ? dpr[ct][0] : (gate == 1 ? dpz[ct][0] : dpnr[ct][0]);
            unsigned pv;
            if constexpr (BF) {
              const float v1 = gate == 0 ? dpr[ct][OWN_R - 1] : (gate == 1 ? dpz[ct][OWN_R - 1] : dpnr[ct][OWN_R - 1]);
              pv = pack_bf16x2(v0, v1);
            } else {
              pv = __float_as_uint(v0);
            }
            if (pv == 0xFFFFFFFFu) pv = 0xFFFFFFFEu;                  // all ones means "not yet published": such a NaN travels with its lowest bit cleared
            if (local) __builtin_amdgcn_raw_buffer_store_b32(pv, rs, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(pv, rs, off, 0, AUX_SC1);
          }
        }
      }
    }
    // ---- the step's results for the weight-gradient GEMMs, fire and forget (behind the publish, not in front of it) -------------
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      if (tfirst[ct] < na && sidx[ct] < na) {
        const size_t go = (size_t)(row_t + sidx[ct]) * K3 + ucol;
        // a.dGI / a.dGH (fp32 copies) are NULL when nobody reads them: a bf16 handle's weight gradients, bias sums and dgrad all take
        // the operand copies (gemm_tn.hip).  With the pair stores below a step issues 6 result stores instead of 24 (round 6)
        if (a.dGI != nullptr) {
#pragma unroll
          for (int e = 0; e < OWN_R; ++e) {
            a.dGI[go + e] = dpr[ct][e]; a.dGI[go + HID + e] = dpz[ct][e]; a.dGI[go + 2 * HID + e] = dpn[ct][e];
            a.dGH[go + e] = dpr[ct][e]; a.dGH[go + HID + e] = dpz[ct][e]; a.dGH[go + 2 * HID + e] = dpnr[ct][e];
          }
        }
        if constexpr (BF && OWN_R == 2) {
          bf16_t* gi = (bf16_t*)a.dGIop + go;
          bf16_t* gh = (bf16_t*)a.dGHop + go;
          const unsigned pr = pack_bf16x2(dpr[ct][0], dpr[ct][1]), pz = pack_bf16x2(dpz[ct][0], dpz[ct][1]);
          *(unsigned*)gi = pr; *(unsigned*)(gi + HID) = pz; *(unsigned*)(gi + 2 * HID) = pack_bf16x2(dpn[ct][0], dpn[ct][1]);
          *(unsigned*)gh = pr; *(unsigned*)(gh + HID) = pz; *(unsigned*)(gh + 2 * HID) = pack_bf16x2(dpnr[ct][0], dpnr[ct][1]);
        } else {
#pragma unroll
          for (int e = 0; e < OWN_R; ++e) {
            if constexpr (BF) {
              ((bf16_t*)a.dGIop)[go + e] = f2bf(dpr[ct][e]); ((bf16_t*)a.dGIop)[go + HID + e] = f2bf(dpz[ct][e]); ((bf16_t*)a.dGIop)[go + 2 * HID + e] = f2bf(dpn[ct][e]);
              ((bf16_t*)a.dGHop)[go + e] = f2bf(dpr[ct][e]); ((bf16_t*)a.dGHop)[go + HID + e] = f2bf(dpz[ct][e]); ((bf16_t*)a.dGHop)[go + 2 * HID + e] = f2bf(dpnr[ct][e]);
            } else {
              ((float*)a.dGIop)[go + e] = dpr[ct][e]; ((float*)a.dGIop)[go + HID + e] = dpz[ct][e]; ((float*)a.dGIop)[go + 2 * HID + e] = dpn[ct][e];
              ((float*)a.dGHop)[go + e] = dpr[ct][e]; ((float*)a.dGHop)[go + HID + e] = dpz[ct][e]; ((float*)a.dGHop)[go + 2 * HID + e] = dpnr[ct][e];
            }
          }
        }
      }
    }
  }
}

size_t gru_bptt_hx_bytes(bool bf16, int hid, int G) { return (size_t)BPTT_BUFS * G * (3 * hid / (bf16 ? 32 : 16)) * BPTT_MAX_TILES * 1024; }

// returns 0 on success, -1 for an unsupported shape (the caller falls back to the step-by-step loop).  Hidden sizes 1024 and 512, bf16
// (32 units per workgroup) or fp32 (16) operands.  hidden_dim 2048 trains through the step-by-step loop: its workgroups would own 16
// units (3 x 2048 / 4 columns of W_hh^T per wave = 192 registers), i.e. ONE bf16 element per lane - the exchange below publishes, resets
// and validates 32-bit words (a lane's bf16 PAIR), a word shared by two waves would look published when half of it is
template <int HID>
static int launch_gru_bptt_hid(bool bf16, int nct, const BpttArgs& a, hipStream_t s) {
  constexpr int UTB = 2;
  const int P = HID / (bf16 ? 16 * UTB : 16);
  const int grid = a.G * P;
#define LAUNCHB(WT, UT, NCT)                                                                       \
  do {                                                                                             \
    const size_t lds = (size_t)2 * 4 * UT * 64 * 16;                                               \
    gru_bptt_kernel<WT, HID, UT, NCT><<<grid, 256, lds, s>>>(a);                                   \
  } while (0)
  if (bf16) {
    if (nct == 1) LAUNCHB(bf16_t, UTB, 1);
    else if (nct == 2) LAUNCHB(bf16_t, UTB, 2);
    else LAUNCHB(bf16_t, UTB, 4);
  } else {
    if (nct == 1) LAUNCHB(float, 1, 1);
    else if (nct == 2) LAUNCHB(float, 1, 2);
    else LAUNCHB(float, 1, 4);
  }
#undef LAUNCHB
  return 0;
}

int launch_gru_bptt(bool bf16, int hid, int nct, BpttArgs a, hipStream_t s) {
  if ((hid != 1024 && hid != 512) || nct > BPTT_MAX_TILES) return -1;
  (void)hipMemsetAsync(a.sync, 0, 1024 * sizeof(unsigned), s);      // [0,64): placement words
  (void)hipMemsetAsync(a.hx, 0xFF, gru_bptt_hx_bytes(bf16, hid, a.G), s);      // every word "not yet published"
  return hid == 1024 ? launch_gru_bptt_hid<1024>(bf16, nct, a, s) : launch_gru_bptt_hid<512>(bf16, nct, a, s);
}
