// Persistent GRU recurrence on SPLIT fp16 operands ("fp16x2", common.h): the argmax-identical mode of nn.GRU(2048,1024,1)
// (model/rnn/rnn.py:38,61) at 3/16 of the matrix cost of the exact-fp32 kernel in gru_recurrence.hip.
//   gh = W_hh h + b_hh with W_hh = (W_hi + W_lo) / scale and h = h_hi + h_lo (fp16 pairs): per 32-wide k-step and gate
//   acc += W_lo.h_hi; acc += W_hi.h_lo; acc += W_hi.h_hi   (v_mfma_f32_16x16x32_f16, fp32 accumulate, fixed order)
// Same decomposition as the fp32-operand kernel: a workgroup owns 16 hidden units (all three gates), its 48 x 1024 slice of W_hi AND
// W_lo sits in VGPRs as MFMA A-fragments (wave q: the K-quarter, 2 x 96 = 192 registers), 64 workgroups form a group (G = 4), the
// 4 x 16 x NCT slots are dealt to the groups interleaved, state / gates / sums are fp32.  72 MFMAs of 16 cycles per clip tile and
// wave instead of 192 of 32.
//
// Hand-off.  h_{t-1} of a group is all-gathered through global memory as TWO fp16 planes (hi, lo) in MFMA B-fragment order;
// the data is the flag: |h_hi| < 2 and |h_lo| <= 2^-11, so bit 14 of every element is free and carries the epoch tag (as in
// gru_recurrence.hip).  A group of 64 workgroups spans two XCDs, whose L2s are not coherent with each other, so when the launch
// has verified that every XCD holds exactly 32 workgroups (group := XCD pair, member := half * 32 + ticket) every element is
// published twice: a plain store into region L (stays in the producer XCD's L2; read by the same XCD's consumers with
// L1-bypassing nt loads) and a write-through sc1 store into region R (read by the partner XCD with sc1 loads).  A consumer wave's
// K-quarter is produced on ONE XCD (units [0, 512) by members 0..31), so a wave reads either region L or region R, never both.
// On any other placement everything goes through region R with sc1 on both sides.  The tags decide validity either way:
// placement only changes speed.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

#define X2_SPIN_LIMIT (1u << 22)
#define X2_TILES 4                              // clip tiles per group the exchange buffer is laid out for (gru_max_tiles)
#ifndef X2_NSEG
#define X2_NSEG 2
#endif

template <int HID, int NCT>
__global__ __launch_bounds__(256, 1) void gru_recurrence_x2_kernel(GruArgs a, const float* __restrict__ inv_scale_p) {
  constexpr int UNITS = 16, KQ = HID / 4, NKS = KQ / 32, KF = 32, EPL = 8;
  constexpr unsigned TAGM = 0x40004000u;
  constexpr int PLANE_BYTES = (HID / KF) * X2_TILES * 1024;      // one plane (hi or lo) of one group: [k-step][tile][lane][16 B]
  constexpr int REGION_BYTES = 2 * PLANE_BYTES;                  // hi | lo
  constexpr int GROUP_BYTES = 2 * REGION_BYTES;                  // region L | region R
  constexpr int P = HID / UNITS;                                 // 64 workgroups per group

  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* red = (f32x4*)smem;                                     // [2 parities][4 waves][3 gates][64 lanes]
  constexpr int RED_STRIDE = 4 * 3 * 64;

  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- placement rendezvous: group := XCD pair when every XCD holds exactly 32 workgroups
  __shared__ int s_place[4];
  if (tid == 0) {
    int gg = blockIdx.x % a.G, ww = blockIdx.x / a.G, loc = 0, half = 0;
    if (a.sync != nullptr && a.G == 4 && gridDim.x == 4 * P) {
      const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;           // HW_REG_XCC_ID
      const unsigned ticket = __hip_atomic_fetch_add(a.sync + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(a.sync + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      bool ok = true;
      while (__hip_atomic_load(a.sync + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        if (++spins > X2_SPIN_LIMIT) { ok = false; break; }
        __builtin_amdgcn_s_sleep(4);
      }
      if (!ok) { __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); loc = -1; }
      else {
        loc = 1;
        for (int i = 0; i < 8; ++i)
          if (__hip_atomic_load(a.sync + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 32u) loc = 0;
        if (loc) { gg = xcc >> 1; half = xcc & 1; ww = half * 32 + (int)ticket; }
      }
    }
    s_place[0] = gg; s_place[1] = ww; s_place[2] = loc; s_place[3] = half;
  }
  __syncthreads();
  const int g = __builtin_amdgcn_readfirstlane(s_place[0]);
  const int w = __builtin_amdgcn_readfirstlane(s_place[1]);
  const int place = __builtin_amdgcn_readfirstlane(s_place[2]);
  const int my_half = __builtin_amdgcn_readfirstlane(s_place[3]);
  if (place < 0) return;                                          // rendezvous timed out (abort word set)
  const bool paired = place == 1;
  if (g >= a.n_clips) return;
  {
    typedef const __attribute__((address_space(4))) int* cint_p0;
    if (g >= ((cint_p0)a.nact)[a.t0]) return;                     // every slot of this group ended before this launch
  }
  const int l15 = lane & 15, l4 = lane >> 4;
  const float inv_scale = *inv_scale_p;

  // ---- resident weights: rows of a.whh are [HID hi | HID lo] fp16 of W_hh * scale
  bf16x8 wh[3][NKS], wl[3][NKS];
#pragma unroll
  for (int gate = 0; gate < 3; ++gate)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const size_t row = (size_t)gate * HID + w * UNITS + l15;
      const int k = q * KQ + ks * 32 + 8 * l4;
      wh[gate][ks] = *(const bf16x8*)((const bf16_t*)a.whh + row * (2 * HID) + k);
      wl[gate][ks] = *(const bf16x8*)((const bf16_t*)a.whh + row * (2 * HID) + HID + k);
    }

  // gate-phase ownership: register q of the 16x16 accumulator tile (unit l4 * 4 + q of the workgroup's 16, clip column l15)
  const int ucol = w * UNITS + l4 * 4 + q;
  float hreg[NCT];
  int sidx[NCT], tfirst[NCT];
  const float bhn = a.b_hn[ucol];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    sidx[ct] = ct * 16 * a.G + l15 * a.G + g;
    tfirst[ct] = ct * 16 * a.G + g;
    hreg[ct] = (sidx[ct] < a.n_clips) ? a.h_state[(size_t)sidx[ct] * HID + ucol] : 0.f;
  }
  int nstart[NCT], segp[NCT], sege[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    nstart[ct] = 0x7fffffff; segp[ct] = 0; sege[ct] = 0;
    if (a.seg_start != nullptr && sidx[ct] < a.n_clips) {
      int k = a.seg_off[sidx[ct]];
      sege[ct] = a.seg_off[sidx[ct] + 1];
      while (k < sege[ct] && a.seg_start[k] < a.t0) ++k;
      if (k < sege[ct] && a.seg_start[k] == a.t0 && a.t0 > 0) hreg[ct] = 0.f;      // a clip starts on this launch's first step
      while (k < sege[ct] && a.seg_start[k] <= a.t0) ++k;
      segp[ct] = k;
      nstart[ct] = k < sege[ct] ? a.seg_start[k] : 0x7fffffff;
    }
  }

  // exchange buffers: [2 buffers][G][GROUP_BYTES]; one descriptor per group
  const int buf_stride = a.G * GROUP_BYTES;
  char* hx_base = (char*)a.hx + (size_t)g * GROUP_BYTES;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)hx_base, 0, buf_stride + GROUP_BYTES, 0x00020000);
  // this wave's K-quarter is produced by members [16 q, 16 q + 16): XCD half q >> 1 of the pair
  const bool rd_local = paired && (q >> 1) == my_half;
  const int rd_region = rd_local ? 0 : REGION_BYTES;

  // publish this lane's element of tile ct (hi and lo planes; region R always, region L when paired); no waiting
  auto publish = [&](int ct, int buf, unsigned tag, bool zero) {
    const int off = buf * buf_stride + ((ucol / KF) * X2_TILES + ct) * 1024 + ((((ucol % KF) / EPL) << 4) + l15) * 16 + (ucol % EPL) * 2;
    float hv = zero ? 0.f : hreg[ct];
    hv = __builtin_amdgcn_fmed3f(hv, -1.9990234375f, 1.9990234375f);
    const _Float16 hi = (_Float16)hv;
    const _Float16 lo = (_Float16)(hv - (float)hi);
    const unsigned short t14 = tag ? 0x4000u : 0u;
    const unsigned short vh = (unsigned short)((__builtin_bit_cast(unsigned short, hi) & 0xBFFFu) | t14);
    const unsigned short vl = (unsigned short)((__builtin_bit_cast(unsigned short, lo) & 0xBFFFu) | t14);
    __builtin_amdgcn_raw_buffer_store_b16(vh, rs, off + REGION_BYTES, 0, AUX_SC1);
    __builtin_amdgcn_raw_buffer_store_b16(vl, rs, off + REGION_BYTES + PLANE_BYTES, 0, AUX_SC1);
    if (paired) {
      __builtin_amdgcn_raw_buffer_store_b16(vh, rs, off, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b16(vl, rs, off + PLANE_BYTES, 0, 0);
    }
  };
  // gi loads are unconditional (inactive lanes read row 0 of the step and ignore it), as in gru_recurrence.hip
  auto load_gi = [&](float (&dst)[3], int ct, int na, int rbase) {
    const int r = rbase + (sidx[ct] < na ? sidx[ct] : 0);
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) dst[gate] = ((const float*)a.gi)[(size_t)r * (3 * HID) + gate * HID + ucol];
  };

  typedef const __attribute__((address_space(4))) int* cint_p;
  cint_p nact_c = (cint_p)a.nact;
  cint_p rowoff_c = (cint_p)a.rowoff;
  const int nsteps = a.t1 - a.t0;
  int na_c = nact_c[a.t0], rb_c = rowoff_c[a.t0];
  int na_n = nsteps > 1 ? nact_c[a.t0 + 1] : 0, rb_n = nsteps > 1 ? rowoff_c[a.t0 + 1] : 0;
  float giA[NCT][3], giB[NCT][3];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
    if (tfirst[ct] < na_c) { publish(ct, 1, 1u, false); load_gi(giA[ct], ct, na_c, rb_c - a.row_base); }
  int parity = 0;

  auto step = [&](const int tl, float (&gir)[NCT][3], float (&gin)[NCT][3]) -> bool {
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

#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      if (tfirst[ct] < na) {
        // ---- (1) gather h_{t-1} of tile ct: 8 hi + 8 lo fragments of this wave's K-quarter, validated by their tags in NSEG segments
        constexpr int NSEG = X2_NSEG, SEGK = NKS / NSEG;
        f32x4 acc[3];
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        u32x4 hbh[NKS], hbl[NKS];
        unsigned spins = 0;
        const bool col_live = sidx[ct] < na;
        const int lane_off = col_live ? lane * 16 : 0x7F000000;     // dead columns: past num_records, zeros, no memory access
        const int base_off = rbuf * buf_stride + rd_region + ((q * NKS) * X2_TILES + ct) * 1024 + lane_off;
        if (rd_local) {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            hbh[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + ks * X2_TILES * 1024, 0, AUX_NT);
            hbl[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + ks * X2_TILES * 1024 + PLANE_BYTES, 0, AUX_NT);
          }
        } else {
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) {
            hbh[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + ks * X2_TILES * 1024, 0, AUX_SC1);
            hbl[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + ks * X2_TILES * 1024 + PLANE_BYTES, 0, AUX_SC1);
          }
        }
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg) {
          auto seg_stale = [&]() -> bool {
            unsigned badv = 0u;
#pragma unroll
            for (int ks = sg * SEGK; ks < (sg + 1) * SEGK; ++ks) {
              // the xor that tests the tags also removes them (gru_recurrence.hip, round 6): the fragments go to the MFMAs as they are
              hbh[ks][0] ^= eword; hbh[ks][1] ^= eword; hbh[ks][2] ^= eword; hbh[ks][3] ^= eword;
              hbl[ks][0] ^= eword; hbl[ks][1] ^= eword; hbl[ks][2] ^= eword; hbl[ks][3] ^= eword;
              badv |= ((hbh[ks][0] | hbh[ks][1]) | (hbh[ks][2] | hbh[ks][3])) | ((hbl[ks][0] | hbl[ks][1]) | (hbl[ks][2] | hbl[ks][3]));
            }
            return !__all(!col_live || (badv & TAGM) == 0u);
          };
          if (seg_stale()) {
            for (;;) {
              if (++spins > X2_SPIN_LIMIT) {
                if (lane == 0) __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
              }
              if ((spins & 255u) == 0u) {
                if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
              }
              if (rd_local) {
#pragma unroll
                for (int ks = sg * SEGK; ks < NKS; ++ks) {
                  hbh[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + ks * X2_TILES * 1024, 0, AUX_NT);
                  hbl[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + ks * X2_TILES * 1024 + PLANE_BYTES, 0, AUX_NT);
                }
              } else {
#pragma unroll
                for (int ks = sg * SEGK; ks < NKS; ++ks) {
                  hbh[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + ks * X2_TILES * 1024, 0, AUX_SC1);
                  hbl[ks] = __builtin_amdgcn_raw_buffer_load_b128(rs, base_off + ks * X2_TILES * 1024 + PLANE_BYTES, 0, AUX_SC1);
                }
              }
              if (!seg_stale()) break;
            }
          }
#pragma unroll
          for (int ks = sg * SEGK; ks < (sg + 1) * SEGK; ++ks) {
            const u32x4 vh = hbh[ks], vl = hbl[ks];
            const bf16x8 fh = __builtin_bit_cast(bf16x8, vh), fl = __builtin_bit_cast(bf16x8, vl);
#pragma unroll
            for (int gate = 0; gate < 3; ++gate) {
              acc[gate] = op16<f16_t>::mfma(wl[gate][ks], fh, ks == 0 ? zero4 : acc[gate]);   // small terms first
              acc[gate] = op16<f16_t>::mfma(wh[gate][ks], fl, acc[gate]);
              acc[gate] = op16<f16_t>::mfma(wh[gate][ks], fh, acc[gate]);
            }
          }
        }
        // the gather's waits have retired this step's gi loads too: pin that for the compiler, then prefetch gi(t + 1)
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) asm volatile("" : "+v"(gir[ct][gate]));
        if (more && tfirst[ct] < na_n) load_gi(gin[ct], ct, na_n, rb_n - a.row_base);

        // ---- (2) K-quarter reduction through LDS (double buffered: one barrier per tile)
        f32x4* redw = red + parity * RED_STRIDE;
        parity ^= 1;
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) redw[(q * 3 + gate) * 64 + lane] = acc[gate];
        __syncthreads();

        // ---- (3) gates + state update of this lane's (unit, clip)
        {
          float part[3][4];
#pragma unroll
          for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) part[gate][qq] = *((const float*)&redw[(qq * 3 + gate) * 64 + lane] + q);
          float gh[3];
#pragma unroll
          for (int gate = 0; gate < 3; ++gate) gh[gate] = ((part[gate][0] + part[gate][1]) + (part[gate][2] + part[gate][3])) * inv_scale;
          if (col_live) {
            const float r = sigmoidf_(gir[ct][0] + gh[0]);
            const float z = sigmoidf_(gir[ct][1] + gh[1]);
            const float n = tanhf_(gir[ct][2] + r * (gh[2] + bhn));
            hreg[ct] = (1.0f - z) * n + z * hreg[ct];
          }
        }
        // ---- (4) publish h_t for step t + 1
        const bool restart = (t + 1 == nstart[ct]);
        if (more && sidx[ct] < na_n) publish(ct, tl & 1, (unsigned)((tl >> 1) & 1), restart);
        // ---- (5) relu(h_t): fp32 rows for the fp32 classifier kernel
        if (col_live && a.h_relu_out) ((float*)a.h_relu_out)[(size_t)(rbase + sidx[ct]) * HID + ucol] = fmaxf(hreg[ct], 0.f);
        if (restart) {
          hreg[ct] = 0.f;
          ++segp[ct];
          nstart[ct] = segp[ct] < sege[ct] ? a.seg_start[segp[ct]] : 0x7fffffff;
          asm volatile("" : "+v"(nstart[ct]));
        }
      }
    }
    na_c = na_n; rb_c = rb_n; na_n = na_2; rb_n = rb_2;
    return true;
  };
  for (int tl = 0; tl < nsteps; tl += 2) {
    if (tfirst[0] >= na_c) break;
    if (!step(tl, giA, giB)) return;
    if (tl + 1 < nsteps && !step(tl + 1, giB, giA)) return;
  }
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
    if (sidx[ct] < a.n_clips) a.h_state[(size_t)sidx[ct] * HID + ucol] = hreg[ct];
}

size_t gru_x2_hx_bytes(int hid, int G) { return (size_t)2 * G * 4 * (hid / 32) * X2_TILES * 1024; }

// what a launch needs re-armed: buffer 0 := tag 1 everywhere, buffer 1 := 0, sync[0..15] := 0 (kernels.h: GruArm)
GruArm gru_x2_arm_desc(int hid, int G, void* hx, unsigned* sync) {
  return GruArm{(unsigned*)hx, (unsigned long long)(gru_x2_hx_bytes(hid, G) / 2 / 4), 0x40004000u, sync};
}

__global__ void gru_x2_arm_kernel(unsigned* __restrict__ hx, size_t words_per_buf, unsigned* __restrict__ sync) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = i; k < words_per_buf / 4; k += stride) {
    ((uint4*)hx)[k] = make_uint4(0x40004000u, 0x40004000u, 0x40004000u, 0x40004000u);
    ((uint4*)(hx + words_per_buf))[k] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (sync != nullptr && i < 16) sync[i] = 0u;
}

// 0 on success, -1 for an unsupported (hid, nct)
int launch_gru_recurrence_x2(int hid, int nct, GruArgs a, const float* inv_scale, hipStream_t s) {
  if (hid != 1024 || a.G < 1 || a.G > 4 || !inv_scale) return -1;
  if (!a.armed) gru_x2_arm_kernel<<<256, 256, 0, s>>>((unsigned*)a.hx, gru_x2_hx_bytes(hid, a.G) / 2 / 4, a.sync);
  const int grid = a.G * 64;
  const size_t lds = (size_t)2 * 4 * 3 * 64 * 16;
  if (nct == 1) gru_recurrence_x2_kernel<1024, 1><<<grid, 256, lds, s>>>(a, inv_scale);
  else if (nct == 2) gru_recurrence_x2_kernel<1024, 2><<<grid, 256, lds, s>>>(a, inv_scale);
  else return -1;                 // more tiles would spill (2 x 96 weight registers): the planner keeps fp16x2 handles at <= 2 tiles
  return 0;
}
