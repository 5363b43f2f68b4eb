// The feed-forward half of a SPLIT PASS: pack -> layer1 GEMM -> LayerNorm + ReLU -> W_ih GEMM
// (model/rnn/rnn.py:38-43,53-61) as ONE persistent kernel that lives on the XCDs the recurrence does not hold, for the whole pass.
//
// Why one kernel.  A kernel's workgroups are dealt to all eight XCDs and the launch completes only when every one of them has run;
// the recurrence holds every CU of its XCDs with 312 registers per lane, and the 256 x 256 ping-pong GEMM (2 x 224 registers per
// SIMD) cannot even be dispatched beside it.  So feed-forward work can only run BESIDE a resident recurrence (instead of between
// its launches) if it never needs a workgroup on the recurrence's XCDs: a persistent launch whose workgroups read HW_REG_XCC_ID,
// leave at once below `xcd_lo`, and otherwise pull jobs from their XCD's queue until the pass is over.  The recurrence is the other
// persistent launch (gru_recurrence.hip, PASS instantiation) on XCDs 0 .. xcd_lo - 1; the two meet in memory:
//     gi_cnt[c]   += 1 per completed 256-row UNIT of chunk c        (this kernel -> recurrence: "the input projection of chunk c is there")
//     rec_cnt[c]  += 1 per recurrence wave that has consumed chunk c (recurrence -> this kernel: "the GI ring slot of chunk c is free")
//
// Units and rings.  The packed rows of a pass are cut into units of 256 rows (one GEMM tile row); unit u belongs to feed-forward
// XCD u mod nf for ALL of its jobs, so X, Y and E of a unit are produced and consumed through ONE XCD's L2 (which is coherent for
// its own CUs: plain stores, then `s_waitcnt vmcnt(0)` + a counter; the consumer invalidates its CU's vector L1 with `buffer_inv
// sc0` behind the counter and loads normally).  Only GI crosses to other XCDs: its stores are write-through (sc1), the recurrence
// reads it with sc1 loads.  X / Y / E live in rings of `ring_units` units (a multiple of nf: a slot is only ever touched by one XCD,
// so no XCD can write back a stale dirty line over another's data), GI in a ring of `gi_ring_units` (a power of two).
//
// Jobs of a unit: PACK (1), L1 tile (nt1 = E / 256), LN (1), WIH tile (nt2 = 3H / 256).  An XCD's queue is a ticket counter.  Its units
// are taken `sg` at a time: ticket k is job k mod T of SUPER-ROUND k div T (T = sg (2 + nt1 + nt2)), and super-round r holds PACK of its
// own sg units, the L1 tiles of the units of super-round r - lag1, LN of r - lag2 and the WIH tiles of r - lag3, so that by the time a job
// is claimed its producers (claimed a super-round = ~0.2 ms earlier) have normally finished: the dependency waits below are a safety
// net, not a pipeline stage.  Inside a GEMM section the tiles are ordered weight-slab-major over the sg units: the 32 workgroups of an
// XCD then work on sg row blocks x 8 slabs at a time, every slab feeding sg of them from the XCD's L2 (unit-major order made every
// running tile stream its own 2 MB slab over the fabric: 0.7 TB/s per XCD, and the pass ran at half the chunked GEMM rate).
// Tickets are claimed in order by resident workgroups that run every job to completion, so a wait can only be for a job that is
// already running.
// Both launches start with a bounded HANDSHAKE (kernels.h: PassHandshake): no job runs before the recurrence is known to be resident
// beside this kernel, so a pass that cannot run concurrently ends before it has written anything.  Every later wait is bounded too
// (2 s of s_memrealtime: a timeout raises the abort word that ends both kernels - no hung GPU), but behind a successful handshake every
// producer of every wait is resident and running.
//
// Arithmetic: the tile loop is the ping-pong K loop of gemm_pp.hip (same fragment order, same MFMA order), the row jobs are the
// bodies of pack_rows_kernel / ln_relu_rows_kernel (rowwise.hip) on a wave per row: every GI element is bit-identical to the
// chunked pass.
#include "common.h"
#include "kernels.h"
#include "pass_handshake.h"

#define FBK 64
#define FHALF 16384
#define FBUF 65536
#define FF_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define FF_TIMEOUT_TICKS 200000000ull          // s_memrealtime runs at 100 MHz: 2 s

__device__ __forceinline__ int ffp_key_b(int r) { return ((r >> 1) & 1) | (((r >> 3) & 3) << 1); }

// tid 0 only: wait until *p >= need.  false = aborted (by this wait's timeout or by someone else)
__device__ __forceinline__ bool ffp_wait_ge(const unsigned* p, unsigned need, unsigned* abort_word, unsigned code) {
  if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned spins = 0;
  for (;;) {
    __builtin_amdgcn_s_sleep(8);
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
    if ((++spins & 31u) == 0u) {
      if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
      if (__builtin_amdgcn_s_memrealtime() - t0 > FF_TIMEOUT_TICKS) {
        __hip_atomic_store(abort_word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
}

// One 256 x 256 output tile: C[rows <= 256, 256] (16-bit) = A[rows, K] . B[256, K]^T + bias, the 8-phase ping-pong loop of gemm_pp.hip
// (see there for the schedule).  A, B, bias, C point at the tile's first row / column.  SC1OUT: the stores are write-through (the tile
// is read on another XCD).
template <typename OT, bool SC1OUT>
__device__ __forceinline__ void ffp_tile(char* smem, const bf16_t* __restrict__ A, int lda, int rows, const bf16_t* __restrict__ B, int ldb,
                                         const float* __restrict__ bias, bf16_t* __restrict__ C, int ldc, int K) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));       // opaque per call: the lane-derived addresses below must not be hoisted out of the job loop (they would
                                      // be ~20 VGPRs alive across every job, spilled around the K loop)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int nk = K / FBK;
  const int sr = lane >> 3, scp = lane & 7;
  int a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + sr;
    a_off[i] = r * lda * 2 + ((scp ^ ((r >> 1) & 7)) << 4);
    b_off[i] = r * ldb * 2 + ((scp ^ ffp_key_b(r)) << 4);
  }
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, rows * lda * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 256 * ldb * 2, 0x00020000);
  auto stage_a = [&](int hf, int kt) {
    char* dst = smem + (kt & 1) * FBUF + hf * FHALF + wave * 2048;
    const int so = hf * 128 * lda * 2 + kt * (FBK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, a_off[i], so, 0, 0);
  };
  auto stage_b = [&](int hf, int kt) {
    char* dst = smem + (kt & 1) * FBUF + (2 + hf) * FHALF + wave * 2048;
    const int so = hf * 128 * ldb * 2 + kt * (FBK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, b_off[i], so, 0, 0);
  };
  const int fr = lane & 15, fq = lane >> 4;
  bf16x8 a0[2][4], a1[2][4], b0[2][2], b1[2][2];
  auto read_a = [&](bf16x8 (&af)[2][4], int slot, int kt) {
    const char* base = smem + (kt & 1) * FBUF + slot * FHALF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = grp * 64 + i * 16 + fr;
        af[ks][i] = *(const bf16x8*)(base + r * 128 + (((ks * 4 + fq) ^ ((r >> 1) & 7)) << 4));
      }
  };
  auto read_b = [&](bf16x8 (&bf)[2][2], int slot, int kt) {
    const char* base = smem + (kt & 1) * FBUF + slot * FHALF;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = wc * 32 + (fr >> 2) * 8 + j * 4 + (fr & 3);
        bf[ks][j] = *(const bf16x8*)(base + r * 128 + (((ks * 4 + fq) ^ ffp_key_b(r)) << 4));
      }
  };
  f32x4 acc[2][2][4][2];
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
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) c[i][j] = op16<OT>::mfma(bf[ks][j], af[ks][i], c[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  stage_a(0, 0); stage_b(0, 0); stage_b(1, 0); stage_a(1, 0);
  stage_a(0, 1); stage_b(0, 1); stage_b(1, 1);
  FF_WAIT(10);
  bar();
  read_a(a0, 0, 0);
  if (grp == 1) bar();
  for (int kt = 0; kt < nk; ++kt) {
    const bool full = kt + 2 < nk;
    read_b(b0, 2, kt);
    if (kt + 1 < nk) stage_a(1, kt + 1);
    if (full) FF_WAIT(10); else FF_WAIT(0);
    bar();
    mma(acc[0][0], a0, b0);
    bar();
    read_b(b1, 3, kt);
    if (full) { stage_a(0, kt + 2); FF_WAIT(10); } else FF_WAIT(0);
    bar();
    mma(acc[0][1], a0, b1);
    bar();
    read_a(a1, 1, kt);
    if (full) { stage_b(0, kt + 2); FF_WAIT(10); } else FF_WAIT(0);
    bar();
    mma(acc[1][1], a1, b1);
    bar();
    if (kt + 1 < nk) read_a(a0, 0, kt + 1);
    if (full) { stage_b(1, kt + 2); FF_WAIT(10); } else FF_WAIT(0);
    bar();
    mma(acc[1][0], a1, b0);
    bar();
  }
  if (grp == 0) bar();
  float4 bv[2][2];
#pragma unroll
  for (int y = 0; y < 2; ++y)
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[y][j] = *(const float4*)(bias + y * 128 + wc * 32 + fq * 8 + j * 4);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, rows * ldc * 2, 0x00020000);
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int n = y * 128 + wc * 32 + fq * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = x * 128 + grp * 64 + i * 16 + fr;
        const f32x4 v0 = acc[x][y][i][0], v1 = acc[x][y][i][1];
        u32x4 pk;
        pk[0] = op16<OT>::pack2_sat(v0[0] + bv[y][0].x, v0[1] + bv[y][0].y); pk[1] = op16<OT>::pack2_sat(v0[2] + bv[y][0].z, v0[3] + bv[y][0].w);
        pk[2] = op16<OT>::pack2_sat(v1[0] + bv[y][1].x, v1[1] + bv[y][1].y); pk[3] = op16<OT>::pack2_sat(v1[2] + bv[y][1].z, v1[3] + bv[y][1].w);
        // rows past `rows` fall outside the resource's num_records: the store is dropped
        if constexpr (SC1OUT) __builtin_amdgcn_raw_buffer_store_b128(pk, rs_c, (m * ldc + n) * 2, 0, AUX_SC1);
        else __builtin_amdgcn_raw_buffer_store_b128(pk, rs_c, (m * ldc + n) * 2, 0, 0);
      }
    }
}

// rows [row0, row0 + nrows) of the packed feature matrix (pack_rows_kernel's element arithmetic).  A wave owns 32 consecutive rows: lanes
// 0 .. 31 look their (clip, frame) up side by side (one chain of table reads per wave instead of one per row: a persistent workgroup has
// no other workgroups to hide that latency behind), then the wave streams its rows through two register buffers.
//
// Round 6: every access of the stream is a BUFFER access on a per-row resource.  The source rows come out of a pointer TABLE, so as plain
// pointers they are generic-address-space: hipcc loaded them with flat_load_dwordx4 behind a branch per load (a NULL flow row, a column
// beyond the row) and - flat accesses return in no order the counter can express - waited with vmcnt(0) + lgkmcnt(0) in front of every
// drain: burst, wait, burst, 202 us per 256-row job = 31 GB/s per CU (ISA; profiles/r06_pass_clock.log).  A buffer resource whose
// num_records is the row's byte length (0 for a missing row) returns zeros for what lies outside and drops stores outside, so the stream
// has no branch, the loads are counted in order, and a round of loads is in flight while the round before it is converted and stored.
// What that bought: 202 -> 190 us per job, no more - with two rounds in flight the job moves 6.3 MB at 33-37 GB/s, which is what ONE CU
// takes in from HBM (the stand-alone pack kernel at 5.3 TB/s is 21 GB/s per CU; the job alone on the chip, every other job skipped, takes
// 162 us: profiles/r06_rowjobs_alone.log).  Eight rows x one 512-column chunk per round instead of two rows x four (the rows' pages opened
// side by side) is 7 % slower (profiles/r06_pack_wide.log).  The job's cost is the CU it occupies while it waits for memory.
template <typename OT, bool IN16>
__device__ __forceinline__ void ffp_pack_rows(const FfPassArgs& a, int my_rows, int wave, int lane, const char* srgb, const char* sflow,
                                              bf16_t* __restrict__ Xs) {
  constexpr int ES = IN16 ? 2 : 4, NV = IN16 ? 1 : 2;
  constexpr int PR = 2;                                          // rows per round
  const int din = a.kx;
  // a round = PR rows x 2 048 columns of ONE source (four 512-column chunks, 8 columns per lane and chunk): rgb's column rounds, then flow's
  const int cr_rgb = (a.d_rgb + 2047) >> 11, cpr = cr_rgb + ((a.d_flow + 2047) >> 11);
  const int n_rounds = my_rows > 0 ? ((my_rows + PR - 1) / PR) * cpr : 0;
  auto readlane_ptr = [](const char* p, int l) -> const char* {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return (const char*)(((unsigned long long)hi << 32) | lo);
  };
  // lane offsets of the stream, loop-invariant and opaque (the compiler otherwise rebuilds them in registers a pending load still owns
  // and waits for that load): the column window moves the resource's BASE, not the offset
  int voff_in = lane * 8 * ES, voff_in2 = voff_in + 1024 * ES, voff_out = lane * 16;
  asm volatile("" : "+v"(voff_in), "+v"(voff_in2), "+v"(voff_out));
  auto issue = [&](u32x4 (&v)[PR][4][NV], int k) {
    const int rp = k / cpr, cj = k - rp * cpr;
    const bool fl = cj >= cr_rgb;
    const int cwin = (fl ? cj - cr_rgb : cj) << 11, seg_len = fl ? a.d_flow : a.d_rgb;
#pragma unroll
    for (int e = 0; e < PR; ++e) {
      const int r = rp * PR + e, rr = r < my_rows ? r : 0;
      const char* s0 = readlane_ptr(srgb, rr);
      const char* s1 = readlane_ptr(sflow, rr);
      const char* src = fl ? s1 : s0;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)cwin * ES), 0,
                                                                          (r < my_rows && src != nullptr) ? (seg_len - cwin) * ES : 0, 0x00020000);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int w = 0; w < NV; ++w)
          v[e][b][w] = __builtin_amdgcn_raw_buffer_load_b128(rs, (b < 2 ? voff_in : voff_in2) + ((b & 1) * 512 * ES + 16 * w), 0, AUX_NT);
    }
  };
  auto drain = [&](const u32x4 (&v)[PR][4][NV], int k) {
    const int rp = k / cpr, cj = k - rp * cpr;
    const bool fl = cj >= cr_rgb;
    const int cwin = (fl ? cj - cr_rgb : cj) << 11, seg_len = fl ? a.d_flow : a.d_rgb, seg_base = fl ? a.d_rgb : 0;
#pragma unroll
    for (int e = 0; e < PR; ++e) {
      const int r = rp * PR + e;
      bf16_t* dst = Xs + (size_t)(wave * 32 + (r < my_rows ? r : 0)) * din + seg_base + cwin;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, r < my_rows ? (seg_len - cwin) * 2 : 0, 0x00020000);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        u32x4 o = v[e][b][0];
        if constexpr (!IN16) {
          const u32x4 x = v[e][b][0], y = v[e][b][NV - 1];
          o[0] = op16<OT>::pack2_sat(__uint_as_float(x[0]), __uint_as_float(x[1])); o[1] = op16<OT>::pack2_sat(__uint_as_float(x[2]), __uint_as_float(x[3]));
          o[2] = op16<OT>::pack2_sat(__uint_as_float(y[0]), __uint_as_float(y[1])); o[3] = op16<OT>::pack2_sat(__uint_as_float(y[2]), __uint_as_float(y[3]));
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, rs, voff_out + b * 1024, 0, 0);
      }
    }
  };
  // steady state without a condition on the issues: the wait-count pass takes the FEWEST outstanding loads over the paths that meet at a
  // join, so a skipped issue on one path makes every wait of the other one wait for the round it should leave in flight
  u32x4 va[PR][4][NV], vb[PR][4][NV];
  if (n_rounds == 0) return;
  issue(va, 0);
  int k = 0;
  // (and with the phases fenced: left alone, the scheduler converts a round right behind its own loads - ISA)
#define PACK_FENCE() __builtin_amdgcn_sched_barrier(0)
  for (; k + 2 < n_rounds; k += 2) {
    issue(vb, k + 1); PACK_FENCE();
    drain(va, k); PACK_FENCE();
    issue(va, k + 2); PACK_FENCE();
    drain(vb, k + 1); PACK_FENCE();
  }
  if (k + 1 < n_rounds) {
    issue(vb, k + 1); PACK_FENCE();
    drain(va, k); PACK_FENCE();
    drain(vb, k + 1);
  } else {
    drain(va, k);
  }
#undef PACK_FENCE
}

template <typename OT>
__device__ __forceinline__ void ffp_pack(const FfPassArgs& a, int row0, int nrows, bf16_t* __restrict__ Xs) {
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));      // per call (see ffp_tile): nothing lane-derived is carried across the job loop
  const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int rl = wave * 32 + (lane & 31);                       // my row of the unit (lanes 32 .. 63 mirror 0 .. 31)
  int clip = 0, t = 0;
  if (rl < nrows) {
    const int row = row0 + rl;
    int lo = a.plan.blk_step[row >> 5];                         // step of packed row 32 (row / 32): zero or a few hops from there
    while (lo + 1 < a.plan.s_max && a.plan.rowoff[lo + 1] <= row) ++lo;
    const int slot = row - a.plan.rowoff[lo];
    int k = a.plan.seg_off[slot];
    const int kend = a.plan.seg_off[slot + 1];
    while (k + 1 < kend && a.plan.seg_start[k + 1] <= lo) ++k;
    clip = a.plan.seg_clip[k];
    t = lo - a.plan.seg_start[k];
    if (a.rowmap != nullptr && lane < 32) ((int2*)a.rowmap)[row] = make_int2(clip, t);
  }
  const char* srgb = nullptr; const char* sflow = nullptr;       // my row's source rows (byte pointers; nullptr = zeros)
  const int es = a.in16 ? 2 : 4;
  if (rl < nrows) {
    const float* rgb = a.rgb_ptrs ? a.rgb_ptrs[clip] : nullptr;
    const float* flow = a.flow_ptrs ? a.flow_ptrs[clip] : nullptr;
    if (rgb) srgb = (const char*)rgb + (size_t)t * a.d_rgb * es;
    if (flow) sflow = (const char*)flow + (size_t)t * a.d_flow * es;
  }
  const int my_rows = nrows - wave * 32 < 32 ? nrows - wave * 32 : 32;
  if (a.in16) ffp_pack_rows<OT, true>(a, my_rows, wave, lane, srgb, sflow, Xs);
  else ffp_pack_rows<OT, false>(a, my_rows, wave, lane, srgb, sflow, Xs);
}

constexpr int FLN_MAX_E = 2048;       // embedding_dim up to which ff_pass_kernel keeps LayerNorm's gamma / beta in LDS (16 KB beside the GEMM buffers)

// LayerNorm + ReLU of nrows 16-bit rows (ln_relu_rows_kernel's arithmetic: two-pass statistics, same summation order), a wave per row, NR
// rows of a wave in flight together and the NEXT NR rows requested before this batch is reduced (a lone row is three dependent round
// trips - load, two wave reductions - and a persistent workgroup has nobody to hide them behind: 111 us per 256-row job without the prefetch).
// Round 6: that prefetch never overlapped anything (ISA: the request sat behind a condition, so the wait-count pass waited with vmcnt(0)
// where the paths met, and the gamma / beta loads inside the store loop waited for it a second time): NV is a template parameter (no
// branch per load), gamma / beta come from an LDS copy the workgroup makes once per launch (`gb`: LDS reads are counted by lgkmcnt and
// wait for no row in flight; as registers they are 16 NV per lane and the job spills), the batches alternate between two register sets
// with unconditional requests in the steady state, and the phases are fenced against the scheduler.
template <typename OT, int NV, int NR>
__device__ __forceinline__ void ffp_ln(const FfPassArgs& a, int nrows, const bf16_t* __restrict__ Ys, bf16_t* __restrict__ Es, const float* gb) {
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));      // per call (see ffp_tile)
  const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int E = NV * 512;
  const float fE = (float)a.E;        // the divisor stays a run-time value
  if (wave >= nrows) return;
  auto request = [&](u32x4 (&dst)[NR][NV], int rb) {
#pragma unroll
    for (int e = 0; e < NR; ++e) {
      int r = rb + 8 * e;
      if (r >= nrows) r = wave;                                  // a missing row repeats an existing one (its result is not stored)
      const bf16_t* y = Ys + (size_t)r * E;
#pragma unroll
      for (int i = 0; i < NV; ++i) dst[e][i] = *(const u32x4*)(y + (i * 64 + lane) * 8);
    }
  };
  // the rows stay in their 16-bit form (raw) and are unpacked where they are used - three times: the fp32 copy of NR rows would be
  // 32 NR registers on top of the two raw batches
  auto unpack = [](const u32x4 w, float (&f)[8]) {
    f[0] = op16<OT>::lo(w[0]); f[1] = op16<OT>::hi(w[0]); f[2] = op16<OT>::lo(w[1]); f[3] = op16<OT>::hi(w[1]);
    f[4] = op16<OT>::lo(w[2]); f[5] = op16<OT>::hi(w[2]); f[6] = op16<OT>::lo(w[3]); f[7] = op16<OT>::hi(w[3]);
  };
  auto process = [&](const u32x4 (&raw)[NR][NV], int rb) {
    float s[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) {
      s[e] = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float f[8];
        unpack(raw[e][i], f);
        s[e] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
      }
    }
    // wave_sum's butterfly on the NR rows side by side (same pairs, same order: bit-identical; one cross-lane round trip per level
    // instead of NR)
    auto wave_sum_rows = [](float (&v)[NR]) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        float t[NR];
#pragma unroll
        for (int e = 0; e < NR; ++e) t[e] = __shfl_xor(v[e], o, 64);
#pragma unroll
        for (int e = 0; e < NR; ++e) v[e] += t[e];
      }
    };
    float mu[NR], rstd[NR];
    wave_sum_rows(s);
#pragma unroll
    for (int e = 0; e < NR; ++e) mu[e] = s[e] / fE;
#pragma unroll
    for (int e = 0; e < NR; ++e) {
      float qq = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float f[8];
        unpack(raw[e][i], f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          // d * d is rounded before it is added: ln_relu_rows_kernel's squares compile to packed multiplies (v_pk_mul_f32), which cannot
          // fuse with the add; under -ffp-contract=fast this loop would become v_fmac_f32 and about one row in 150 would differ from the
          // chunked pass in the last bit of an output (tests/test_gpu_split.py holds the two passes to bit-identity)
#pragma clang fp contract(off)
          const float d = f[k] - mu[e];
          const float dd = d * d;
          qq = qq + dd;
        }
      }
      rstd[e] = qq;
    }
    wave_sum_rows(rstd);
#pragma unroll
    for (int e = 0; e < NR; ++e) rstd[e] = 1.0f / sqrtf(rstd[e] / fE + a.ln_eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (i * 64 + lane) * 8;
      const float4 g0 = *(const float4*)(gb + c), g1 = *(const float4*)(gb + c + 4);
      const float4 b0 = *(const float4*)(gb + FLN_MAX_E + c), b1 = *(const float4*)(gb + FLN_MAX_E + c + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < NR; ++e) {
        float f[8], o[8];
        unpack(raw[e][i], f);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaxf((f[k] - mu[e]) * rstd[e] * gg[k] + bb[k], 0.f);
        uint4 w;
        w.x = op16<OT>::pack2_sat(o[0], o[1]); w.y = op16<OT>::pack2_sat(o[2], o[3]); w.z = op16<OT>::pack2_sat(o[4], o[5]); w.w = op16<OT>::pack2_sat(o[6], o[7]);
        if (rb + 8 * e < nrows) *(uint4*)(Es + (size_t)(rb + 8 * e) * E + c) = w;
      }
    }
  };
  constexpr int STEP = 8 * NR;
#define LN_FENCE() __builtin_amdgcn_sched_barrier(0)
  u32x4 ra[NR][NV], rbuf[NR][NV];
  int rb = wave;
  request(ra, rb); LN_FENCE();
  for (; rb + 2 * STEP < nrows; rb += 2 * STEP) {
    request(rbuf, rb + STEP); LN_FENCE();
    process(ra, rb); LN_FENCE();
    request(ra, rb + 2 * STEP); LN_FENCE();
    process(rbuf, rb + STEP); LN_FENCE();
  }
  if (rb + STEP < nrows) {
    request(rbuf, rb + STEP); LN_FENCE();
    process(ra, rb); LN_FENCE();
    process(rbuf, rb + STEP);
  } else {
    process(ra, rb);
  }
#undef LN_FENCE
}

// any E / 512 (runtime nv <= MAXV): the form every embedding_dim but 512 / 1 024 / 2 048 runs (loads behind a branch each, gamma / beta
// per batch: correct, and as slow as the comment above says)
template <typename OT, int MAXV, int NR>
__device__ __forceinline__ void ffp_ln_generic(const FfPassArgs& a, int nrows, const bf16_t* __restrict__ Ys, bf16_t* __restrict__ Es) {
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));      // per call (see ffp_tile)
  const int lane = tid_ & 63, wave = tid_ >> 6;
  const int E = a.E, nv = E / 512;
  u32x4 raw[NR][MAXV], nxt[NR][MAXV];
  auto request = [&](u32x4 (&dst)[NR][MAXV], int rb) {
#pragma unroll
    for (int e = 0; e < NR; ++e) {
      int r = rb + 8 * e;
      if (r >= nrows) r = wave < nrows ? wave : 0;              // a missing row repeats an existing one (its result is not stored)
      const bf16_t* y = Ys + (size_t)r * E;
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
        if (i < nv) dst[e][i] = *(const u32x4*)(y + (i * 64 + lane) * 8);
    }
  };
  if (wave < nrows) request(raw, wave);
  for (int rb = wave; rb < nrows; rb += 8 * NR) {
    if (rb + 8 * NR < nrows) request(nxt, rb + 8 * NR);
    // the rows stay in their 16-bit form (raw) and are unpacked where they are used - three times: the fp32 copy of NR rows would be
    // 32 NR registers on top of the two raw batches
    auto unpack = [](const u32x4 w, float (&f)[8]) {
      f[0] = op16<OT>::lo(w[0]); f[1] = op16<OT>::hi(w[0]); f[2] = op16<OT>::lo(w[1]); f[3] = op16<OT>::hi(w[1]);
      f[4] = op16<OT>::lo(w[2]); f[5] = op16<OT>::hi(w[2]); f[6] = op16<OT>::lo(w[3]); f[7] = op16<OT>::hi(w[3]);
    };
    float s[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) {
      s[e] = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
          float f[8];
          unpack(raw[e][i], f);
          s[e] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
        }
    }
    float mu[NR], rstd[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) mu[e] = wave_sum(s[e]) / (float)E;
#pragma unroll
    for (int e = 0; e < NR; ++e) {
      float qq = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
        if (i < nv) {
          float f[8];
          unpack(raw[e][i], f);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
#pragma clang fp contract(off)                                    // as in ffp_ln below
            const float d = f[k] - mu[e];
            const float dd = d * d;
            qq = qq + dd;
          }
        }
      rstd[e] = qq;
    }
#pragma unroll
    for (int e = 0; e < NR; ++e) rstd[e] = 1.0f / sqrtf(wave_sum(rstd[e]) / (float)E + a.ln_eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (i < nv) {
        const int c = (i * 64 + lane) * 8;
        const float4 g0 = *(const float4*)(a.ln_g + c), g1 = *(const float4*)(a.ln_g + c + 4);
        const float4 b0 = *(const float4*)(a.ln_b + c), b1 = *(const float4*)(a.ln_b + c + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < NR; ++e) {
          float f[8], o[8];
          unpack(raw[e][i], f);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = fmaxf((f[k] - mu[e]) * rstd[e] * gg[k] + bb[k], 0.f);
          uint4 w;
          w.x = op16<OT>::pack2_sat(o[0], o[1]); w.y = op16<OT>::pack2_sat(o[2], o[3]); w.z = op16<OT>::pack2_sat(o[4], o[5]); w.w = op16<OT>::pack2_sat(o[6], o[7]);
          if (rb + 8 * e < nrows) *(uint4*)(Es + (size_t)(rb + 8 * e) * E + c) = w;
        }
      }
    if (rb + 8 * NR < nrows) {
#pragma unroll
      for (int e = 0; e < NR; ++e)
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
          if (i < nv) raw[e][i] = nxt[e][i];
    }
  }
}

template <typename OT>
__device__ __forceinline__ void ffp_ln_any(const FfPassArgs& a, int nrows, const bf16_t* __restrict__ Ys, bf16_t* __restrict__ Es, const float* gb) {
  switch (a.E >> 9) {                                            // E / 512 (launch_ff_pass: E % 512 == 0, E <= 4096)
    case 4: ffp_ln<OT, 4, 2>(a, nrows, Ys, Es, gb); break;      // the shipped embedding_dim
    case 2: ffp_ln<OT, 2, 4>(a, nrows, Ys, Es, gb); break;
    case 1: ffp_ln<OT, 1, 4>(a, nrows, Ys, Es, gb); break;
    default: ffp_ln_generic<OT, 8, 2>(a, nrows, Ys, Es); break;
  }
}

template <typename OT>
__global__ __launch_bounds__(512, 2) void ff_pass_kernel(FfPassArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_job[2];
  const int tid = threadIdx.x;
  const int xcc = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);           // HW_REG_XCC_ID
  if (xcc < a.xcd_lo) return;                                                // the recurrence's XCDs
  // start handshake (kernels.h: PassHandshake): nothing is read or written before the recurrence's leader has seen both launches resident;
  // a bounded wait that runs out ends BOTH launches before either has touched memory (the host then runs the chunked pass)
  if (tid == 0) {
    const unsigned nth = __hip_atomic_fetch_add(a.hs.ff_here + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_job[1] = hs_wait(a.hs, a.hs.ticks_all) == PREGO_HS_GO ? 1 : 0;
    if (a.max_wg > 0 && (int)nth >= a.max_wg) s_job[1] = 0;        // debug library: this XCD runs with fewer workgroups (contention experiments)
  }
  __syncthreads();
  if (!__builtin_amdgcn_readfirstlane(s_job[1])) return;
  // LayerNorm's gamma / beta, once per launch (ffp_ln; embedding_dim <= FLN_MAX_E - larger ones take the generic form, which reads memory)
  __shared__ __attribute__((aligned(16))) float s_gb[2 * FLN_MAX_E];
  if (a.E <= FLN_MAX_E)
    for (int c = tid; c < a.E; c += 512) { s_gb[c] = a.ln_g[c]; s_gb[FLN_MAX_E + c] = a.ln_b[c]; }
  const int q = xcc - a.xcd_lo, nf = 8 - a.xcd_lo;
  const int n_q = a.n_units > q ? (a.n_units - q + nf - 1) / nf : 0;         // units of this XCD: q, q + nf, ...
  // ticket k = job (k mod T) of super-round (k div T), T = sg (2 + nt1 + nt2): the sg units of a super-round go through each job kind
  // TOGETHER, and the tiles of a GEMM section are ordered weight-slab-major: [slab 0 of units 0 .. sg - 1], [slab 1 of ...], so the
  // workgroups that run side by side share a weight slab (read once from the fabric, sg - 1 times from this XCD's L2) and each unit's A rows
  const int sg = a.sg, tpr = sg * (2 + a.nt1 + a.nt2);
  const int n_rounds = (n_q + sg - 1) / sg + a.lag3;
  // job-time sums (debug): kept in LDS, touched by thread 0 only - in registers they would be 22 VGPRs alive across the GEMM tiles
  __shared__ unsigned long long s_stat[11];                      // [0..7] sums, [8] last stamp, [9] first stamp, [10] first shader-clock stamp
  const bool stats = a.stats != nullptr && tid == 0;
  if (stats) {
    for (int e = 0; e < 8; ++e) s_stat[e] = 0;
    s_stat[9] = __builtin_amdgcn_s_memrealtime(); s_stat[8] = s_stat[9]; s_stat[10] = __builtin_amdgcn_s_memtime();
  }
#define FSTAT(i) do { if (stats) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); s_stat[i] += n_ - s_stat[8]; s_stat[8] = n_; } } while (0)
  for (;;) {
    __syncthreads();                                                         // s_job of the previous iteration has been read
    if (tid == 0) s_job[0] = (int)__hip_atomic_fetch_add(a.tick + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int k = __builtin_amdgcn_readfirstlane(s_job[0]);                  // wave-uniform: everything derived from it stays in SGPRs
    const int round = k / tpr, j = k - round * tpr;
    FSTAT(5);
    if (round >= n_rounds) break;
    int type, i, nb = 0;
    if (j < sg) { type = 0; i = round * sg + j; }
    else if (j < sg + sg * a.nt1) { const int jj = j - sg; type = 1; nb = jj / sg; i = (round - a.lag1) * sg + (jj - nb * sg); }
    else if (j < 2 * sg + sg * a.nt1) { type = 2; i = (round - a.lag2) * sg + (j - sg - sg * a.nt1); }
    else { const int jj = j - 2 * sg - sg * a.nt1; type = 3; nb = jj / sg; i = (round - a.lag3) * sg + (jj - nb * sg); }
    if (i < 0 || i >= n_q) continue;
    const int u = q + i * nf;
    if (tid == 0) {
      bool ok = true;
      const int up = u - a.ring_units;                                       // the unit that held this unit's ring slot before
      if (type == 0) {
        if (up >= 0) ok = ffp_wait_ge(a.l1_cnt + up, (unsigned)a.nt1, a.abort_word, 0x100u);
      } else if (type == 1) {
        ok = ffp_wait_ge(a.pack_done + u, 1u, a.abort_word, 0x101u);
        if (ok && up >= 0) ok = ffp_wait_ge(a.ln_done + up, 1u, a.abort_word, 0x102u);
      } else if (type == 2) {
        ok = ffp_wait_ge(a.l1_cnt + u, (unsigned)a.nt1, a.abort_word, 0x103u);
        if (ok && up >= 0) ok = ffp_wait_ge(a.wih_cnt + up, (unsigned)a.nt2, a.abort_word, 0x104u);
      } else {
        ok = ffp_wait_ge(a.ln_done + u, 1u, a.abort_word, 0x105u);
        const int ug = u - a.gi_ring_units;                                  // the recurrence must have consumed that unit's chunk
        if (ok && ug >= 0) ok = ffp_wait_ge(a.rec_cnt + (ug >> a.chunk_unit_shift), (unsigned)a.rec_expect, a.abort_word, 0x106u);
      }
      s_job[1] = ok ? 1 : 0;
    }
    __syncthreads();
    if (!__builtin_amdgcn_readfirstlane(s_job[1])) break;
    FSTAT(4);
    asm volatile("buffer_inv sc0" ::: "memory");                            // this CU's vector L1 may hold the slot's previous contents
    const int row0 = u * 256;
    const int nrows = a.total_rows - row0 < 256 ? a.total_rows - row0 : 256;
    const size_t slot = (size_t)(u % a.ring_units) * 256;
    if (type == 0) {
      if (!(a.dbg & 1)) ffp_pack<OT>(a, row0, nrows, a.X + slot * a.kx);
    } else if (type == 1) {
      if (!(a.dbg & 4))
      ffp_tile<OT, false>(smem, a.X + slot * a.kx, a.kx, nrows, a.w1 + (size_t)nb * 256 * a.ld_w1, a.ld_w1, a.b1 + nb * 256,
                          a.Y + slot * a.E + nb * 256, a.E, a.kx);
    } else if (type == 2) {
      if (a.dbg & 2) { }
      else ffp_ln_any<OT>(a, nrows, a.Y + slot * a.E, a.Eb + slot * a.E, s_gb);
    } else {
      const size_t gslot = (size_t)(u & (a.gi_ring_units - 1)) * 256;
      if (!(a.dbg & 8))
      ffp_tile<OT, true>(smem, a.Eb + slot * a.E, a.E, nrows, a.w_ih + (size_t)nb * 256 * a.E, a.E, a.bias2 + nb * 256,
                         a.GI + gslot * a.n3 + nb * 256, a.n3, a.E);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // my stores have reached the L2 (sc1: memory)
    __syncthreads();
    FSTAT(type);
    if (tid == 0) {
      if (type == 0) __hip_atomic_store(a.pack_done + u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (type == 1) __hip_atomic_fetch_add(a.l1_cnt + u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (type == 2) __hip_atomic_store(a.ln_done + u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else {
        const unsigned old = __hip_atomic_fetch_add(a.wih_cnt + u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == (unsigned)a.nt2)                                     // the unit's input projection is complete
          __hip_atomic_fetch_add(a.gi_cnt + (u >> a.chunk_unit_shift), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (stats) {                               // debug (PREGO_SPLIT_STATS=1): 10 ns ticks summed over the feed-forward workgroups
    s_stat[7] = __builtin_amdgcn_s_memrealtime() - s_stat[9];
    s_stat[6] = __builtin_amdgcn_s_memtime() - s_stat[10];     // shader-clock cycles over the same span: [6] / [7] x 100 MHz = this XCD's clock
    for (int e = 0; e < 8; ++e) atomicAdd(a.stats + e, s_stat[e]);
  }
#undef FSTAT
}

// 0 on success, -1 = shape not supported.  One launch per pass; `grid` workgroups (256: one per CU, an eighth lands on every XCD).
int launch_ff_pass(const FfPassArgs& a, hipStream_t s) {
  if (a.E % 512 || a.E > 4096 || a.kx % FBK || a.kx < 2 * FBK || a.n3 % 256 || a.xcd_lo < 1 || a.xcd_lo > 7) return -1;
  if (a.sg < 1 || a.ring_units % (8 - a.xcd_lo) || (a.gi_ring_units & (a.gi_ring_units - 1)) || a.nt1 != a.E / 256 || a.nt2 != a.n3 / 256) return -1;
  static DeviceOnce once;
  once.run([&] {
    (void)hipFuncSetAttribute((const void*)ff_pass_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * FBUF);
    (void)hipFuncSetAttribute((const void*)ff_pass_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * FBUF);
  });
  if (a.f16) ff_pass_kernel<f16_t><<<256, 512, 2 * FBUF, s>>>(a);
  else ff_pass_kernel<bf16_t><<<256, 512, 2 * FBUF, s>>>(a);
  return 0;
}
