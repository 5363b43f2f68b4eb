// Per-frame average precision on the device: step_recognition/utils/metrics.py:25-62 calls
// sklearn.metrics.average_precision_score once per class on the [frames x classes] score matrix of an eval pass (main.py:101 runs
// that after every epoch).  The definition (sklearn: precision_recall_curve + step-wise sum): a class's thresholds are its DISTINCT
// score values in descending order (ties share one threshold); with tps_k / cnt_k = positives / samples at or above threshold k and
// P positives in all,   AP = sum_k (tps_k - tps_{k-1}) / P * tps_k / cnt_k.
//
// Only thresholds that hold a positive contribute, so only the POSITIVES need ranks (round 5; rounds 2-4 sorted every one of the
// frames x classes pairs: four radix passes of scattered 8-byte stores, 18-25 ms for the eval set - the tail of `Evaluate`):
//   ap_extract   one pass over scores / targets [frames][classes] (row-major, as the head kernel writes them): every score becomes a
//                32-bit order-reversing key (ascending key = descending score), written class-major (column c = n keys in a row);
//                the keys of a class's positives are appended to that class's list (order irrelevant: the list is sorted next)
//   ap32_hist / ap32_scan / ap32_scatter   four stable LSD radix passes over 8-bit digits, one segment per class of ITS OWN length
//                P_c (read from the device: nothing returns to the host) -> q_c[0 .. P_c) ascending.  One wave per 4096-key tile.
//   ap_count     for every key k of column c: b = lower_bound(q_c, k) = the first positive at or below this score; cnt_c[b] += 1
//                unless b = P_c (a score below every positive is at or above no threshold).  A workgroup owns a range of q_c and
//                its counters in LDS and streams a slice of the column (comment at the kernel; multi-label targets with more
//                positives than 16 workgroups hold fall back to a sampled table + one device atomic per key).
//                samples at or above positive i's score = cnt_c[0] + ... + cnt_c[i]   (lower_bound never returns the inside of a
//                tie run, so the inclusive prefix is the same for every member of the run)
//   ap_reduce    one workgroup per class walks q_c and cnt_c once: running prefix of cnt, run ends of q, fp64 accumulation of
//                (positives in the run) * (positives so far) / (samples so far).
// Exact integer ranks; the only floating-point work is the final fp64 sum (differs from sklearn's by summation order, ~1e-16) and
// the per-class score mass (fixed summation order: the same bits on every run).
#include "common.h"
#include "kernels.h"

#define AP_TILE 4096          // keys per radix tile (one wave, 64 steps of 64)
#define AP_RADIX 256
#define AP_PASSES 4           // 32-bit keys, 8-bit digits
#define AP_TGRID 128          // radix tiles of a class are walked by at most this many workgroups
#define AP_XF 128             // ap_extract: frames per tile
#define AP_XC 96              // ap_extract: classes per tile (column blocks of wider matrices)
#define AP_TAB 16384          // ap_count: LDS words (64 KB: two 1024-thread workgroups per CU)
#define AP_OWN (AP_TAB / 2)   // ap_count: positives a workgroup owns (their keys + their counters in LDS)
#define AP_SPLITS_MAX 64      // ap_count: workgroups per class

// order-reversing key of a float score: larger score -> smaller key; -0.0 == +0.0 (sklearn compares values, not bits)
__device__ __forceinline__ unsigned ap_desc_key(float s) {
  if (s == 0.f) s = 0.f;
  const unsigned b = __float_as_uint(s);
  const unsigned asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);     // ascending-order key of an IEEE float
  return ~asc;
}
__device__ __forceinline__ float ap_key_score(unsigned k) {
  const unsigned asc = ~k;
  const unsigned b = (asc & 0x80000000u) ? (asc & 0x7FFFFFFFu) : ~asc;
  return __uint_as_float(b);
}


// [AP_XF frames] x [<= AP_XC classes] tile: keys written class-major through an LDS transpose, positives appended per class
// LABELS: the positives are given as one class id per frame (one-hot targets the feeder reduced on the host: 4 bytes per frame over
// the link instead of 4 x classes); a frame whose id is outside [0, C) has no positive
template <bool LABELS>
__global__ __launch_bounds__(256) void ap_extract_kernel(const float* __restrict__ scores, const float* __restrict__ target,
                                                         const int* __restrict__ labels, long long n, int C, unsigned* __restrict__ keys,
                                                         unsigned* __restrict__ pos, unsigned* __restrict__ cursor) {
  __shared__ unsigned tile[AP_XC][AP_XF + 1];
  __shared__ unsigned lcnt[AP_XC], lbase[AP_XC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long f0 = (long long)blockIdx.x * AP_XF;
  const int c0 = blockIdx.y * AP_XC;
  const int cb = min(AP_XC, C - c0);
  const int nf = (int)min((long long)AP_XF, n - f0);
  if (tid < AP_XC) lcnt[tid] = 0u;
  __syncthreads();
  const int total = nf * cb;
  for (int idx = tid; idx < total; idx += 256) {
    const int f = idx / cb, c = idx - f * cb;
    const size_t g = (size_t)(f0 + f) * C + c0 + c;
    tile[c][f] = ap_desc_key(scores[g]);
    const bool positive = LABELS ? labels[f0 + f] == c0 + c : target[g] != 0.f;
    if (positive) atomicAdd(&lcnt[c], 1u);
  }
  __syncthreads();
  if (tid < cb) {
    const unsigned m = lcnt[tid];
    lbase[tid] = m ? atomicAdd(&cursor[c0 + tid], m) : 0u;
    lcnt[tid] = 0u;
  }
  __syncthreads();
  for (int idx = tid; idx < total; idx += 256) {
    const int f = idx / cb, c = idx - f * cb;
    const bool positive = LABELS ? labels[f0 + f] == c0 + c : target[(size_t)(f0 + f) * C + c0 + c] != 0.f;
    if (positive)
      pos[(size_t)(c0 + c) * n + lbase[c] + atomicAdd(&lcnt[c], 1u)] = tile[c][f];
  }
  for (int c = wave; c < cb; c += 4) {
    unsigned* col = keys + (size_t)(c0 + c) * n + f0;
    for (int f = lane; f < nf; f += 64) col[f] = tile[c][f];
  }
}

// digit histogram of a class's tiles: hist[c][digit][tile] with the class's own tile count as the stride; the first pass also
// clears the class's counters for ap_count
__global__ __launch_bounds__(64) void ap32_hist_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ len, long long n,
                                                       int ntiles_max, int shift, unsigned* __restrict__ hist, unsigned* __restrict__ zero) {
  __shared__ unsigned h[AP_RADIX];
  const int c = blockIdx.y, lane = threadIdx.x;
  const unsigned P = len[c];
  const int ntc = (int)((P + AP_TILE - 1) / AP_TILE);
  const unsigned* col = keys + (size_t)c * n;
  unsigned* hc = hist + (size_t)c * AP_RADIX * ntiles_max;
  for (int tile = blockIdx.x; tile < ntc; tile += gridDim.x) {
    for (int d = lane; d < AP_RADIX; d += 64) h[d] = 0u;
    __syncthreads();
    const unsigned i0 = (unsigned)tile * AP_TILE;
    for (int j = lane; j < AP_TILE; j += 64) {
      const unsigned i = i0 + j;
      if (i < P) {
        atomicAdd(&h[(col[i] >> shift) & (AP_RADIX - 1)], 1u);
        if (zero) zero[(size_t)c * n + i] = 0u;
      }
    }
    __syncthreads();
    for (int d = lane; d < AP_RADIX; d += 64) hc[(size_t)d * ntc + tile] = h[d];
    __syncthreads();
  }
}

// per class: exclusive scan over its (digit, tile) counts in place (digit-major = the order a stable scatter fills the output)
__global__ __launch_bounds__(256) void ap32_scan_kernel(unsigned* __restrict__ hist, const unsigned* __restrict__ len, int ntiles_max) {
  __shared__ unsigned wsum[4];
  __shared__ unsigned carry_s;
  const int ntc = (int)((len[blockIdx.x] + AP_TILE - 1) / AP_TILE);
  unsigned* h = hist + (size_t)blockIdx.x * AP_RADIX * ntiles_max;
  const int total = AP_RADIX * ntc, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0u;
  __syncthreads();
  for (int base = 0; base < total; base += 1024) {
    const int j = base + tid * 4;
    unsigned v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = j + e < total ? h[j + e] : 0u;
    const unsigned mine = v[0] + v[1] + v[2] + v[3];
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned off = carry_s;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    unsigned run = off + incl - mine;
#pragma unroll
    for (int e = 0; e < 4; ++e) { if (j + e < total) h[j + e] = run; run += v[e]; }
    __syncthreads();
    if (tid == 255) carry_s = off + incl;
    __syncthreads();
  }
}

// stable scatter of a class's tiles by the current digit
__global__ __launch_bounds__(64) void ap32_scatter_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst,
                                                          const unsigned* __restrict__ len, long long n, int ntiles_max, int shift,
                                                          const unsigned* __restrict__ hist) {
  __shared__ unsigned cur[AP_RADIX];
  const int c = blockIdx.y, lane = threadIdx.x;
  const unsigned P = len[c];
  const int ntc = (int)((P + AP_TILE - 1) / AP_TILE);
  const unsigned* col = src + (size_t)c * n;
  unsigned* out = dst + (size_t)c * n;
  const unsigned* hc = hist + (size_t)c * AP_RADIX * ntiles_max;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int tile = blockIdx.x; tile < ntc; tile += gridDim.x) {
    __syncthreads();
    for (int d = lane; d < AP_RADIX; d += 64) cur[d] = hc[(size_t)d * ntc + tile];
    __syncthreads();
    const unsigned i0 = (unsigned)tile * AP_TILE;
    for (int j = 0; j < AP_TILE; j += 64) {
      const unsigned i = i0 + j + lane;
      const bool live = i < P;
      const unsigned k = live ? col[i] : 0u;
      const unsigned d = (k >> shift) & (AP_RADIX - 1);
      // lanes of this step that hold my digit: AND over the digit's bits of (ballot if my bit is set, else its complement)
      unsigned long long same = __ballot(live);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long m = __ballot(live && ((d >> b) & 1u));
        same &= ((d >> b) & 1u) ? m : ~m;
      }
      const unsigned rank = (unsigned)__popcll(same & below);
      const unsigned base = live ? cur[d] : 0u;
      __syncthreads();                                            // every lane has read its digit's cursor
      if (live) {
        out[base + rank] = k;
        if ((same >> lane) >> 1 == 0ull) cur[d] = base + rank + 1u;   // the highest lane of the group moves the cursor
      }
      __syncthreads();
    }
  }
}

// branch-free lower_bound over t[0 .. m), m >= 1: the first index whose key is >= k (m if none)
__device__ __forceinline__ unsigned ap_lower_bound(const unsigned* t, unsigned m, unsigned k) {
  unsigned base = 0u, nn = m;
  while (nn > 1u) {
    const unsigned half = nn >> 1;
    base += (t[base + half - 1u] < k) ? half : 0u;
    nn -= half;
  }
  return base + (t[base] < k ? 1u : 0u);
}

// ap_count: the keys of class c against the class's sorted positives q[0 .. P).  Workgroup (x, c) of AP_SPLITS per class.
// * OWNED RANGES (P <= AP_SPLITS * AP_OWN, every eval set of the shipped configs): the positives are cut into Sb = ceil(P / AP_OWN)
//   ranges, the frames into Sf = AP_SPLITS / Sb slices; a workgroup holds ITS range of q and its counters in LDS, streams its slice of
//   the column, keeps the keys whose lower_bound falls into its range (two compares) and counts them with LDS atomics - hot counters
//   (ties, a constant column) cost LDS cycles, not device-wide atomics.  The Sf slices of a range add their counters into cnt behind
//   the loop: P * Sf adds per class instead of one per key (measured: one device atomic per key = 5.6 ms of the eval set's 7.0 with
//   distinct scores, 14.5 of 17.9 with heavy ties).
// * FALLBACK (more positives than that: multi-label targets): every AP_SPLITS-th slice of the column against a table of every sub-th
//   positive + a second search level in the list itself, one device atomic per counted key (equal counters of a wave merged first).
__global__ __launch_bounds__(1024) void ap_count_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ q_all,
                                                        const unsigned* __restrict__ len, long long n, int C, unsigned* __restrict__ cnt_all,
                                                        double* __restrict__ partial) {
  extern __shared__ unsigned ap_lds[];
  unsigned* tab = ap_lds;                                  // AP_OWN keys + AP_OWN counters, or AP_TAB keys (fallback)
  unsigned* lc = ap_lds + AP_OWN;
  double* red = (double*)(ap_lds + AP_TAB);                // 16 wave sums
  const int c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = gridDim.x;
  const unsigned P = len[c];
  const unsigned* q = q_all + (size_t)c * n;
  unsigned* cnt = cnt_all + (size_t)c * n;
  const unsigned* col = keys + (size_t)c * n;
  const unsigned Sb = (P + AP_OWN - 1) / AP_OWN;
  double ss = 0.0;
  if (Sb <= (unsigned)S) {
    // ---- owned ranges (P = 0: one range of nothing, the score mass only) ----
    const unsigned sbn = Sb ? Sb : 1u;
    const unsigned Sf = (unsigned)S / sbn;
    const unsigned sb = blockIdx.x % sbn, sf = blockIdx.x / sbn;
    if (sf < Sf) {
      const unsigned per = (P + sbn - 1) / sbn;
      const unsigned b0 = min(P, sb * per), b1 = min(P, b0 + per), m = b1 - b0;
      for (unsigned j = tid; j < m; j += 1024) { tab[j] = q[b0 + j]; lc[j] = 0u; }
      const bool open_lo = b0 == 0u;
      const unsigned klo = open_lo ? 0u : q[b0 - 1];         // a key belongs here iff klo < k <= khi
      const unsigned khi = m ? q[b1 - 1] : 0u;
      __syncthreads();
      const long long chunk = (n + Sf - 1) / Sf;
      const long long lo = (long long)sf * chunk, hi = min(n, lo + chunk);
      const bool mass = sb == 0u;                            // one range per slice adds up the scores
      for (long long base = lo; base < hi; base += 4096) {
        unsigned k[4];
        bool mine[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const long long i = base + e * 1024 + tid;
          const bool live = i < hi;
          k[e] = live ? col[i] : 0xFFFFFFFFu;
          if (live && mass) ss += (double)ap_key_score(k[e]);
          mine[e] = live && m && (open_lo || k[e] > klo) && k[e] <= khi;
        }
        if (!(mine[0] || mine[1] || mine[2] || mine[3])) continue;
        unsigned bs[4] = {0u, 0u, 0u, 0u}, nn = m;
        while (nn > 1u) {
          const unsigned half = nn >> 1;
#pragma unroll
          for (int e = 0; e < 4; ++e) if (mine[e]) bs[e] += (tab[bs[e] + half - 1u] < k[e]) ? half : 0u;
          nn -= half;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) if (mine[e]) atomicAdd(&lc[bs[e] + (tab[bs[e]] < k[e] ? 1u : 0u)], 1u);
      }
      __syncthreads();
      for (unsigned j = tid; j < m; j += 1024) {
        const unsigned v = lc[j];
        if (v) { if (Sf == 1u) cnt[b0 + j] = v; else atomicAdd(&cnt[b0 + j], v); }
      }
    }
  } else {
    // ---- fallback: one device atomic per counted key ----
    const unsigned sub = (P + AP_TAB - 1) / AP_TAB;          // >= 2 here
    const unsigned m = (P + sub - 1) / sub;                  // table entry j = the LAST key of block j (blocks of `sub` keys)
    for (unsigned j = tid; j < m; j += 1024) { const unsigned long long e = (unsigned long long)(j + 1) * sub - 1; tab[j] = q[e < P ? e : P - 1]; }
    __syncthreads();
    const long long chunk = (n + S - 1) / S;
    const long long lo = (long long)blockIdx.x * chunk, hi = min(n, lo + chunk);
    for (long long base = lo; base < hi; base += 4096) {
      unsigned k[4], b[4];
      bool live[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long long i = base + e * 1024 + tid;
        live[e] = i < hi;
        k[e] = live[e] ? col[i] : 0xFFFFFFFFu;
        if (live[e]) ss += (double)ap_key_score(k[e]);
      }
      {                                                       // four searches in step (same trip count for every lane)
        unsigned bs[4] = {0u, 0u, 0u, 0u}, nn = m;
        while (nn > 1u) {
          const unsigned half = nn >> 1;
#pragma unroll
          for (int e = 0; e < 4; ++e) bs[e] += (tab[bs[e] + half - 1u] < k[e]) ? half : 0u;
          nn -= half;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = bs[e] + (tab[bs[e]] < k[e] ? 1u : 0u);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {                           // second level: inside block b[e] of the list itself
        if (b[e] >= m) { b[e] = P; continue; }
        const unsigned s0 = b[e] * sub, cntk = min(sub, P - s0);
        b[e] = s0 + ap_lower_bound(q + s0, cntk, k[e]);       // < s0 + cntk: the block's last key is >= k
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bool todo = live[e] && b[e] < P;
        // equal counters inside the wave first (a constant column, a score above every positive, heavy ties): one add for all of them
        for (int r = 0; r < 2; ++r) {
          const unsigned long long act = __ballot(todo);
          if (act == 0ull) break;
          const int first = __ffsll((long long)act) - 1;
          const unsigned bf = (unsigned)__shfl((int)b[e], first, 64);
          const unsigned long long same = __ballot(todo && b[e] == bf);
          const int ns = __popcll(same);
          if (ns < 8 && r == 0) break;                        // no hot counter in this wave
          if (lane == first) atomicAdd(&cnt[bf], (unsigned)ns);
          if (todo && b[e] == bf) todo = false;
        }
        if (todo) atomicAdd(&cnt[b[e]], 1u);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  if (tid == 0) {
    double a = 0.0;
    for (int w = 0; w < 16; ++w) a += red[w];
    partial[(size_t)blockIdx.x * C + c] = a;
  }
}

// one workgroup per class over its sorted positives and their counters
__global__ __launch_bounds__(256) void ap_reduce_kernel(const unsigned* __restrict__ q_all, const unsigned* __restrict__ cnt_all,
                                                        const unsigned* __restrict__ len, long long n, int C, const double* __restrict__ partial,
                                                        int splits, double* __restrict__ ap, long long* __restrict__ n_pos,
                                                        double* __restrict__ score_sum) {
  __shared__ long long w_sum[4], w_max[4];
  __shared__ double w_acc[4];
  __shared__ long long carry_cnt, carry_end;
  const int c = blockIdx.x;
  const long long P = (long long)len[c];
  const unsigned* q = q_all + (size_t)c * n;
  const unsigned* cnt = cnt_all + (size_t)c * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { carry_cnt = 0; carry_end = 0; }
  __syncthreads();
  double acc = 0.0;
  for (long long base = 0; base < P; base += 1024) {
    const long long j = base + (long long)tid * 4;
    unsigned k[5];
    long long v[4], mine = 0;
#pragma unroll
    for (int e = 0; e < 5; ++e) k[e] = j + e < P ? q[j + e] : 0u;             // k[4]: the neighbour behind my last element
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = j + e < P ? (long long)cnt[j + e] : 0; mine += v[e]; }
    long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const long long u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    if (lane == 63) w_sum[wave] = incl;
    __syncthreads();
    long long seen = carry_cnt + incl - mine;                                 // samples at or above the score in front of my first element
    for (int w = 0; w < wave; ++w) seen += w_sum[w];
    // my elements: samples at or above each, and whether it ends a run of equal scores (the last positive ends one)
    long long s_after[4], end_local = 0;                                      // largest run-end rank among my elements (0 = none)
    bool is_end[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      seen += v[e];
      s_after[e] = seen;
      is_end[e] = (j + e < P) && ((j + e + 1 >= P) || (k[e] != k[e + 1]));
      if (is_end[e]) end_local = j + e + 1;
    }
    // positives at the most recent run end BEFORE my first element: exclusive max-scan (ranks are monotone)
    long long mx = end_local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const long long u = __shfl_up(mx, o, 64); if (lane >= o && u > mx) mx = u; }
    if (lane == 63) w_max[wave] = mx;
    __syncthreads();
    long long prev = __shfl_up(mx, 1, 64);
    if (lane == 0) prev = 0;
    if (carry_end > prev) prev = carry_end;
    for (int w = 0; w < wave; ++w) if (w_max[w] > prev) prev = w_max[w];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (is_end[e]) {
        const long long tps = j + e + 1;
        acc += (double)(tps - prev) * ((double)tps / (double)s_after[e]);
        prev = tps;
      }
    __syncthreads();
    if (tid == 255) {
      carry_cnt = seen;
      long long mm = carry_end;
      for (int w = 0; w < 4; ++w) if (w_max[w] > mm) mm = w_max[w];
      carry_end = mm;
    }
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) w_acc[wave] = acc;
  __syncthreads();
  if (tid == 0) {
    const double a = (w_acc[0] + w_acc[1]) + (w_acc[2] + w_acc[3]);
    ap[c] = P > 0 ? a / (double)P : __longlong_as_double(0x7FF8000000000000ll);     // NaN: a class without positives
    if (n_pos) n_pos[c] = P;
    if (score_sum) {
      double t = 0.0;
      for (int sp = 0; sp < splits; ++sp) t += partial[(size_t)sp * C + c];
      score_sum[c] = t;
    }
  }
}

static int ap_splits(int C) {
  int s = 1024 / C;                      // workgroups per class: 16 (owned ranges up to 131 072 positives per class) or, for few classes, enough to fill the chip
  return s < 16 ? 16 : (s > AP_SPLITS_MAX ? AP_SPLITS_MAX : s);
}

size_t perframe_ap_workspace_bytes(long long n, int C) {
  const size_t ntiles = (size_t)((n + AP_TILE - 1) / AP_TILE);
  return 4 * (size_t)C * (size_t)n * 4 + (size_t)C * AP_RADIX * ntiles * 4 + (size_t)C * 4 + 8 + (size_t)AP_SPLITS_MAX * C * 8 + 1024;
}

// Returns 0, or -1 on a bad argument.  ws must hold perframe_ap_workspace_bytes(n, C) bytes.  Positives: target != 0, or (labels != NULL)
// the frame's class id.
int launch_perframe_ap(const float* scores, const float* target, const int* labels, long long n, int C, double* ap, long long* n_pos,
                       double* score_sum, void* ws, hipStream_t s) {
  if (n <= 0 || C <= 0 || C > 65535 || n >= (1ll << 31)) return -1;    // per-class cursors are 32-bit
  static DeviceOnce once;
  once.run([] { (void)hipFuncSetAttribute((const void*)ap_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, AP_TAB * 4 + 128); });
  const int ntiles = (int)((n + AP_TILE - 1) / AP_TILE);
  const size_t cn = (size_t)C * (size_t)n;
  unsigned* keys = (unsigned*)ws;
  unsigned* p0 = keys + cn;
  unsigned* p1 = p0 + cn;
  unsigned* cnt = p1 + cn;
  unsigned* hist = cnt + cn;
  unsigned* cursor = hist + (size_t)C * AP_RADIX * ntiles;
  double* partial = (double*)(((uintptr_t)(cursor + C) + 7) & ~(uintptr_t)7);
  (void)hipMemsetAsync(cursor, 0, (size_t)C * 4, s);
  const dim3 xg((unsigned)((n + AP_XF - 1) / AP_XF), (unsigned)((C + AP_XC - 1) / AP_XC));
  if (labels) ap_extract_kernel<true><<<xg, 256, 0, s>>>(scores, nullptr, labels, n, C, keys, p0, cursor);
  else ap_extract_kernel<false><<<xg, 256, 0, s>>>(scores, target, nullptr, n, C, keys, p0, cursor);
  const int tg = ntiles < AP_TGRID ? ntiles : AP_TGRID;
  for (int p = 0; p < AP_PASSES; ++p) {
    ap32_hist_kernel<<<dim3(tg, C), 64, 0, s>>>(p0, cursor, n, ntiles, 8 * p, hist, p == 0 ? cnt : nullptr);
    ap32_scan_kernel<<<C, 256, 0, s>>>(hist, cursor, ntiles);
    ap32_scatter_kernel<<<dim3(tg, C), 64, 0, s>>>(p0, p1, cursor, n, ntiles, 8 * p, hist);
    unsigned* t = p0; p0 = p1; p1 = t;
  }
  const int splits = ap_splits(C);
  ap_count_kernel<<<dim3(splits, C), 1024, AP_TAB * 4 + 128, s>>>(keys, p0, cursor, n, C, cnt, partial);
  ap_reduce_kernel<<<C, 256, 0, s>>>(p0, cnt, cursor, n, C, partial, splits, ap, n_pos, score_sum);
  return 0;
}
