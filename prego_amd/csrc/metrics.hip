// Per-frame average precision on the device: step_recognition/utils/metrics.py:25-62 calls
// sklearn.metrics.average_precision_score once per class on the [frames x classes] score matrix of an eval pass (main.py:101 runs
// that after every epoch).  The definition (sklearn: precision_recall_curve + step-wise sum): sort a class's scores descending;
// the thresholds are the DISTINCT score values (ties share one threshold); with tps_k / cnt_k = positives / samples at or above
// threshold k and P positives in all,   AP = sum_k (tps_k - tps_{k-1}) / P * tps_k / cnt_k.
//
// HBM-bound integer work, done as a segmented LSD radix sort (one segment per class) plus one scan:
//   ap_keys     [frames][classes] scores / targets (row-major, as the head kernel writes them)  ->  per class a column of 64-bit
//               keys (order-reversing transform of the fp32 bits) << 1 | label: ascending key order = descending score, the label
//               rides in bit 0 (the order inside a tie run does not matter: a run is one threshold)
//   ap_hist / ap_scan / ap_scatter   five stable passes over 8-bit digits of key bits [1, 41) - bit 0 is the label and needs no
//               sorting, bits 33.. are zero - one wave per 4096-element tile: digit histogram in LDS -> per-class exclusive scan
//               over (digit, tile) -> stable scatter (a lane's rank inside a 64-element step comes from eight ballots: the lanes
//               that hold the same digit, below it)
//   ap_reduce   one workgroup per class walks the sorted column once: running positives (block sum-scan), positives at the previous
//               threshold (block max-scan over the run ends - tps is monotone), fp64 accumulation of the step-wise sum.
// Exact integer ranks; the only floating-point work is the final fp64 sum (differs from sklearn's by summation order, ~1e-16).
#include "common.h"
#include "kernels.h"

#define AP_TILE 4096          // elements per radix tile (one wave, 64 steps of 64)
#define AP_RADIX 256
#define AP_PASSES 4           // key bits [1, 33): the key is the 32-bit score key << 1 | label, bits above 32 are zero (a fifth pass only copied)

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

// keys[c][i] for a block of 256 frames x all classes
__global__ __launch_bounds__(256) void ap_keys_kernel(const float* __restrict__ scores, const float* __restrict__ target, long long n,
                                                      int C, unsigned long long* __restrict__ keys) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* s = scores + i * C;
  const float* t = target + i * C;
  for (int c = 0; c < C; ++c)
    keys[(size_t)c * n + i] = ((unsigned long long)ap_desc_key(s[c]) << 1) | (t[c] != 0.f ? 1ull : 0ull);
}

// digit histogram of one tile: hist[c][digit][tile]
__global__ __launch_bounds__(64) void ap_hist_kernel(const unsigned long long* __restrict__ keys, long long n, int ntiles, int shift,
                                                     unsigned* __restrict__ hist) {
  __shared__ unsigned h[AP_RADIX];
  const int tile = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
  for (int d = lane; d < AP_RADIX; d += 64) h[d] = 0u;
  __syncthreads();
  const unsigned long long* col = keys + (size_t)c * n;
  const long long i0 = (long long)tile * AP_TILE;
  for (int j = lane; j < AP_TILE; j += 64) {
    const long long i = i0 + j;
    if (i < n) atomicAdd(&h[(unsigned)(col[i] >> shift) & (AP_RADIX - 1)], 1u);
  }
  __syncthreads();
  for (int d = lane; d < AP_RADIX; d += 64) hist[((size_t)c * AP_RADIX + d) * ntiles + tile] = h[d];
}

// per class: exclusive scan over the (digit, tile) counts in place (digit-major = the order a stable scatter fills the output)
__global__ __launch_bounds__(256) void ap_scan_kernel(unsigned* __restrict__ hist, int ntiles) {
  __shared__ unsigned wsum[4];
  __shared__ unsigned carry_s;
  unsigned* h = hist + (size_t)blockIdx.x * AP_RADIX * ntiles;
  const int total = AP_RADIX * ntiles, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

// stable scatter of one tile by the current digit
__global__ __launch_bounds__(64) void ap_scatter_kernel(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst,
                                                        long long n, int ntiles, int shift, const unsigned* __restrict__ hist) {
  __shared__ unsigned pos[AP_RADIX];
  const int tile = blockIdx.x, c = blockIdx.y, lane = threadIdx.x;
  for (int d = lane; d < AP_RADIX; d += 64) pos[d] = hist[((size_t)c * AP_RADIX + d) * ntiles + tile];
  __syncthreads();
  const unsigned long long* col = src + (size_t)c * n;
  unsigned long long* out = dst + (size_t)c * n;
  const long long i0 = (long long)tile * AP_TILE;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int j = 0; j < AP_TILE; j += 64) {
    const long long i = i0 + j + lane;
    const bool live = i < n;
    const unsigned long long k = live ? col[i] : 0ull;
    const unsigned d = (unsigned)(k >> shift) & (AP_RADIX - 1);
    // lanes of this step that hold my digit: AND over the digit's bits of (ballot if my bit is set, else its complement)
    unsigned long long same = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot(live && ((d >> b) & 1u));
      same &= ((d >> b) & 1u) ? m : ~m;
    }
    const unsigned rank = (unsigned)__popcll(same & below);
    const unsigned base = live ? pos[d] : 0u;
    __syncthreads();                                            // every lane has read its digit's cursor
    if (live) {
      out[base + rank] = k;
      if ((same >> lane) >> 1 == 0ull) pos[d] = base + rank + 1u;   // the highest lane of the group moves the cursor
    }
    __syncthreads();
  }
}

// one workgroup per class over its sorted column
__global__ __launch_bounds__(256) void ap_reduce_kernel(const unsigned long long* __restrict__ keys, long long n, double* __restrict__ ap,
                                                        long long* __restrict__ n_pos, double* __restrict__ score_sum) {
  __shared__ long long w_sum[4], w_max[4];
  __shared__ double w_acc[4], w_ss[4];
  __shared__ long long carry_tps, carry_end;
  const unsigned long long* col = keys + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { carry_tps = 0; carry_end = 0; }
  __syncthreads();
  double acc = 0.0, ss = 0.0;
  for (long long base = 0; base < n; base += 1024) {
    const long long j = base + (long long)tid * 4;
    unsigned long long k[5];
#pragma unroll
    for (int e = 0; e < 5; ++e) k[e] = j + e < n ? col[j + e] : ~0ull;       // k[4]: the neighbour behind my last element
    int lab[4];
    long long mine = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { lab[e] = (j + e < n) ? (int)(k[e] & 1ull) : 0; mine += lab[e]; if (j + e < n) ss += (double)ap_key_score((unsigned)(k[e] >> 1)); }
    long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const long long u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    if (lane == 63) w_sum[wave] = incl;
    __syncthreads();
    long long tps = carry_tps + incl - mine;                                  // positives before my first element
    for (int w = 0; w < wave; ++w) tps += w_sum[w];
    // my elements: tps after each, and whether it ends a run of equal scores (the last element of the column ends one)
    long long t_after[4], end_tps_local = 0;                                  // largest run-end tps among my elements (0 = none)
    bool is_end[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      tps += lab[e];
      t_after[e] = tps;
      is_end[e] = (j + e < n) && ((j + e + 1 >= n) || ((k[e] >> 1) != (k[e + 1] >> 1)));
      if (is_end[e]) end_tps_local = tps;
    }
    // tps at the most recent run end BEFORE my first element: exclusive max-scan (tps is monotone)
    long long mx = end_tps_local;
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
        acc += (double)(t_after[e] - prev) * ((double)t_after[e] / (double)(j + e + 1));
        prev = t_after[e];
      }
    __syncthreads();
    if (tid == 255) {
      carry_tps = tps;
      long long m = carry_end;
      for (int w = 0; w < 4; ++w) if (w_max[w] > m) m = w_max[w];
      carry_end = m;
    }
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o, 64); ss += __shfl_xor(ss, o, 64); }
  if (lane == 0) { w_acc[wave] = acc; w_ss[wave] = ss; }
  __syncthreads();
  if (tid == 0) {
    const long long P = carry_tps;
    const double a = (w_acc[0] + w_acc[1]) + (w_acc[2] + w_acc[3]);
    ap[blockIdx.x] = P > 0 ? a / (double)P : __longlong_as_double(0x7FF8000000000000ll);     // NaN: a class without positives
    if (n_pos) n_pos[blockIdx.x] = P;
    if (score_sum) score_sum[blockIdx.x] = (w_ss[0] + w_ss[1]) + (w_ss[2] + w_ss[3]);
  }
}

size_t perframe_ap_workspace_bytes(long long n, int C) {
  const size_t ntiles = (size_t)((n + AP_TILE - 1) / AP_TILE);
  return 2 * (size_t)C * (size_t)n * 8 + (size_t)C * AP_RADIX * ntiles * 4 + 1024;
}

// Returns 0, or -1 on a bad argument.  ws must hold perframe_ap_workspace_bytes(n, C) bytes.
int launch_perframe_ap(const float* scores, const float* target, long long n, int C, double* ap, long long* n_pos, double* score_sum,
                       void* ws, hipStream_t s) {
  if (n <= 0 || C <= 0 || C > 65535 || n >= (1ll << 31)) return -1;    // per-class cursors are 32-bit
  const int ntiles = (int)((n + AP_TILE - 1) / AP_TILE);
  unsigned long long* k0 = (unsigned long long*)ws;
  unsigned long long* k1 = k0 + (size_t)C * n;
  unsigned* hist = (unsigned*)(k1 + (size_t)C * n);
  ap_keys_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(scores, target, n, C, k0);
  for (int p = 0; p < AP_PASSES; ++p) {
    const int shift = 1 + 8 * p;
    ap_hist_kernel<<<dim3(ntiles, C), 64, 0, s>>>(k0, n, ntiles, shift, hist);
    ap_scan_kernel<<<C, 256, 0, s>>>(hist, ntiles);
    ap_scatter_kernel<<<dim3(ntiles, C), 64, 0, s>>>(k0, k1, n, ntiles, shift, hist);
    unsigned long long* t = k0; k0 = k1; k1 = t;
  }
  ap_reduce_kernel<<<C, 256, 0, s>>>(k0, n, ap, n_pos, score_sum);
  return 0;
}
