// Post-processing of the per-frame predictions on the device:
//   window_vote   utils/aggregate.py:55-72 - non-overlapping windows of `window` (200) frames, each replaced by its most frequent
//                 class (np.argmax(np.bincount(...)): the LOWEST class id wins a tie); the int32 per-frame argmax the head kernel
//                 wrote never leaves HBM, only one int32 per window does.
// One wave per window: a segmented histogram in LDS (integer atomics: exact and order-independent), then a 64-lane argmax over
// (count, -class).
//   format_ids    trainer/eval.py:59-65 writes {vid: {"pred": [...], "gt": [...]}} as JSON text: 2 x frames integers.  The ids are on
//                 the device; every id becomes its four bytes "%3d," here, so the host cuts the text per video instead of formatting
//                 4.6 M numbers behind the last frame of an eval pass (blanks in front of a number are JSON whitespace).
#include "common.h"
#include "kernels.h"

#define VOTE_MAX_CLASSES 128

__global__ __launch_bounds__(256) void window_vote_kernel(const int* __restrict__ pred, long long n_frames, int window, int n_classes,
                                                          int* __restrict__ votes) {
  __shared__ int hist[4][VOTE_MAX_CLASSES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long n_win = (n_frames + window - 1) / window;
  for (long long w0 = (long long)blockIdx.x * 4; w0 < n_win; w0 += (long long)gridDim.x * 4) {   // block-uniform trip count
    const long long w = w0 + wave;
    bool bad_id = false;                      // an id outside [0, n_classes): np.bincount would raise (negative) or count a class the
                                              // model does not have - the window votes -1 and the host turns that into an error
    hist[wave][lane] = 0; hist[wave][lane + 64] = 0;
    __syncthreads();
    if (w < n_win) {
      const long long s = w * window;
      const long long e = s + window < n_frames ? s + window : n_frames;
      for (long long i = s + lane; i < e; i += 64) {
        const int c = pred[i];
        if (c >= 0 && c < n_classes) atomicAdd(&hist[wave][c], 1);
        else bad_id = true;
      }
    }
    __syncthreads();
    if (w < n_win) {
      // best = max count, ties -> lowest class id (np.argmax takes the first maximum)
      int c0 = hist[wave][lane], c1 = hist[wave][lane + 64];
      int best_cnt = c0, best_id = lane;
      if (c1 > best_cnt) { best_cnt = c1; best_id = lane + 64; }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const int oc = __shfl_xor(best_cnt, off, 64), oi = __shfl_xor(best_id, off, 64);
        if (oc > best_cnt || (oc == best_cnt && oi < best_id)) { best_cnt = oc; best_id = oi; }
      }
      if (__any(bad_id)) best_id = -1;
      if (lane == 0) votes[w] = best_id;
    }
    __syncthreads();
  }
}

int launch_window_vote(const int* pred, long long n_frames, int window, int n_classes, int* votes, hipStream_t s) {
  if (n_frames <= 0 || window <= 0 || n_classes <= 0 || n_classes > VOTE_MAX_CLASSES) return -1;
  const long long n_win = (n_frames + window - 1) / window;
  long long blocks = (n_win + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  window_vote_kernel<<<(int)blocks, 256, 0, s>>>(pred, n_frames, window, n_classes, votes);
  return 0;
}

__global__ __launch_bounds__(256) void format_ids_kernel(const int* __restrict__ ids, long long n, unsigned* __restrict__ text, int* __restrict__ bad) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int v = ids[i];
  if (v < 0 || v > 999) { if (bad) *bad = 1; text[i] = 0x2C302020u; return; }      // "  0,": the host does not use a text with the flag set
  const unsigned h = (unsigned)v / 100u, t = ((unsigned)v / 10u) % 10u, u = (unsigned)v % 10u;
  const unsigned c0 = h ? '0' + h : ' ';
  const unsigned c1 = (h || t) ? '0' + t : ' ';
  text[i] = c0 | (c1 << 8) | (('0' + u) << 16) | ((unsigned)',' << 24);             // little-endian: byte 0 first
}

int launch_format_ids(const int* ids, long long n, unsigned* text, int* bad, hipStream_t s) {
  if (n <= 0) return -1;
  format_ids_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(ids, n, text, bad);
  return 0;
}
