// C ABI + host-side planner of the MiniROAD hot path (include/prego_amd.h).
// Host logic only: packing plan (sort clips by length, packed time-major rows, chunking), workspace
// carving, weight ingestion, and the per-chunk launch sequence
//   pack -> GEMM(layer1) -> LayerNorm+ReLU -> GEMM(W_ih) -> persistent GRU recurrence -> head+softmax+argmax.
#include "../../include/prego_amd.h"
#include "kernels.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <numeric>
#include <string>
#include <vector>

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) return fail(PREGO_EHIP, "%s failed: %s", #x, hipGetErrorString(e_));      \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct EventPair { hipEvent_t a, b; };

struct prego_miniroad {
  int d_rgb, d_flow, emb, hid, ncls, ncls_pad;
  bool bf16;
  int n_cu;
  int G, P;                     // recurrence groups / workgroups per group
  // ingested weights (device, handle-owned)
  void* w1 = nullptr;           // [emb][d_rgb+d_flow] WT
  float* b1 = nullptr;
  float* ln_g = nullptr;
  float* ln_b = nullptr;
  void* w_ih = nullptr;         // [3H][emb] WT
  void* w_hh = nullptr;         // [3H][H] WT
  float* bias2 = nullptr;       // b_ih + (b_hh for r,z rows)
  float* b_hn = nullptr;        // [H]
  void* w_c = nullptr;          // [ncls_pad][H] WT zero padded
  float* b_c = nullptr;         // [ncls_pad]
  bool have_weights = false;
  // recurrence scratch
  void* hx = nullptr;           // [G][2][64][H] WT
  unsigned* flags = nullptr;    // [G*P] + abort word
  unsigned* abort_word = nullptr;
  float* h_state = nullptr;     // [max_clips][H]
  unsigned long long* stamps = nullptr;   // debug phase counters (PREGO_GRU_STAMPS=1)
  bool use_stamps = false;
  // plan cache
  std::vector<int32_t> plan_lens;
  std::vector<int> h_rowoff, h_nact, h_sorted;
  int t_max = 0;
  int* d_rowoff = nullptr; int* d_nact = nullptr; int* d_sorted = nullptr;
  size_t cap_t = 0, cap_c = 0;
  // per-call pointer tables (device)
  void** d_ptrs = nullptr;      // [4][max_clips]
  // timing
  bool timing = false;
  std::vector<EventPair> ev_pool;
  std::vector<int> ev_kind;     // 0 gemm, 1 gru, 2 pack
  size_t ev_used = 0;
  double gemm_flop = 0, pack_bytes = 0;
};

static int max_clips_of(const prego_miniroad* h) { return h->G * 64; }

extern "C" int prego_abi_version(void) { return PREGO_ABI_VERSION; }
extern "C" const char* prego_last_error(void) { return g_err.c_str(); }

extern "C" int prego_miniroad_create(prego_miniroad** out, int d_rgb, int d_flow, int emb, int hid, int n_classes,
                                     int compute_dtype) {
  if (!out) return fail(PREGO_EINVAL, "out is NULL");
  *out = nullptr;
  if (compute_dtype != PREGO_F32 && compute_dtype != PREGO_BF16) return fail(PREGO_EINVAL, "compute_dtype %d", compute_dtype);
  if (hid != 1024) return fail(PREGO_EINVAL, "hidden_dim %d unsupported: the register-resident recurrence is built for 1024", hid);
  if (emb <= 0 || emb % 256 || emb > 4096) return fail(PREGO_EINVAL, "embedding_dim %d must be a multiple of 256, <= 4096", emb);
  if (d_rgb < 0 || d_flow < 0 || d_rgb + d_flow <= 0 || (d_rgb % 64) || (d_flow % 64))
    return fail(PREGO_EINVAL, "feature sizes %d/%d must be multiples of 64", d_rgb, d_flow);
  if (n_classes <= 0 || n_classes > 128) return fail(PREGO_EINVAL, "num_classes %d must be in 1..128", n_classes);
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, dev));
  prego_miniroad* h = new prego_miniroad();
  h->d_rgb = d_rgb; h->d_flow = d_flow; h->emb = emb; h->hid = hid; h->ncls = n_classes;
  h->ncls_pad = (n_classes + 15) / 16 * 16;
  h->bf16 = compute_dtype == PREGO_BF16;
  h->n_cu = prop.multiProcessorCount;
  h->P = h->bf16 ? 32 : 64;
  h->G = std::min(h->bf16 ? 8 : 4, h->n_cu / h->P);
  if (h->G < 1) { delete h; return fail(PREGO_EINVAL, "device has %d CUs, the recurrence needs >= %d", prop.multiProcessorCount, h->bf16 ? 32 : 64); }
  const size_t es = h->bf16 ? 2 : 4;
  const int din = d_rgb + d_flow, H = hid;
  hipError_t e = hipSuccess;
  auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
  A(&h->w1, (size_t)emb * din * es); A((void**)&h->b1, emb * 4); A((void**)&h->ln_g, emb * 4); A((void**)&h->ln_b, emb * 4);
  A(&h->w_ih, (size_t)3 * H * emb * es); A(&h->w_hh, (size_t)3 * H * H * es);
  A((void**)&h->bias2, 3 * H * 4); A((void**)&h->b_hn, H * 4);
  A(&h->w_c, (size_t)h->ncls_pad * H * es); A((void**)&h->b_c, h->ncls_pad * 4);
  A(&h->hx, (size_t)h->G * 2 * 64 * H * es);
  A((void**)&h->flags, ((size_t)h->G * h->P + 16) * sizeof(unsigned));
  A((void**)&h->h_state, (size_t)max_clips_of(h) * H * 4);
  A((void**)&h->d_ptrs, (size_t)4 * max_clips_of(h) * sizeof(void*));
  A((void**)&h->stamps, 8 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(h->stamps, 0, 8 * sizeof(unsigned long long));
  h->use_stamps = getenv("PREGO_GRU_STAMPS") != nullptr;
  if (e == hipSuccess) e = hipMemset(h->hx, 0, (size_t)h->G * 2 * 64 * H * es);
  if (e == hipSuccess) e = hipMemset(h->flags, 0, ((size_t)h->G * h->P + 16) * sizeof(unsigned));
  if (e != hipSuccess) { prego_miniroad_destroy(h); return fail(PREGO_EHIP, "hipMalloc: %s", hipGetErrorString(e)); }
  h->abort_word = h->flags + (size_t)h->G * h->P;
  *out = h;
  return PREGO_OK;
}

extern "C" void prego_miniroad_destroy(prego_miniroad* h) {
  if (!h) return;
  void* ptrs[] = {h->w1, h->b1, h->ln_g, h->ln_b, h->w_ih, h->w_hh, h->bias2, h->b_hn, h->w_c, h->b_c, h->hx,
                  h->flags, h->h_state, h->stamps, h->d_rowoff, h->d_nact, h->d_sorted, h->d_ptrs};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (auto& ev : h->ev_pool) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  delete h;
}

extern "C" int prego_miniroad_max_clips(const prego_miniroad* h) { return h ? max_clips_of(h) : 0; }

extern "C" int prego_miniroad_set_weights(prego_miniroad* h, const float* layer1_w, const float* layer1_b,
                                          const float* ln_w, const float* ln_b, const float* w_ih, const float* w_hh,
                                          const float* b_ih, const float* b_hh, const float* fc_w, const float* fc_b,
                                          prego_stream_t stream) {
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  if (!layer1_w || !layer1_b || !ln_w || !ln_b || !w_ih || !w_hh || !b_ih || !b_hh || !fc_w || !fc_b)
    return fail(PREGO_EINVAL, "set_weights: NULL tensor");
  hipStream_t s = (hipStream_t)stream;
  const int din = h->d_rgb + h->d_flow, E = h->emb, H = h->hid;
  launch_pad_convert(h->bf16, layer1_w, E, din, din, h->w1, E, din, s);
  launch_pad_convert(h->bf16, w_ih, 3 * H, E, E, h->w_ih, 3 * H, E, s);
  launch_pad_convert(h->bf16, w_hh, 3 * H, H, H, h->w_hh, 3 * H, H, s);
  launch_pad_convert(h->bf16, fc_w, h->ncls, H, H, h->w_c, h->ncls_pad, H, s);
  launch_pad_convert(false, fc_b, 1, h->ncls, h->ncls, h->b_c, 1, h->ncls_pad, s);
  HIPCHK(hipMemcpyAsync(h->b1, layer1_b, E * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(h->ln_g, ln_w, E * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(h->ln_b, ln_b, E * 4, hipMemcpyDeviceToDevice, s));
  launch_add_vec(b_ih, b_hh, h->bias2, 3 * H, 2 * H, s);   // r,z rows: b_ih + b_hh ; n rows: b_ih
  HIPCHK(hipMemcpyAsync(h->b_hn, b_hh + 2 * H, H * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipGetLastError());
  h->have_weights = true;
  return PREGO_OK;
}

// ---- plan -------------------------------------------------------------------------------------
static int build_plan(prego_miniroad* h, int n, const int32_t* lens, hipStream_t s) {
  if ((int)h->plan_lens.size() == n && std::equal(lens, lens + n, h->plan_lens.begin())) return PREGO_OK;
  int tmax = 0;
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    if (lens[i] <= 0) return fail(PREGO_EINVAL, "clip %d has %d frames", i, lens[i]);
    tmax = std::max(tmax, lens[i]);
    total += lens[i];
  }
  if (total >= (1ll << 31)) return fail(PREGO_EINVAL, "more than 2^31 frames in one call");
  h->h_sorted.resize(n);
  std::iota(h->h_sorted.begin(), h->h_sorted.end(), 0);
  std::stable_sort(h->h_sorted.begin(), h->h_sorted.end(), [&](int a, int b) { return lens[a] > lens[b]; });
  h->h_nact.assign(tmax, 0);
  // nact[t] = #clips with len > t: histogram of lengths, suffix sum
  std::vector<int> cnt(tmax + 1, 0);
  for (int i = 0; i < n; ++i) cnt[lens[i]]++;
  int alive = 0;
  for (int t = tmax; t >= 1; --t) { alive += cnt[t]; h->h_nact[t - 1] = alive; }
  h->h_rowoff.assign(tmax + 1, 0);
  for (int t = 0; t < tmax; ++t) h->h_rowoff[t + 1] = h->h_rowoff[t] + h->h_nact[t];
  if ((size_t)tmax + 1 > h->cap_t) {
    if (h->d_rowoff) (void)hipFree(h->d_rowoff);
    if (h->d_nact) (void)hipFree(h->d_nact);
    h->cap_t = (size_t)tmax + 1 + 1024;
    HIPCHK(hipMalloc((void**)&h->d_rowoff, h->cap_t * 4));
    HIPCHK(hipMalloc((void**)&h->d_nact, h->cap_t * 4));
  }
  if ((size_t)n > h->cap_c) {
    if (h->d_sorted) (void)hipFree(h->d_sorted);
    h->cap_c = (size_t)n + 64;
    HIPCHK(hipMalloc((void**)&h->d_sorted, h->cap_c * 4));
  }
  // pageable-source async copies: the runtime stages the host data before returning
  HIPCHK(hipMemcpyAsync(h->d_rowoff, h->h_rowoff.data(), ((size_t)tmax + 1) * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(h->d_nact, h->h_nact.data(), (size_t)tmax * 4, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(h->d_sorted, h->h_sorted.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
  h->t_max = tmax;
  h->plan_lens.assign(lens, lens + n);
  return PREGO_OK;
}

struct RowBytes { size_t x, y, e, gi, hr, hraw, total; };
static RowBytes row_bytes(const prego_miniroad* h, bool with_flow, int flags) {
  const size_t es = h->bf16 ? 2 : 4;
  RowBytes r;
  r.x = (size_t)(h->d_rgb + (with_flow ? h->d_flow : 0)) * es;
  r.y = (size_t)h->emb * 4;
  r.e = (size_t)h->emb * es;
  r.gi = (size_t)3 * h->hid * 4;
  r.hr = (size_t)h->hid * es;
  r.hraw = (flags & PREGO_FWD_KEEP) ? (size_t)h->hid * 4 : 0;
  r.total = r.x + r.y + r.e + r.gi + r.hr + r.hraw;
  return r;
}

extern "C" size_t prego_miniroad_workspace_bytes(const prego_miniroad* h, int n_clips, const int32_t* lens,
                                                 int64_t rows_per_chunk, int flags) {
  if (!h || n_clips <= 0) return 0;
  long long total = 0;
  if (lens) for (int i = 0; i < n_clips; ++i) total += lens[i];
  long long rows = std::max<long long>(rows_per_chunk, n_clips);
  if (lens && rows > total) rows = std::max<long long>(total, n_clips);
  if (flags & PREGO_FWD_KEEP) rows = std::max<long long>(rows, total);
  rows = (long long)align_up((size_t)rows, 128);
  const RowBytes rb = row_bytes(h, true, flags);
  return (size_t)rows * rb.total + 6 * 256;
}

static EventPair* ev_begin(prego_miniroad* h, int kind, hipStream_t s) {
  if (!h->timing) return nullptr;
  if (h->ev_used == h->ev_pool.size()) {
    EventPair p;
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
    h->ev_pool.push_back(p);
    h->ev_kind.push_back(kind);
  }
  h->ev_kind[h->ev_used] = kind;
  EventPair* p = &h->ev_pool[h->ev_used++];
  (void)hipEventRecord(p->a, s);
  return p;
}
static void ev_end(EventPair* p, hipStream_t s) { if (p) (void)hipEventRecord(p->b, s); }

extern "C" int prego_miniroad_forward(prego_miniroad* h, int n_clips, const int32_t* lens, const float* const* rgb,
                                      const float* const* flow, float* const* out, int32_t* const* argmax,
                                      const float* h0, float* h_last, int flags, void* workspace,
                                      size_t workspace_bytes, prego_stream_t stream) {
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  if (!h->have_weights) return fail(PREGO_EINVAL, "forward before set_weights");
  if (n_clips <= 0 || !lens) return fail(PREGO_EINVAL, "no clips");
  if (n_clips > max_clips_of(h)) return fail(PREGO_EINVAL, "%d clips > max_clips %d per call", n_clips, max_clips_of(h));
  if (h->d_rgb > 0 && !rgb) return fail(PREGO_EINVAL, "rgb pointer array is NULL");
  if (!workspace) return fail(PREGO_EINVAL, "workspace is NULL");
  hipStream_t s = (hipStream_t)stream;
  int rc = build_plan(h, n_clips, lens, s);
  if (rc) return rc;

  // pointer tables -> device
  const int MC = max_clips_of(h);
  bool any_flow = false;
  std::vector<const void*> tab((size_t)4 * MC, nullptr);
  for (int i = 0; i < n_clips; ++i) {
    tab[0 * MC + i] = rgb ? rgb[i] : nullptr;
    if (h->d_rgb > 0 && !tab[i]) return fail(PREGO_EINVAL, "rgb[%d] is NULL", i);
    tab[1 * MC + i] = (flow && h->d_flow > 0) ? flow[i] : nullptr;
    any_flow |= tab[1 * MC + i] != nullptr;
    tab[2 * MC + i] = out ? out[i] : nullptr;
    tab[3 * MC + i] = argmax ? argmax[i] : nullptr;
  }
  HIPCHK(hipMemcpyAsync(h->d_ptrs, tab.data(), tab.size() * sizeof(void*), hipMemcpyHostToDevice, s));
  const float* const* d_rgb_ptrs = h->d_rgb > 0 ? (const float* const*)(h->d_ptrs + 0 * MC) : nullptr;
  const float* const* d_flow_ptrs = any_flow ? (const float* const*)(h->d_ptrs + 1 * MC) : nullptr;
  float* const* d_out_ptrs = out ? (float* const*)(h->d_ptrs + 2 * MC) : nullptr;
  int* const* d_arg_ptrs = argmax ? (int* const*)(h->d_ptrs + 3 * MC) : nullptr;

  // workspace carve
  const bool with_flow = any_flow;
  const int kx = h->d_rgb + (with_flow ? h->d_flow : 0);      // K of the layer1 GEMM actually multiplied
  const int din = h->d_rgb + h->d_flow;
  const RowBytes rb = row_bytes(h, with_flow, flags);
  const int total_rows = h->h_rowoff[h->t_max];
  if (workspace_bytes < 6 * 256 + 128 * rb.total) return fail(PREGO_EWORKSPACE, "workspace %zu B is too small", workspace_bytes);
  long long cap_rows = (long long)((workspace_bytes - 6 * 256) / rb.total) / 128 * 128;
  if (cap_rows < n_clips) return fail(PREGO_EWORKSPACE, "workspace holds %lld rows, need >= %d (one time step)", cap_rows, n_clips);
  if ((flags & PREGO_FWD_KEEP) && cap_rows < total_rows)
    return fail(PREGO_EWORKSPACE, "PREGO_FWD_KEEP needs the whole batch resident: %d rows, workspace holds %lld", total_rows, cap_rows);
  char* wp = (char*)workspace;
  auto carve = [&](size_t bytes) { char* p = wp; wp += align_up(bytes, 256); return (void*)p; };
  void* X = carve((size_t)cap_rows * rb.x);
  float* Y = (float*)carve((size_t)cap_rows * rb.y);
  void* Eb = carve((size_t)cap_rows * rb.e);
  float* GI = (float*)carve((size_t)cap_rows * rb.gi);
  void* HR = carve((size_t)cap_rows * rb.hr);
  float* HRAW = rb.hraw ? (float*)carve((size_t)cap_rows * rb.hraw) : nullptr;

  // initial state (sorted order)
  const int H = h->hid, E = h->emb;
  if (h0) launch_permute_rows(h0, h->h_state, h->d_sorted, n_clips, H, 1, s);
  else HIPCHK(hipMemsetAsync(h->h_state, 0, (size_t)n_clips * H * 4, s));

  const int slots = (n_clips + h->G - 1) / h->G;
  const int nct = slots <= 16 ? 1 : slots <= 32 ? 2 : 4;

  int t0 = 0;
  while (t0 < h->t_max) {
    // largest t1 with rowoff[t1] - rowoff[t0] <= cap_rows
    const int base = h->h_rowoff[t0];
    int t1 = (int)(std::upper_bound(h->h_rowoff.begin() + t0, h->h_rowoff.end(), base + (int)std::min<long long>(cap_rows, total_rows)) -
                   h->h_rowoff.begin()) - 1;
    if (t1 <= t0) t1 = t0 + 1;
    if (t1 > h->t_max) t1 = h->t_max;
    const int rows = h->h_rowoff[t1] - base;

    EventPair* ev = ev_begin(h, 2, s);
    launch_pack_rows(h->bf16, d_rgb_ptrs, d_flow_ptrs, h->d_rowoff, h->d_sorted, h->t_max, base, rows, h->d_rgb,
                     with_flow ? h->d_flow : 0, X, s);
    ev_end(ev, s);
    if (h->timing) h->pack_bytes += (double)rows * (kx * 4.0 + rb.x);

    ev = ev_begin(h, 0, s);
    if (h->bf16) launch_gemm_bf16_nt(X, kx, h->w1, din, h->b1, Y, rows, E, kx, s);
    else launch_gemm_f32_nt((const float*)X, kx, (const float*)h->w1, din, h->b1, Y, rows, E, kx, s);
    ev_end(ev, s);
    launch_ln_relu(h->bf16, Y, h->ln_g, h->ln_b, rows, E, 1e-5f, Eb, nullptr, s);
    ev = ev_begin(h, 0, s);
    if (h->bf16) launch_gemm_bf16_nt(Eb, E, h->w_ih, E, h->bias2, GI, rows, 3 * H, E, s);
    else launch_gemm_f32_nt((const float*)Eb, E, (const float*)h->w_ih, E, h->bias2, GI, rows, 3 * H, E, s);
    ev_end(ev, s);
    if (h->timing) h->gemm_flop += 2.0 * rows * ((double)E * kx + 3.0 * H * E);

    GruArgs ga;
    ga.whh = h->w_hh; ga.b_hn = h->b_hn; ga.gi = GI; ga.h_relu_out = HR; ga.h_raw_out = HRAW;
    ga.h_state = h->h_state; ga.hx = h->hx; ga.flags = h->flags; ga.abort_word = h->abort_word;
    ga.rowoff = h->d_rowoff; ga.nact = h->d_nact; ga.t0 = t0; ga.t1 = t1; ga.row_base = base;
    ga.n_clips = n_clips; ga.G = h->G; ga.stamps = h->use_stamps ? h->stamps : nullptr;
    ga.sync = (getenv("PREGO_GRU_NO_LOCAL") == nullptr) ? h->flags : nullptr;   // flags[0..15] double as the rendezvous words
    ev = ev_begin(h, 1, s);
    if (launch_gru_recurrence(h->bf16, H, nct, ga, s)) return fail(PREGO_EINVAL, "recurrence: unsupported hid=%d nct=%d", H, nct);
    ev_end(ev, s);

    if (out || argmax) {
      if (launch_head_softmax(h->bf16, HR, h->w_c, h->b_c, h->d_rowoff, h->d_sorted, h->t_max, base, rows, H, h->ncls,
                              (flags & PREGO_FWD_SOFTMAX) ? 1 : 0, d_out_ptrs, d_arg_ptrs, s))
        return fail(PREGO_EINVAL, "head: unsupported num_classes %d", h->ncls);
    }
    t0 = t1;
  }
  if (h_last) launch_permute_rows(h->h_state, h_last, h->d_sorted, n_clips, H, 0, s);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

extern "C" int prego_miniroad_check(prego_miniroad* h, prego_stream_t stream) {
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  unsigned ab = 0;
  HIPCHK(hipMemcpy(&ab, h->abort_word, sizeof ab, hipMemcpyDeviceToHost));
  if (ab) {
    (void)hipMemset(h->abort_word, 0, sizeof ab);
    return fail(PREGO_ETIMEOUT, "GRU recurrence kernel timed out waiting for a producer workgroup (not all %d workgroups resident?)", h->G * h->P);
  }
  return PREGO_OK;
}

extern "C" int prego_miniroad_timing_enable(prego_miniroad* h, int enable) {
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  h->timing = enable != 0;
  h->ev_used = 0;
  h->gemm_flop = 0;
  h->pack_bytes = 0;
  return PREGO_OK;
}

extern "C" int prego_miniroad_timing_read(prego_miniroad* h, double* gemm_ms, int64_t* gemm_launches, double* gemm_flop,
                                          double* gru_ms, int64_t* gru_launches, double* pack_ms,
                                          int64_t* pack_launches, double* pack_bytes) {
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  double ms[3] = {0, 0, 0};
  int64_t n[3] = {0, 0, 0};
  for (size_t i = 0; i < h->ev_used; ++i) {
    HIPCHK(hipEventSynchronize(h->ev_pool[i].b));
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, h->ev_pool[i].a, h->ev_pool[i].b));
    ms[h->ev_kind[i]] += t;
    n[h->ev_kind[i]]++;
  }
  if (gemm_ms) *gemm_ms = ms[0];
  if (gemm_launches) *gemm_launches = n[0];
  if (gemm_flop) *gemm_flop = h->gemm_flop;
  if (gru_ms) *gru_ms = ms[1];
  if (gru_launches) *gru_launches = n[1];
  if (pack_ms) *pack_ms = ms[2];
  if (pack_launches) *pack_launches = n[2];
  if (pack_bytes) *pack_bytes = h->pack_bytes;
  h->ev_used = 0;
  h->gemm_flop = 0;
  h->pack_bytes = 0;
  return PREGO_OK;
}

// debug: per-phase shader-cycle sums of workgroup 0 / wave 0 of the recurrence kernel (PREGO_GRU_STAMPS=1):
// out[0..4] = poll, mfma, reduce+barrier, gates+publish, outputs; out[5] = poll retry rounds; out[6] = time steps
extern "C" int prego_miniroad_debug_stamps(prego_miniroad* h, unsigned long long* out8) {
  if (!h || !out8) return fail(PREGO_EINVAL, "NULL");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(out8, h->stamps, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(h->stamps, 0, 8 * sizeof(unsigned long long)));
  return PREGO_OK;
}
