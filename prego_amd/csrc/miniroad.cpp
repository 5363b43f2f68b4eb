// C ABI + host-side planner of the MiniROAD hot path (include/prego_amd.h).
// Host logic only: packing plan (sort clips by length, packed time-major rows, chunking), workspace
// carving, weight ingestion, and the per-chunk launch sequence
//   pack -> GEMM(layer1) -> LayerNorm+ReLU -> GEMM(W_ih) -> persistent GRU recurrence -> head+softmax+argmax.
#include "../../include/prego_amd.h"
#ifdef PREGO_DEBUG_ABI
#include "../../include/prego_amd_debug.h"
#endif
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

static thread_local std::string g_err;
// the handle whose entry point is running on this thread (set by HandleScope): errors are recorded in it as well, so that
// prego_miniroad_last_error(h) of one handle is never overwritten by another handle's failure
struct prego_miniroad;
static thread_local prego_miniroad* g_cur = nullptr;
static void note_handle_error(const char* msg);
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  note_handle_error(buf);
  return code;
}
// shared with vit.cpp
int prego_fail_(int code, const char* fmt, ...) {
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

const char* prego_tune_env(const char* name) {          // kernels.h
#ifdef PREGO_DEBUG_ABI
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct EventPair { hipEvent_t a, b; };

struct prego_miniroad {
  int d_rgb, d_flow, emb, hid, ncls, ncls_pad;
  bool bf16;                    // 16-bit MFMA operands (bf16, or IEEE fp16 when f16 is set as well); false = exact-fp32 MFMA
  bool f16 = false;             // PREGO_F16: the 16-bit operand / intermediate type is fp16 (inference entry points only)
  bool x2 = false;              // PREGO_F16X2: split fp16 operands (hi + lo, three products; csrc/common.h), fp32 intermediates; bf16 is false
  float* x2_scale = nullptr;    // device [3][2]: (scale, 1 / scale) of w1, w_ih, w_hh (powers of two chosen by set_weights)
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
  // split pass (round 6): W_ih / bias2 with their rows PERMUTED so that a recurrence lane's r / z / n pairs of a GI row are 12 adjacent
  // bytes: row (u / 2) * 6 + 2 * gate + u % 2 holds nn.GRU's row gate * H + u.  Built lazily in front of a split pass when the weights
  // have changed since (set_weights, the fused AdamW step); 16-bit handles of hidden_dim 1024 / one layer only
  void* w_ih_perm = nullptr; float* bias2_perm = nullptr; bool perm_stale = true;
  float* b_hn = nullptr;        // [H]
  void* w_c = nullptr;          // [ncls_pad][H] WT zero padded
  float* b_c = nullptr;         // [ncls_pad]
  bool have_weights = false;
  // nn.GRU(embedding_dim, hidden_dim, num_layers) with num_layers == 2 (rnn.py:32,38): layer 1's operands (gru.*_l1; its input is layer 0's
  // h_t, so weight_ih_l1 is [3H][H]).  Inference only; hidden state [layers][slots][H]
  int layers = 1;
  void* l2_w_ih = nullptr; void* l2_w_hh = nullptr; float* l2_bias2 = nullptr; float* l2_b_hn = nullptr; bool have_layer2 = false;
  // recurrence scratch
  void* hx = nullptr;           // [G][2][64][H] WT
  unsigned* flags = nullptr;    // [G*P] + abort word
  unsigned* abort_word = nullptr;
  float* h_state = nullptr;     // [max_clips][H]
  unsigned long long* stamps = nullptr;   // debug phase counters (PREGO_GRU_STAMPS=1)
  char* st_scratch = nullptr;   // streaming step: y [16][emb] f32 | e [16][emb] bf16 | gi [16][3H] f32 | gh [16][3H] f32
  bool use_stamps = false;
  // training
  float drop_p = 0.f;
  unsigned long long drop_seed = 0;
  int kept_kx = 0;              // K of layer1 actually multiplied by the last PREGO_FWD_KEEP forward
  int kept_rows = 0;
  // data-parallel training: events the NEXT backward records when a group of gradient tensors is final (prego_miniroad_backward_events),
  // so that the caller can start reducing that bucket on another stream while the rest of the backward runs
  float* g_l2[4] = {nullptr, nullptr, nullptr, nullptr};     // prego_miniroad_set_gru_layer_grads: dW_ih_l1, dW_hh_l1, db_ih_l1, db_hh_l1
  hipEvent_t bwd_ev[2] = {nullptr, nullptr};
  prego_bucket_fn bwd_cb = nullptr; void* bwd_cb_user = nullptr;     // prego_miniroad_backward_callback: called right behind each event record
  // plan cache
  std::vector<int32_t> plan_lens;
  std::vector<int> h_rowoff, h_nact, h_sorted;      // h_sorted: first clip of each slot (slot order)
  std::vector<int> h_seg_off, h_seg_clip, h_seg_start;
  std::vector<int> h_blkstep;    // step of packed row 32 b
  int* d_blkstep = nullptr; size_t cap_b = 0;
  int n_slots = 0;
  bool plan_single = true;       // one clip per slot (required for h0 / h_last / training)
  bool plan_want_single = false;
  int plan_host_row_bytes = 0;   // PCIe bytes per packed row the cached plan was costed with (0 = features in HBM)
  // feed events of the NEXT forward (prego_miniroad_set_feed_events): rows of steps < feed_upto[j] are valid once feed_ev[0..j] have fired
  std::vector<int> feed_upto; std::vector<hipEvent_t> feed_ev; size_t feed_pos = 0; int feed_row_bytes = 0;
  int t_max = 0;
  int* d_rowoff = nullptr; int* d_nact = nullptr; int* d_sorted = nullptr;
  int* d_seg_off = nullptr; int* d_seg_clip = nullptr; int* d_seg_start = nullptr;
  size_t cap_t = 0, cap_c = 0;
  // per-call pointer tables (device)
  void** d_ptrs = nullptr;      // [4][max_clips]
  // pinned host staging for the per-call tables (pointer table, plan arrays): the async H2D copies read it after the call
  // returns, so it is handle-owned and fenced by an event (never a stack or pageable buffer)
  char* pin = nullptr; size_t pin_bytes = 0; hipEvent_t pin_ev = nullptr; bool pin_busy = false;
  bool plan_dirty = false;      // host plan arrays changed, device copies pending
  bool no_local = false;        // PREGO_GRU_NO_LOCAL (read once at create): skip the XCD-local hand-off fast path
  bool no_mt = false;           // PREGO_GRU_NO_MT (read once at create): multi-tile steps on the classic kernel
  // feature streaming of chunk c+1 under the recurrence of chunk c: the pack kernel (22 registers, no LDS) fits beside a
  // recurrence workgroup on every CU, so it runs on a handle-owned side stream, forked from and joined to the caller's stream
  // by events (the caller still sees one in-order stream)
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; bool pack_prefetch = true;
  // layer1 GEMM of chunk c + 1 on the XCDs the (compacted) recurrence of chunk c does not hold (off: PREGO_NO_XCD_OVERLAP=1):
  // one tile counter per chunk, zeroed once per forward
  unsigned* tile_ctr = nullptr; bool xcd_overlap = false;
  // host mirror of the kernel's verified-placement word (rendezvous word 20: 1 = an earlier full-width launch found exactly 32 workgroups
  // on every XCD).  The device decides whether a launch is really compacted; the host only launches the layer1 worker beside a
  // recurrence it KNOWS will be compacted (advisor, round 3: with word 20 != 1 the recurrence ran full width while the persistent
  // worker competed for the same CUs).  -1 = not read yet: copied out behind the first full-width launch, read when that copy is done
  int placement = -1; unsigned* pin_place = nullptr; hipEvent_t ev_place = nullptr; bool place_pending = false;
  int prefetch_grid = 0;        // workgroup cap of the prefetching pack launch (0 = unthrottled)
  // split pass: recurrence on XCDs 0 .. split_r - 1 and the feed-forward of the whole pass on the others, two persistent
  // launches.  Their whole-call buffer [relu(h) rows of the pass | row map | counters] is the CALLER's resident buffer (res_buf,
  // prego_miniroad_set_resident): forward() allocates nothing and synchronises nothing for it (round 6)
  int split_r = 0; int plan_force_slots = 0;
  int split_env = -1;           // PREGO_SPLIT_PASS at create: -1 unset = decide per call (cost model), 0 = never, R = whenever a call is eligible
  double plan_cost_us = 0;      // recurrence cost estimate of the cached plan (kStepCost tables)
  std::vector<int32_t> split_seen_lens; int split_seen_key = -1, split_seen_r = 3; double split_seen_est_c = 0, split_seen_est_s = 0;   // the last estimates (same clips, same call shape)
  // the cost model is corrected by what passes of either kind actually took on THIS device (devices of one pool differ: a sustained
  // split pass runs its GEMM tiles 35 % slower on some, where it then loses to the chunked pass): measured / estimated, per kind
  hipEvent_t ev_meas[2] = {nullptr, nullptr}; bool meas_pending = false, meas_armed = false; int meas_mode = 0; double meas_est = 0;
  double ratio_chunked = 1.0, ratio_split = 1.0; bool have_ratio_chunked = false, have_ratio_split = false, split_warm = false;
  char* res_buf = nullptr; size_t res_bytes = 0;       // caller-owned (prego_miniroad_set_resident); NULL = per-chunk head, chunked pass
  const float* peer_guard = nullptr;                   // caller-owned device word (prego_miniroad_set_peer_guard); NULL = none
  // start handshake of a split pass (kernels.h: PassHandshake): the pinned word the two launches report their GO / FAIL decision in, the
  // pass counter, and the back-off after a FAIL (the call itself is re-run as a chunked pass: no call is ever lost)
  unsigned* pin_hs = nullptr; unsigned hs_seq = 0; int split_fails = 0; long long split_skip = 0; long long split_fallbacks = 0;
  int dbg_fault = 0;            // debug library only (prego_debug_split_fault): what the NEXT split pass does differently, one shot
  // chunked pass with the classifier ONCE behind the pass (as the split pass runs it): relu(h) of every packed row of the call stays
  // in the caller's resident buffer (capped at 24 GB) instead of one head launch per chunk
  hipEvent_t ev_split[4] = {nullptr, nullptr, nullptr, nullptr};   // timing of the two launches (timing_enable)
  double split_rec_ms = 0, split_ff_ms = 0; long long split_passes = 0, split_steps = 0; bool split_ev_pending = false;
  std::string err;              // last error of THIS handle (prego_miniroad_last_error)
  // timing
  bool timing = false;
  std::vector<EventPair> ev_pool;
  std::vector<int> ev_kind;     // 0 gemm (static launches), 1 gru, 2 pack, 3 overlapped layer1 worker
  size_t ev_used = 0;
  double gemm_flop = 0, pack_bytes = 0;
};

static void note_handle_error(const char* msg) { if (g_cur) g_cur->err = msg; }
struct HandleScope {
  explicit HandleScope(prego_miniroad* h) { g_cur = h; }
  ~HandleScope() { g_cur = nullptr; }
};
extern "C" const char* prego_miniroad_last_error(const prego_miniroad* h) { return h ? h->err.c_str() : "handle is NULL"; }

// split-operand recurrence: two clip tiles per group at most (the four-tile instantiation would spill: 2 x 96 weight registers)
static int max_slots_of(const prego_miniroad* h) { return h->G * 16 * (h->x2 ? 2 : gru_max_tiles()); }
#define PREGO_MAX_CLIPS 8192     // clips per call (continuous batching packs them into <= max_slots slots)
static int max_clips_of(const prego_miniroad*) { return PREGO_MAX_CLIPS; }

extern "C" int prego_abi_version(void) { return PREGO_ABI_VERSION; }
#ifdef PREGO_DEBUG_ABI
// unit-test hook (prego_debug_alloc_count): every device allocation and every stream / event wait this file makes is counted, so a test
// can hold "a hot call allocates nothing and waits for nothing" to zero
#include <atomic>
static std::atomic<long long> g_dbg_mallocs{0}, g_dbg_syncs{0};
#define hipMalloc(p, n) (++g_dbg_mallocs, (hipMalloc)(p, n))
#define hipStreamSynchronize(s) (++g_dbg_syncs, (hipStreamSynchronize)(s))
#define hipEventSynchronize(e) (++g_dbg_syncs, (hipEventSynchronize)(e))
void launch_debug_hog(int xcd_lo, int kind, int ms, const void* buf, void* wbuf, size_t bytes, float* sink, hipStream_t s);     // debug_hog.hip
extern "C" int prego_debug_hog(int kind, int xcd_lo, int ms, const void* read_buf, void* write_buf, size_t bytes, float* sink, prego_stream_t stream) {
  if (kind < 1 || kind > 3 || xcd_lo < 0 || xcd_lo > 7 || ms <= 0 || !sink || ((kind & 2) && (!read_buf || !write_buf || bytes < (1u << 20))))
    return fail(PREGO_EINVAL, "debug hog: bad arguments");
  launch_debug_hog(xcd_lo, kind, ms, read_buf, write_buf, bytes, sink, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}
extern "C" int prego_debug_alloc_count(int64_t* device_mallocs, int64_t* host_waits) {
  if (device_mallocs) *device_mallocs = g_dbg_mallocs.load();
  if (host_waits) *host_waits = g_dbg_syncs.load();
  return PREGO_OK;
}
#endif

extern "C" const char* prego_last_error(void) { return g_err.c_str(); }

extern "C" int prego_miniroad_create(prego_miniroad** out, int d_rgb, int d_flow, int emb, int hid, int n_classes,
                                     int compute_dtype) {
  return prego_miniroad_create_layers(out, d_rgb, d_flow, emb, hid, n_classes, 1, compute_dtype);
}
extern "C" int prego_miniroad_create_layers(prego_miniroad** out, int d_rgb, int d_flow, int emb, int hid, int n_classes, int num_layers,
                                            int compute_dtype) {
  if (!out) return fail(PREGO_EINVAL, "out is NULL");
  *out = nullptr;
  if (compute_dtype != PREGO_F32 && compute_dtype != PREGO_BF16 && compute_dtype != PREGO_F16 && compute_dtype != PREGO_F16X2)
    return fail(PREGO_EINVAL, "compute_dtype %d", compute_dtype);
  // The recurrence keeps a workgroup's slice of W_hh in registers (3 gates x 16 or 32 rows x H): what fits decides.  16-bit operands:
  // 512, 1024, 2048; exact-fp32 operands: 512, 1024 (a 16-row slice of H = 2048 is 384 registers per lane); split operands: 1024
  {
    const bool op16 = compute_dtype == PREGO_BF16 || compute_dtype == PREGO_F16;
    const bool ok = compute_dtype == PREGO_F16X2 ? hid == 1024 : gru_hidden_supported(op16, hid);
    if (!ok) return fail(PREGO_EINVAL, "hidden_dim %d unsupported with compute_dtype %d: 512 / 1024 / 2048 with 16-bit operands, 512 / 1024 with fp32 "
                                       "operands, 1024 with fp16x2 (the recurrence keeps its W_hh slice in registers)", hid, compute_dtype);
  }
  if (num_layers < 1 || num_layers > 2) return fail(PREGO_EINVAL, "num_layers %d: 1 or 2", num_layers);
  if (num_layers == 2 && compute_dtype == PREGO_F16X2)
    return fail(PREGO_EINVAL, "num_layers 2 with fp16x2 operands: the split-operand recurrence hands fp32 relu(h) to the classifier only (use fp32)");
  if (emb <= 0 || emb % 512 || emb > 4096) return fail(PREGO_EINVAL, "embedding_dim %d must be a multiple of 512, <= 4096", emb);
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
  h->bf16 = compute_dtype == PREGO_BF16 || compute_dtype == PREGO_F16;
  h->f16 = compute_dtype == PREGO_F16;
  h->x2 = compute_dtype == PREGO_F16X2;      // operand storage 4 bytes per element ([hi | lo] fp16), P = 64, G = 4 like fp32 operands
  h->n_cu = prop.multiProcessorCount;
  h->layers = num_layers;
  h->P = h->x2 ? hid / 16 : gru_group_size(h->bf16, hid);      // 1024: 32 (16-bit) / 64 workgroups per group
  h->G = std::min(h->bf16 ? 8 : 4, h->n_cu / h->P);
  if (h->G < 1) { delete h; return fail(PREGO_EINVAL, "device has %d CUs, the recurrence needs >= %d", prop.multiProcessorCount, h->bf16 ? 32 : 64); }
  const size_t es = h->bf16 ? 2 : 4;
  const int din = d_rgb + d_flow, H = hid;
  hipError_t e = hipSuccess;
  auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
  A(&h->w1, (size_t)emb * din * es); A((void**)&h->b1, emb * 4); A((void**)&h->ln_g, emb * 4); A((void**)&h->ln_b, emb * 4);
  A(&h->w_ih, (size_t)3 * H * emb * es); A(&h->w_hh, (size_t)3 * H * H * es);
  A((void**)&h->bias2, 3 * H * 4); A((void**)&h->b_hn, H * 4);
  if (h->bf16 && H == 1024 && h->layers == 1) { A(&h->w_ih_perm, (size_t)3 * H * emb * es); A((void**)&h->bias2_perm, 3 * H * 4); }
  A(&h->w_c, (size_t)h->ncls_pad * H * es); A((void**)&h->b_c, h->ncls_pad * 4);
  A(&h->hx, h->x2 ? gru_x2_hx_bytes(H, h->G) : gru_hx_bytes(h->bf16, H, h->G));
  if (h->x2) A((void**)&h->x2_scale, 6 * sizeof(float));
  A((void**)&h->flags, ((size_t)h->G * h->P + 16) * sizeof(unsigned));
  A((void**)&h->h_state, (size_t)num_layers * max_slots_of(h) * H * 4);
  if (num_layers == 2) {
    A(&h->l2_w_ih, (size_t)3 * H * H * es); A(&h->l2_w_hh, (size_t)3 * H * H * es);
    A((void**)&h->l2_bias2, 3 * H * 4); A((void**)&h->l2_b_hn, H * 4);
  }
  A((void**)&h->d_ptrs, (size_t)4 * max_clips_of(h) * sizeof(void*));
  // plan tables pre-sized here so that forward() allocates nothing: 131 072 steps (a 72-minute clip at 30 frames/s; the longest
  // Epic-tent-O video has 31 114 frames) and max_clips clips; only a longer clip than that makes forward() grow them
  h->cap_t = (size_t)131072 + 1;
  h->cap_c = (size_t)max_clips_of(h) + 64;
  A((void**)&h->d_rowoff, h->cap_t * 4); A((void**)&h->d_nact, h->cap_t * 4);
  A((void**)&h->d_sorted, h->cap_c * 4); A((void**)&h->d_seg_off, (h->cap_c + 1) * 4);
  A((void**)&h->d_seg_clip, h->cap_c * 4); A((void**)&h->d_seg_start, h->cap_c * 4);
  h->cap_b = (size_t)1 << 19;                    // 16.7 M packed rows per call before the table has to grow
  A((void**)&h->d_blkstep, h->cap_b * 4);
  h->pin_bytes = (size_t)4 * max_clips_of(h) * sizeof(void*) + 2 * h->cap_t * 4 + 4 * (h->cap_c + 1) * 4 + h->cap_b * 4 + 1024;
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->pin, h->pin_bytes, hipHostMallocDefault);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->pin_ev, hipEventDisableTiming);
  h->no_local = getenv("PREGO_GRU_NO_LOCAL") != nullptr;
  h->no_mt = getenv("PREGO_GRU_NO_MT") != nullptr;
  h->pack_prefetch = prego_tune_env("PREGO_NO_PACK_PREFETCH") == nullptr;
  // the pack beside the recurrence slows its L2 hand-off; capped at 512 workgroups it still ends inside a 49 152-row launch and
  // costs the pass 0.9 ms less than unthrottled (sweep: scripts/probes/env_sweep.sh, 128: +10 ms, 256: +1, 512: -0.9, 1024: 0)
  h->prefetch_grid = 512;
  if (const char* pg = prego_tune_env("PREGO_PACK_PREFETCH_GRID")) h->prefetch_grid = atoi(pg);
  // The side stream must run BESIDE the caller's stream.  HIP maps streams onto a handful of hardware queues in creation order and
  // two streams on one queue execute in submission order (round 4: an eval loop whose copy stream shared the compute stream's queue
  // lost all of its overlap), and a queue has one priority: a LOW-priority side stream never shares the queue of a normal-priority
  // caller, and its pack / layer1 worker yield to the recurrence where they compete.  PREGO_SIDE_PRIO=0: the plain stream (A/B).
  if (e == hipSuccess) {
    int lo = 0, hi = 0;
    static const bool plain = prego_tune_env("PREGO_SIDE_PRIO") != nullptr && atoi(prego_tune_env("PREGO_SIDE_PRIO")) == 0;
    if (!plain && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
      e = hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, lo);       // lo = numerically greatest = least priority
    else
      e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
  A((void**)&h->stamps, 8 * sizeof(unsigned long long));
  A((void**)&h->tile_ctr, 4096 * sizeof(unsigned));
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->pin_place, 64, hipHostMallocDefault);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_place, hipEventDisableTiming);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->pin_hs, 64, hipHostMallocDefault);
  if (e == hipSuccess) *(volatile unsigned*)h->pin_hs = 0u;
  if (e == hipSuccess) e = hipEventCreate(&h->ev_meas[0]);
  if (e == hipSuccess) e = hipEventCreate(&h->ev_meas[1]);
  h->xcd_overlap = getenv("PREGO_NO_XCD_OVERLAP") == nullptr;       // A/B knob: PREGO_NO_XCD_OVERLAP=1 = the serial pass of round 2
  if (const char* sp = getenv("PREGO_SPLIT_PASS")) h->split_env = atoi(sp);
  A((void**)&h->st_scratch, (size_t)16 * ((size_t)emb * 6 + (size_t)3 * H * 8));
  if (e == hipSuccess) e = hipMemset(h->stamps, 0, 8 * sizeof(unsigned long long));
  h->use_stamps = prego_tune_env("PREGO_GRU_STAMPS") != nullptr;
  if (e == hipSuccess) e = hipMemset(h->hx, 0, h->x2 ? gru_x2_hx_bytes(H, h->G) : gru_hx_bytes(h->bf16, H, h->G));
  if (e == hipSuccess) e = hipMemset(h->flags, 0, ((size_t)h->G * h->P + 16) * sizeof(unsigned));
  if (e != hipSuccess) { prego_miniroad_destroy(h); return fail(PREGO_EHIP, "hipMalloc: %s", hipGetErrorString(e)); }
  h->abort_word = h->flags + (size_t)h->G * h->P;
  *out = h;
  return PREGO_OK;
}

// teardown calls must not leave an error code behind for whatever HIP call the process makes next (a swallowed hipErrorInvalidValue here
// surfaced in an unrelated torch kernel launch of the NEXT test): every failure is consumed, and named in the debug library
#ifdef PREGO_DEBUG_ABI
#define PREGO_TEARDOWN(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "prego_miniroad_destroy: %s -> %s\n", #x, hipGetErrorName(e_)); (void)hipGetLastError(); } } while (0)
#else
#define PREGO_TEARDOWN(x) do { if ((x) != hipSuccess) (void)hipGetLastError(); } while (0)
#endif
extern "C" void prego_miniroad_destroy(prego_miniroad* h) {
  if (!h) return;
  if (h->side) { PREGO_TEARDOWN(hipStreamSynchronize(h->side)); PREGO_TEARDOWN(hipStreamDestroy(h->side)); }
  void* ptrs[] = {h->w1, h->b1, h->ln_g, h->ln_b, h->w_ih, h->w_hh, h->bias2, h->b_hn, h->w_c, h->b_c, h->hx,
                  h->flags, h->h_state, h->stamps, h->tile_ctr, h->d_rowoff, h->d_nact, h->d_sorted, h->d_seg_off, h->d_seg_clip,
                  h->d_seg_start, h->d_ptrs, h->d_blkstep, h->st_scratch, h->x2_scale, h->l2_w_ih, h->l2_w_hh, h->l2_bias2, h->l2_b_hn, h->w_ih_perm, h->bias2_perm};
  for (size_t i = 0; i < sizeof ptrs / sizeof ptrs[0]; ++i)
    if (ptrs[i]) {
#ifdef PREGO_DEBUG_ABI
      const hipError_t e_ = hipFree(ptrs[i]);
      if (e_ != hipSuccess) { fprintf(stderr, "prego_miniroad_destroy: hipFree(ptrs[%zu] = %p) -> %s\n", i, ptrs[i], hipGetErrorName(e_)); (void)hipGetLastError(); }
#else
      PREGO_TEARDOWN(hipFree(ptrs[i]));
#endif
    }
  for (auto& ev : h->ev_pool) { PREGO_TEARDOWN(hipEventDestroy(ev.a)); PREGO_TEARDOWN(hipEventDestroy(ev.b)); }
  if (h->pin_ev) { if (h->pin_busy) PREGO_TEARDOWN(hipEventSynchronize(h->pin_ev)); PREGO_TEARDOWN(hipEventDestroy(h->pin_ev)); }
  if (h->pin) PREGO_TEARDOWN(hipHostFree(h->pin));
  if (h->ev_fork) PREGO_TEARDOWN(hipEventDestroy(h->ev_fork));
  if (h->ev_join) PREGO_TEARDOWN(hipEventDestroy(h->ev_join));
  if (h->ev_place) PREGO_TEARDOWN(hipEventDestroy(h->ev_place));
  for (hipEvent_t ev : h->ev_meas) if (ev) PREGO_TEARDOWN(hipEventDestroy(ev));
  for (hipEvent_t ev : h->ev_split) if (ev) PREGO_TEARDOWN(hipEventDestroy(ev));
  if (h->pin_place) PREGO_TEARDOWN(hipHostFree(h->pin_place));
  if (h->pin_hs) PREGO_TEARDOWN(hipHostFree(h->pin_hs));
  delete h;
}

extern "C" int prego_miniroad_max_clips(const prego_miniroad* h) { return h ? max_clips_of(h) : 0; }

extern "C" int prego_miniroad_set_weights(prego_miniroad* h, const float* layer1_w, const float* layer1_b,
                                          const float* ln_w, const float* ln_b, const float* w_ih, const float* w_hh,
                                          const float* b_ih, const float* b_hh, const float* fc_w, const float* fc_b,
                                          prego_stream_t stream) {
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  if (!layer1_w || !layer1_b || !ln_w || !ln_b || !w_ih || !w_hh || !b_ih || !b_hh || !fc_w || !fc_b)
    return fail(PREGO_EINVAL, "set_weights: NULL tensor");
  hipStream_t s = (hipStream_t)stream;
  const int din = h->d_rgb + h->d_flow, E = h->emb, H = h->hid;
  if (h->x2) {                   // split rows [cols hi | cols lo] of W * 2^k, k per tensor (common.h)
    launch_x2_weight_split(layer1_w, E, din, h->w1, h->x2_scale + 0, s);
    launch_x2_weight_split(w_ih, 3 * H, E, h->w_ih, h->x2_scale + 2, s);
    launch_x2_weight_split(w_hh, 3 * H, H, h->w_hh, h->x2_scale + 4, s);
  } else {
    launch_pad_convert(h->bf16, layer1_w, E, din, din, h->w1, E, din, s, h->f16);
    launch_pad_convert(h->bf16, w_ih, 3 * H, E, E, h->w_ih, 3 * H, E, s, h->f16);
    launch_pad_convert(h->bf16, w_hh, 3 * H, H, H, h->w_hh, 3 * H, H, s, h->f16);
  }
  launch_pad_convert(h->bf16, fc_w, h->ncls, H, H, h->w_c, h->ncls_pad, H, s, h->f16);
  launch_pad_convert(false, fc_b, 1, h->ncls, h->ncls, h->b_c, 1, h->ncls_pad, s);
  HIPCHK(hipMemcpyAsync(h->b1, layer1_b, E * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(h->ln_g, ln_w, E * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(h->ln_b, ln_b, E * 4, hipMemcpyDeviceToDevice, s));
  launch_add_vec(b_ih, b_hh, h->bias2, 3 * H, 2 * H, s);   // r,z rows: b_ih + b_hh ; n rows: b_ih
  h->perm_stale = true;
  HIPCHK(hipMemcpyAsync(h->b_hn, b_hh + 2 * H, H * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipGetLastError());
  h->have_weights = true;
  return PREGO_OK;
}

extern "C" int prego_miniroad_set_gru_layer(prego_miniroad* h, int layer, const float* w_ih, const float* w_hh, const float* b_ih,
                                            const float* b_hh, prego_stream_t stream) {
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  if (layer != 1 || h->layers != 2) return fail(PREGO_EINVAL, "set_gru_layer: layer %d of a %d-layer handle (layer 0 comes with set_weights)", layer, h->layers);
  if (!w_ih || !w_hh || !b_ih || !b_hh) return fail(PREGO_EINVAL, "set_gru_layer: NULL tensor");
  hipStream_t s = (hipStream_t)stream;
  const int H = h->hid;
  launch_pad_convert(h->bf16, w_ih, 3 * H, H, H, h->l2_w_ih, 3 * H, H, s, h->f16);
  launch_pad_convert(h->bf16, w_hh, 3 * H, H, H, h->l2_w_hh, 3 * H, H, s, h->f16);
  launch_add_vec(b_ih, b_hh, h->l2_bias2, 3 * H, 2 * H, s);
  HIPCHK(hipMemcpyAsync(h->l2_b_hn, b_hh + 2 * H, H * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipGetLastError());
  h->have_layer2 = true;
  return PREGO_OK;
}

// ---- plan -------------------------------------------------------------------------------------
// recurrence cost per time step (us) by live 16-clip tiles per group, measured (scripts/probes/slot_sweep.sh, round 2: 512 clips x
// 512 frames forced into 128 / 256 / 512 slots = 2.01 / 4.03 / 8.55 us per step): the tiles of a step run one after the other,
// so the cost is linear in the tile count and the fewest slots that cover the clips win unless a longer slot chain dominates
// Round 3: two or more tiles run on the software-pipelined kernel (gru_recurrence_mt_kernel): 2.0 / 3.84 / 7.41 us per step for
// 1 / 2 / 4 tiles (scripts/probes/mt_ab2.sh; the classic kernel: 2.0 / 3.99 / 8.47 on the same device).
static const double kStepCost[5] = {0.0, 2.0, 3.84, 5.7, 7.41};
// Round 4, the 64-workgroup groups (G = 4): split fp16 operands 3.14 / 4.64 us for 1 / 2 tiles (two tiles at most), exact-fp32 operands
// 6.89 / 9.25 (bench workload forced into 64 / 128 slots, PREGO_PLAN_SLOTS; three and four tiles extrapolated): a second tile costs
// less than the first there (its gather rides under the first tile's MFMAs), so equal-length batches prefer more slots than the
// 16-bit table would choose
static const double kStepCostX2[5] = {0.0, 3.14, 4.64, 1e9, 1e9};
static const double kStepCostF32[5] = {0.0, 6.89, 9.25, 11.6, 14.0};

// Slot schedule.  want_single: one clip per slot (needed when the caller passes h0 / h_last or keeps activations for
// backward); otherwise the clips are packed longest-first into the number of slots (128 / 256 / 512 for bf16) that
// minimises the estimated recurrence time: sequential steps = max(longest clip, frames / slots).
// host_row_bytes > 0: the features live in pinned HOST memory and every packed row costs that many bytes over PCIe (PREGO_FWD_HOSTFEAT):
// a step can then be bound by the link - live slots x row bytes at ~50 GB/s - instead of by the recurrence, and the slot count that
// minimises the pass is the one that keeps the link evenly busy for the whole run (about frames / longest clip slots: every slot
// alive to the end), not the one that minimises the number of steps.
static int build_plan(prego_miniroad* h, int n, const int32_t* lens, bool want_single, int host_row_bytes = 0, int slots_arg = 0) {
  if ((int)h->plan_lens.size() == n && std::equal(lens, lens + n, h->plan_lens.begin()) && h->plan_want_single == want_single &&
      h->plan_host_row_bytes == host_row_bytes && h->plan_force_slots == slots_arg)
    return PREGO_OK;
  long long total = 0;
  int lmax = 0;
  for (int i = 0; i < n; ++i) {
    if (lens[i] <= 0) return fail(PREGO_EINVAL, "clip %d has %d frames", i, lens[i]);
    lmax = std::max(lmax, lens[i]);
    total += lens[i];
  }
  if (total >= (1ll << 31)) return fail(PREGO_EINVAL, "more than 2^31 frames in one call");
  const int per_layer = h->G * 16, max_slots = max_slots_of(h);
  if (want_single && n > max_slots) return fail(PREGO_EINVAL, "%d clips > %d per call when h0/h_last/training is used", n, max_slots);
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lens[a] > lens[b]; });

  // candidate slot counts; LPT packing; exact cost = sum over steps of cost(live tiles)
  struct Cand { int S; std::vector<std::vector<int>> bins; std::vector<long long> load; double cost; };
  auto pack = [&](int S) {
    Cand c; c.S = S; c.bins.assign(S, {}); c.load.assign(S, 0);
    // min-heap on (load, slot)
    std::vector<std::pair<long long, int>> heap;
    for (int i = 0; i < S; ++i) heap.push_back({0, i});
    auto cmp = [](const std::pair<long long, int>& a, const std::pair<long long, int>& b) { return a > b; };
    std::make_heap(heap.begin(), heap.end(), cmp);
    for (int idx : order) {
      std::pop_heap(heap.begin(), heap.end(), cmp);
      auto top = heap.back(); heap.pop_back();
      c.bins[top.second].push_back(idx);
      c.load[top.second] += lens[idx];
      top.first += lens[idx];
      heap.push_back(top); std::push_heap(heap.begin(), heap.end(), cmp);
    }
    std::vector<long long> sorted_load = c.load;
    std::sort(sorted_load.begin(), sorted_load.end(), std::greater<long long>());
    const double* tab = h->x2 ? kStepCostX2 : h->bf16 ? kStepCost : kStepCostF32;
    double cost = 0; long long prev = 0;
    if (host_row_bytes <= 0) {
      const int layers = (S + per_layer - 1) / per_layer;
      for (int k = layers - 1; k >= 0; --k) {            // layer k lives as long as its most loaded slot = sorted_load[k*per_layer]
        const long long life = sorted_load[(size_t)k * per_layer];
        cost += (double)(life - prev) * tab[std::min(4, k + 1)];
        prev = life;
      }
    } else {
      // steps (sorted_load[k + 1], sorted_load[k]] have k + 1 live slots: the step costs what the slower of the recurrence and the
      // link needs (us; 50 GB/s = 50 000 bytes per us: what a throttled pack kernel pulls from pinned memory, h2d copies reach 57)
      for (int k = S - 1; k >= 0; --k) {
        const long long life = sorted_load[(size_t)k];
        if (life <= prev) continue;
        const double rec = tab[std::min(4, k / per_layer + 1)];
        const double link = (double)(k + 1) * host_row_bytes / 50000.0;
        cost += (double)(life - prev) * std::max(rec, link);
        prev = life;
      }
    }
    c.cost = cost;
    return c;
  };
  Cand best;
  static const int force_slots = prego_tune_env("PREGO_PLAN_SLOTS") ? atoi(prego_tune_env("PREGO_PLAN_SLOTS")) : 0;   // debug / calibration of kStepCost
  if (slots_arg > 0) {                                                    // split pass: one tile on each of its groups
    best = pack(std::min(n, slots_arg));
    // every slot is alive to the end of a split pass, so the pass takes as long as the most loaded slot: LPT leaves it a few percent above
    // frames / slots (bench workload: 50 126 vs 48 045 steps).  Local search on the LPT result: move or swap clips between the most loaded
    // slot and any other while that lowers the larger of the two loads
    const int S2 = best.S;
    for (int iter = 0; iter < 4096; ++iter) {
      int A = 0;
      for (int i = 1; i < S2; ++i) if (best.load[i] > best.load[A]) A = i;
      long long best_gain = 0; int bB = -1, ba = -1, bb = -1;
      for (int B = 0; B < S2; ++B) {
        if (B == A) continue;
        const long long la = best.load[A], lb = best.load[B];
        for (size_t ia = 0; ia < best.bins[A].size(); ++ia) {
          const long long a = lens[best.bins[A][ia]];
          if (best.bins[A].size() > 1) {                                  // move a: A -> B
            const long long gain = la - std::max(la - a, lb + a);
            if (gain > best_gain) { best_gain = gain; bB = B; ba = (int)ia; bb = -1; }
          }
          for (size_t ib = 0; ib < best.bins[B].size(); ++ib) {           // swap a <-> b
            const long long b = lens[best.bins[B][ib]];
            if (b >= a) continue;
            const long long gain = la - std::max(la - a + b, lb - b + a);
            if (gain > best_gain) { best_gain = gain; bB = B; ba = (int)ia; bb = (int)ib; }
          }
        }
      }
      if (bB < 0) break;
      const int ca = best.bins[A][ba];
      if (bb < 0) {
        best.bins[A].erase(best.bins[A].begin() + ba); best.bins[bB].push_back(ca);
        best.load[A] -= lens[ca]; best.load[bB] += lens[ca];
      } else {
        const int cb = best.bins[bB][bb];
        best.bins[A][ba] = cb; best.bins[bB][bb] = ca;
        best.load[A] += lens[cb] - lens[ca]; best.load[bB] += lens[ca] - lens[cb];
      }
    }
    long long mx = 0;
    for (int i = 0; i < S2; ++i) mx = std::max(mx, best.load[i]);
    best.cost = (double)mx * (h->x2 ? kStepCostX2 : h->bf16 ? kStepCost : kStepCostF32)[1];
  }
  else if (want_single || (n <= per_layer && host_row_bytes <= 0)) best = pack(n);
  else if (force_slots > 0) best = pack(std::min(n, std::min(force_slots, max_slots)));
  else {
    best = pack(std::min(n, per_layer));
    for (int S = 2 * per_layer; S <= max_slots && n > per_layer; S *= 2) {
      Cand c = pack(std::min(n, S));
      if (c.cost < best.cost) best = std::move(c);
      if (S >= n) break;
    }
    if (host_row_bytes > 0)                              // link-bound candidates: fewer slots than one tile layer, in steps of 8
      for (int S = 8; S < std::min(n, per_layer); S += 8) {
        Cand c = pack(S);
        if (c.cost < best.cost) best = std::move(c);
      }
  }
  const int S = best.S;
  std::vector<int> slot_order(S);
  std::iota(slot_order.begin(), slot_order.end(), 0);
  std::stable_sort(slot_order.begin(), slot_order.end(), [&](int a, int b) { return best.load[a] > best.load[b]; });
  const int smax = (int)best.load[slot_order[0]];
  h->h_seg_off.assign(S + 1, 0); h->h_seg_clip.clear(); h->h_seg_start.clear(); h->h_sorted.assign(S, 0);
  std::vector<int> cnt((size_t)smax + 1, 0);
  bool single = true;
  for (int i = 0; i < S; ++i) {
    const auto& bin = best.bins[slot_order[i]];
    int start = 0;
    for (int idx : bin) { h->h_seg_clip.push_back(idx); h->h_seg_start.push_back(start); start += lens[idx]; }
    h->h_seg_off[i + 1] = (int)h->h_seg_clip.size();
    h->h_sorted[i] = bin.empty() ? 0 : bin[0];
    single = single && bin.size() == 1;
    cnt[start]++;
  }
  h->h_nact.assign(smax, 0);
  int alive = 0;
  for (int t = smax; t >= 1; --t) { alive += cnt[t]; h->h_nact[t - 1] = alive; }
  h->h_rowoff.assign((size_t)smax + 1, 0);
  for (int t = 0; t < smax; ++t) h->h_rowoff[t + 1] = h->h_rowoff[t] + h->h_nact[t];
  {                                  // step of every 32nd packed row (the head kernel's row -> step lookup starts there)
    const int total = h->h_rowoff[smax];
    h->h_blkstep.assign((size_t)(total + 31) / 32, 0);
    int st = 0;
    for (size_t b = 0; b < h->h_blkstep.size(); ++b) {
      const int row = (int)b * 32;
      while (st + 1 < smax && h->h_rowoff[st + 1] <= row) ++st;
      h->h_blkstep[b] = st;
    }
  }
  h->plan_dirty = true;              // device copies are staged by the caller (stage_tables)
  h->t_max = smax;
  h->n_slots = S;
  h->plan_single = single;
  h->plan_want_single = want_single;
  h->plan_host_row_bytes = host_row_bytes;
  h->plan_force_slots = slots_arg;
  h->plan_cost_us = best.cost;
  h->plan_lens.assign(lens, lens + n);
  return PREGO_OK;
}

static SlotPlan device_plan(const prego_miniroad* h) {
  SlotPlan p;
  p.rowoff = h->d_rowoff; p.nact = h->d_nact; p.seg_off = h->d_seg_off; p.seg_clip = h->d_seg_clip; p.seg_start = h->d_seg_start;
  p.blk_step = h->d_blkstep;
  p.s_max = h->t_max; p.n_slots = h->n_slots;
  return p;
}

// Stage the per-call pointer table (and, when the plan changed, the plan arrays) through the handle's pinned buffer.
// `tab4` = 4 * max_clips pointers.  The previous call's copies are fenced by pin_ev before the buffer is rewritten.
// The host blocks here until the PREVIOUS call's table copies have left the pinned buffer: CPU run-ahead is one call deep
// (a second forward() can be enqueued while the first runs, a third waits for the first's H2D copies, not for its kernels).
static int stage_tables(prego_miniroad* h, const void* const* tab4, size_t tab_count, hipStream_t s) {
  if (h->pin_busy) { HIPCHK(hipEventSynchronize(h->pin_ev)); h->pin_busy = false; }
  const size_t smax = (size_t)h->t_max, S = (size_t)h->n_slots, n = h->h_seg_clip.size();
  const size_t nb = h->h_blkstep.size();
  if (h->plan_dirty && (smax + 1 > h->cap_t || n + 1 > h->cap_c || nb > h->cap_b)) {
    // a clip longer than the tables reserved at create (or more clips): grow once, outside the steady state
    HIPCHK(hipStreamSynchronize(s));
    if (smax + 1 > h->cap_t) {
      (void)hipFree(h->d_rowoff); (void)hipFree(h->d_nact);
      h->cap_t = smax + 1 + 4096;
      HIPCHK(hipMalloc((void**)&h->d_rowoff, h->cap_t * 4)); HIPCHK(hipMalloc((void**)&h->d_nact, h->cap_t * 4));
    }
    if (n + 1 > h->cap_c) {
      for (int** p : {&h->d_sorted, &h->d_seg_off, &h->d_seg_clip, &h->d_seg_start}) { (void)hipFree(*p); *p = nullptr; }
      h->cap_c = n + 64;
      HIPCHK(hipMalloc((void**)&h->d_sorted, h->cap_c * 4)); HIPCHK(hipMalloc((void**)&h->d_seg_off, (h->cap_c + 1) * 4));
      HIPCHK(hipMalloc((void**)&h->d_seg_clip, h->cap_c * 4)); HIPCHK(hipMalloc((void**)&h->d_seg_start, h->cap_c * 4));
    }
    if (nb > h->cap_b) {
      (void)hipFree(h->d_blkstep);
      h->cap_b = nb + 4096;
      HIPCHK(hipMalloc((void**)&h->d_blkstep, h->cap_b * 4));
    }
    (void)hipHostFree(h->pin);
    h->pin = nullptr;
    h->pin_bytes = (size_t)4 * max_clips_of(h) * sizeof(void*) + 2 * h->cap_t * 4 + 4 * (h->cap_c + 1) * 4 + h->cap_b * 4 + 1024;
    HIPCHK(hipHostMalloc((void**)&h->pin, h->pin_bytes, hipHostMallocDefault));
  }
  char* p = h->pin;
  auto put = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
    std::memcpy(p, src, bytes);
    const hipError_t e = hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, s);
    p += (bytes + 15) / 16 * 16;
    return e;
  };
  HIPCHK(put(h->d_ptrs, tab4, tab_count * sizeof(void*)));
  if (h->plan_dirty) {
    HIPCHK(put(h->d_rowoff, h->h_rowoff.data(), (smax + 1) * 4));
    HIPCHK(put(h->d_nact, h->h_nact.data(), smax * 4));
    HIPCHK(put(h->d_sorted, h->h_sorted.data(), S * 4));
    HIPCHK(put(h->d_seg_off, h->h_seg_off.data(), (S + 1) * 4));
    HIPCHK(put(h->d_seg_clip, h->h_seg_clip.data(), n * 4));
    HIPCHK(put(h->d_seg_start, h->h_seg_start.data(), n * 4));
    HIPCHK(put(h->d_blkstep, h->h_blkstep.data(), nb * 4));
    h->plan_dirty = false;
  }
  HIPCHK(hipEventRecord(h->pin_ev, s));
  h->pin_busy = true;
  return PREGO_OK;
}

// ---- link-fed inference: features arrive over the host link WHILE the forward runs ---------------------------------------------
// The eval loop's H2D copy of a batch (57 GB/s) and its forward (bound by the longest video's recurrence) are each ~60 ms for the
// bench's 60 videos; run one after the other they are the 46 % of the PCIe floor the round-3 verdict measured.  The caller copies the
// features in the order the packed pipeline NEEDS them (piece (clip, frames [a, b)) is needed at step start_step[clip] + a) and
// records events along the way; the pack of each chunk waits for the events that cover its steps.  plan_starts reports the schedule
// (costed for a link-bound feed: fewer slots than clips, so that rows are needed at the rate the link delivers them).
extern "C" int prego_miniroad_plan_starts(prego_miniroad* h, int n_clips, const int32_t* lens, int link_row_bytes, int32_t* start_step,
                                          int32_t* n_steps) {
  HandleScope scope_(h);
  if (!h || !lens || !start_step || n_clips <= 0) return fail(PREGO_EINVAL, "plan_starts: bad arguments");
  if (n_clips > max_clips_of(h)) return fail(PREGO_EINVAL, "%d clips > max_clips %d per call", n_clips, max_clips_of(h));
  const int rc = build_plan(h, n_clips, lens, false, link_row_bytes > 0 ? link_row_bytes : 0);
  if (rc) return rc;
  for (size_t k = 0; k < h->h_seg_clip.size(); ++k) start_step[h->h_seg_clip[k]] = h->h_seg_start[k];
  if (n_steps) *n_steps = h->t_max;
  return PREGO_OK;
}

extern "C" int prego_miniroad_set_feed_events(prego_miniroad* h, int n_events, const int32_t* upto_step, void* const* events,
                                              int link_row_bytes) {
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  h->feed_ev.clear(); h->feed_upto.clear(); h->feed_pos = 0; h->feed_row_bytes = 0;
  if (n_events == 0) return PREGO_OK;
  if (n_events < 0 || !upto_step || !events || link_row_bytes <= 0) return fail(PREGO_EINVAL, "set_feed_events: bad arguments");
  for (int j = 0; j < n_events; ++j) {
    if (!events[j] || (j > 0 && upto_step[j] < upto_step[j - 1])) { h->feed_ev.clear(); h->feed_upto.clear(); return fail(PREGO_EINVAL, "set_feed_events: event %d", j); }
    h->feed_ev.push_back((hipEvent_t)events[j]);
    h->feed_upto.push_back(upto_step[j]);
  }
  h->feed_row_bytes = link_row_bytes;
  return PREGO_OK;
}

struct RowBytes { size_t x, y, e, gi, hr, hraw, gates, stats, map, l2keep, total; };
// bf16 mode, inference (no PREGO_FWD_KEEP): the two projections' outputs stay bf16 between the kernels (what a bf16 autocast
// of the reference does too).  They are the largest HBM streams of the pass (20 KB per frame in fp32) and the store tail of a
// GEMM tile is bound by bytes: with fp32 C the projections run 1.23 / 1.10 PFLOP/s (K = 4096 / 2048), without any C store 1.41 /
// 1.40; the numpy emulation of the whole path moves the worst probability error from 2.0e-3 to 2.5e-3 (tolerance 1e-2).
// Training keeps them fp32 (LayerNorm backward reads Y).  PREGO_FP32_INTERMEDIATES=1 restores fp32 for A/B.
static bool inter16(const prego_miniroad* h, int flags) {
  static const bool force32 = prego_tune_env("PREGO_FP32_INTERMEDIATES") != nullptr;
  return h->bf16 && !(flags & PREGO_FWD_KEEP) && !force32;
}
static RowBytes row_bytes(const prego_miniroad* h, bool with_flow, int flags) {
  const size_t es = h->bf16 ? 2 : 4;
  const size_t is = inter16(h, flags) ? 2 : 4;
  RowBytes r;
  r.x = (size_t)(h->d_rgb + (with_flow ? h->d_flow : 0)) * es;
  r.y = (size_t)h->emb * is;
  r.e = (size_t)h->emb * es;
  r.gi = (size_t)3 * h->hid * is;
  r.hr = (size_t)h->hid * es;
  const bool keep = (flags & PREGO_FWD_KEEP) != 0;
  r.hraw = keep ? (size_t)h->hid * 4 : 0;
  r.gates = keep ? (size_t)h->hid * 4 * 4 : 0;      // r, z, n, W_hn h + b_hn
  r.stats = keep ? 8 : 0;                           // LayerNorm mean, rstd
  r.map = 16;                                       // row -> (clip, frame) for the head's scatter, two chunks deep
  // training a two-layer GRU (round 6): layer 0's h_t as layer 1's input operand, layer 1's raw state and its four gate activations
  r.l2keep = (keep && h->layers == 2) ? (size_t)h->hid * es + (size_t)h->hid * 4 + (size_t)h->hid * 4 * 4 : 0;
  r.total = r.x + r.y + r.e + r.gi + r.hr + r.hraw + r.gates + r.stats + r.map + r.l2keep;
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
  return (size_t)rows * rb.total + 12 * 256;
}

// Whole-call resident buffer (round 6; SURVEY 8b: "no allocation of caller-visible memory, workspace sized by a query and passed in").
// A pass that runs the classifier once per call keeps relu(h) of every packed row (2 KB per frame with 16-bit operands, 4 KB with fp32 /
// fp16x2), the split pass also its row map and counters.  Until round 5 forward() grew a handle-owned hipMalloc for it (behind a stream
// synchronisation); now the caller sizes it here and hands it over with prego_miniroad_set_resident.
static size_t split_buf_need(const prego_miniroad* h, long long total);
extern "C" size_t prego_miniroad_resident_bytes(const prego_miniroad* h, int n_clips, const int32_t* lens, int flags) {
  if (!h || n_clips <= 0 || !lens) return 0;
  if ((flags & PREGO_FWD_KEEP) || h->layers != 1) return 0;
  long long total = 0;
  for (int i = 0; i < n_clips; ++i) total += lens[i] > 0 ? lens[i] : 0;
  if (total < 65536) return 0;                      // fewer than four chunks of the smallest useful size: the per-chunk head runs
  const RowBytes rb = row_bytes(h, true, flags);
  size_t need = align_up((size_t)total * rb.hr, 256);
  if (h->bf16 && h->hid == 1024) need = std::max(need, split_buf_need(h, total));
  return need <= ((size_t)24 << 30) + ((size_t)1 << 30) ? need : 0;
}

extern "C" int prego_miniroad_set_resident(prego_miniroad* h, void* device_buffer, size_t bytes) {
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  if ((device_buffer == nullptr) != (bytes == 0)) return fail(PREGO_EINVAL, "set_resident: buffer %p with %zu bytes", device_buffer, bytes);
  if ((uintptr_t)device_buffer & 255) return fail(PREGO_EINVAL, "set_resident: the buffer must be 256-byte aligned");
  h->res_buf = (char*)device_buffer; h->res_bytes = bytes;
  return PREGO_OK;
}

// Data-parallel guard (round 6, advisor): a rank whose recurrence / BPTT gave up must stop EVERY rank's optimizer step, not only its own.
// publish: dst[0] = 1.0f if this handle's timeout word is set, else 0.0f - enqueued; dst is an element of the gradient bucket the ranks
// all-reduce (sum).  peer guard: the address of that element; prego_miniroad_adamw_step then changes nothing while it holds a non-zero
// value and raises this handle's own word (code 0x200), so prego_miniroad_check reports the step on every rank.
void launch_guard_publish(const unsigned* abort_word, float* dst, hipStream_t s);
extern "C" int prego_miniroad_guard_publish(prego_miniroad* h, float* dst, prego_stream_t stream) {
  HandleScope scope_(h);
  if (!h || !dst) return fail(PREGO_EINVAL, "guard_publish: NULL argument");
  launch_guard_publish(h->abort_word, dst, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}
extern "C" int prego_miniroad_set_peer_guard(prego_miniroad* h, const float* reduced_word) {
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  h->peer_guard = reduced_word;
  return PREGO_OK;
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

// the side stream was created with the LEAST priority so that it never shares a hardware queue with a normal-priority caller (create);
// a caller's stream of that same priority might: two persistent launches that wait for each other must not be queued one behind the other
static bool side_queue_differs(const prego_miniroad* h, hipStream_t s) {
  int ps = 0, pside = 0;
  if (hipStreamGetPriority(s, &ps) != hipSuccess || hipStreamGetPriority(h->side, &pside) != hipSuccess) return false;
  return ps != pside;
}
static void refresh_placement(prego_miniroad* h) {
  if (h->placement < 0 && h->place_pending && hipEventQuery(h->ev_place) == hipSuccess) {
    const int v = (int)*h->pin_place;             // 1: group := XCD verified; 2: another placement; 0: that launch did not run the
    h->placement = v == 0 ? -1 : v;               // full-width rendezvous (multi-tile kernel): look again behind a later launch
    h->place_pending = false;
  }
}

// ---- split pass ----------------------------------------------------------------------------------------------------------------
// Geometry: units of 256 packed rows; chunks of 8 units (2 048 rows) are what the two kernels tell each other about; X / Y / E rings
// of 24 units (twelve super-rounds of two) per feed-forward XCD, a GI ring of 32 units = 4 chunks.  The rings come out of the caller's
// workspace (0.54 GB: they fit the default one), relu(h) + the row map of the whole pass and the counters live in the caller's resident
// buffer (the head runs once, behind the pass).
// (debug library: PREGO_SPLIT_GI_RING = a power of two of units, at least four chunks; PREGO_SPLIT_RING_PER_XCD = an even number of units
// beyond the largest lag - sweeps of how much of the rings the 256 MB Infinity Cache can hold)
static int tuned_pow2(const char* name, int dflt) {
  const char* v = prego_tune_env(name);
  const int x = v ? atoi(v) : 0;
  return (x >= 16 && (x & (x - 1)) == 0) ? x : dflt;
}
// Round 6: a GI ring of 32 units (50 MB; 4 chunks of 8 units) instead of 256 (0.4 GB; 4 chunks of 64): same-device 91.7-92.0 against
// 93.1 ms and 93.4 against 95.1-95.2 (profiles/r06_split_rings*.log).  The feed-forward launch may run 8 192 rows (~170 recurrence steps)
// ahead instead of 65 536: what it has written and the recurrence has not yet read stays in the Infinity Cache, and so does more of its own
// X -> Y -> E chain.  64 units: -0.5...-0.8 %; 32 units in chunks of 4: -1.0 %; 16 units or two chunks of 16: the launches wait for each
// other (+0.4...+11 %).  The X / Y / E ring's size does not matter (16 / 24 / 32 units per XCD: +-0.1 %).
// The rgb-only pass on 4 + 4 XCDs is bound by its feed-forward launch, not by the recurrence, and keeps the long ring (28.1-28.2 against
// 27.8-28.0 M frames/s, profiles/r06_split_rings_zf.log).
static int split_gi_ring_units(int R) { return tuned_pow2("PREGO_SPLIT_GI_RING", R <= 3 ? 32 : 256); }
static const int kSplitRingPerXcd = (prego_tune_env("PREGO_SPLIT_RING_PER_XCD") && atoi(prego_tune_env("PREGO_SPLIT_RING_PER_XCD")) >= 12 &&
                                     atoi(prego_tune_env("PREGO_SPLIT_RING_PER_XCD")) % 4 == 0) ? atoi(prego_tune_env("PREGO_SPLIT_RING_PER_XCD")) : 24;   // ring: 12 super-rounds of 2 units
// units per super-round (debug library: sweep).  Round 6: 2 instead of 4.  ALONE the feed-forward launch is flat between 2 and 4 (93.0 / 92.3 ms,
// round 5); IN THE PASS 2 is 1.0-1.2 % faster on every box and alternation (profiles/r06_split_knobs.log: 95.6-97.7 against 96.8-98.6 ms):
// half the look-ahead in rows (lags 2 / 3 / 4 super-rounds = 4 / 6 / 8 units) keeps a unit's X -> Y -> E -> GI chain closer together in
// the XCD's L2, and a weight slab is still shared by two row blocks.  1 (no sharing) runs the GEMM tiles at 0.75 of the rate: 127 ms.
static const int kSplitSg = (prego_tune_env("PREGO_SPLIT_SG") && kSplitRingPerXcd % std::max(1, atoi(prego_tune_env("PREGO_SPLIT_SG"))) == 0)
                                ? std::max(1, atoi(prego_tune_env("PREGO_SPLIT_SG"))) : 2;
static int split_chunk_shift(int R) { return prego_tune_env("PREGO_SPLIT_CHUNK_SHIFT") ? atoi(prego_tune_env("PREGO_SPLIT_CHUNK_SHIFT")) : (R <= 3 ? 3 : 6); }
struct SplitRings { size_t x, y, e, gi, total; int ring_units; };
static SplitRings split_rings(const prego_miniroad* h, int R) {
  SplitRings g;
  g.ring_units = kSplitRingPerXcd * (8 - R);
  g.x = align_up((size_t)g.ring_units * 256 * (size_t)(h->d_rgb + h->d_flow) * 2, 256);
  g.y = align_up((size_t)g.ring_units * 256 * (size_t)h->emb * 2, 256);
  g.e = g.y;
  g.gi = align_up((size_t)split_gi_ring_units(R) * 256 * (size_t)3 * h->hid * 2, 256);
  g.total = g.x + g.y + g.e + g.gi;
  return g;
}
static bool split_workspace_ok(const prego_miniroad* h, int R, size_t workspace_bytes) { return workspace_bytes >= split_rings(h, R).total; }

// whole-call buffer of a split pass: relu(h) rows | row map | counters.  It lives in the caller's resident buffer
// (prego_miniroad_resident_bytes / _set_resident); a buffer that is too small keeps the call on the chunked pass
static size_t split_buf_need(const prego_miniroad* h, long long total) {
  const long long n_units = (total + 255) / 256;
  const int shift = split_chunk_shift(1);                       // the smallest chunk any R uses: the most counters
  const long long n_chunks = (n_units + (1 << shift) - 1) >> shift;
  return align_up((size_t)total * h->hid * 2, 256) + align_up((size_t)total * 8, 256) + align_up(((size_t)4 * n_units + 2 * (size_t)n_chunks + 32) * 4, 256);
}
static bool split_resident_ok(const prego_miniroad* h, long long total) { return h->res_buf && split_buf_need(h, total) <= h->res_bytes; }

// Split passes of DIFFERENT handles on one device must not interleave: handle A's feed-forward launch resident on XCDs R .. 7 with handle
// B's recurrence launch resident on XCDs 0 .. R - 1 wait for each other's partner, which can never be dispatched (bounded, but both calls
// are lost).  Every split pass therefore starts behind the end of the previous one on the device, whatever handle / stream it came from.
static std::mutex g_split_mu;
static hipEvent_t g_split_last[64] = {};

// *fell_back = true (with PREGO_OK): the start handshake of the two launches failed - they left without writing anything, the caller
// runs the chunked pass for this call.
static int forward_split(prego_miniroad* h, int R, int flags, bool with_flow, bool in16, int kx, const SlotPlan& plan,
                         const float* const* d_rgb_ptrs, const float* const* d_flow_ptrs, float* const* d_out_ptrs, int* const* d_arg_ptrs,
                         void* workspace, size_t workspace_bytes, hipStream_t s, bool* fell_back) {
  *fell_back = false;
  const int H = h->hid, E = h->emb, din = h->d_rgb + h->d_flow;
  const int total = h->h_rowoff[h->t_max];
  const int n_units = (total + 255) / 256;
  const int chunk_shift = split_chunk_shift(R), gi_ring = split_gi_ring_units(R);
  const int upc = 1 << chunk_shift;
  const int n_chunks = (n_units + upc - 1) >> chunk_shift;
  const SplitRings rg = split_rings(h, R);
  if (workspace_bytes < rg.total) return fail(PREGO_EWORKSPACE, "split pass: workspace %zu B < %zu B of rings", workspace_bytes, rg.total);
  char* wp = (char*)workspace;
  unsigned short* X = (unsigned short*)wp; wp += rg.x;
  unsigned short* Y = (unsigned short*)wp; wp += rg.y;
  unsigned short* Eb = (unsigned short*)wp; wp += rg.e;
  unsigned short* GI = (unsigned short*)wp;
  // handle-owned: relu(h) of every packed row, the row map, the counters
  const size_t hr_bytes = align_up((size_t)total * H * 2, 256), rm_bytes = align_up((size_t)total * 8, 256);
  const size_t n_ctr = (size_t)4 * n_units + 2 * (size_t)n_chunks + 32;
  if (!h->split_warm) { h->meas_armed = false; h->split_warm = true; }       // a handle's first split pass loads kernels: not a measurement
  if (!split_resident_ok(h, total)) return fail(PREGO_EWORKSPACE, "split pass: resident buffer %zu B < %zu B", h->res_bytes, split_buf_need(h, total));
  char* HR = h->res_buf;
  char* RM = HR + hr_bytes;
  unsigned* ctr = (unsigned*)(RM + rm_bytes);
  unsigned* tick = ctr; unsigned* hs_word = ctr + 8; unsigned* ff_here = ctr + 16; unsigned* pack_done = ctr + 32; unsigned* l1_cnt = pack_done + n_units; unsigned* ln_done = l1_cnt + n_units;
  unsigned* wih_cnt = ln_done + n_units; unsigned* gi_cnt = wih_cnt + n_units; unsigned* rec_cnt = gi_cnt + n_chunks;
  HIPCHK(hipMemsetAsync(ctr, 0, n_ctr * 4, s));
  HIPCHK(hipMemsetAsync(h->h_state, 0, (size_t)h->n_slots * H * 4, s));
  // start handshake (kernels.h: PassHandshake).  Bounds: the two launches are released by the same fork point and start microseconds
  // apart; 50 / 100 ms leave room for another stream's kernels draining from the CUs first.  A pass that cannot run side by side costs
  // that long ONCE (the back-off in prego_miniroad_forward keeps the handle chunked afterwards)
  PassHandshake hs{};
  h->hs_seq = (h->hs_seq + 1u) & 0x3FFFFFFFu;
  if (h->hs_seq == 0u) h->hs_seq = 1u;
  hs.word = hs_word; hs.ff_here = ff_here; hs.host = h->pin_hs; hs.seq = h->hs_seq; hs.ticks_lead = 5000000u; hs.ticks_all = 10000000u;
  int fault = 0;
#ifdef PREGO_DEBUG_ABI
  fault = h->dbg_fault; h->dbg_fault = 0;           // prego_debug_split_fault: one shot
  if (fault == 3 || fault == 4) {
    // replay of ONE of the two launches alone (counter collection serialises dispatches, so the pair cannot run under it): the handshake
    // is pre-decided and the other side's counters pre-armed - the feed-forward launch never waits for a GI ring slot, the recurrence
    // launch reads whatever finite rows an earlier pass left in the ring.  Same instruction stream and memory traffic, meaningless outputs
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)hs_word, PREGO_HS_GO, 1, s));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)ff_here, 1, 8, s));
    if (fault == 3) HIPCHK(hipMemsetD32Async((hipDeviceptr_t)rec_cnt, R * h->P * 4, n_chunks, s));
    else HIPCHK(hipMemsetD32Async((hipDeviceptr_t)gi_cnt, upc, n_chunks, s));
    hs.host = nullptr;
  }
#endif

  FfPassArgs fa{};
  fa.rgb_ptrs = d_rgb_ptrs; fa.flow_ptrs = with_flow ? d_flow_ptrs : nullptr; fa.plan = plan; fa.rowmap = RM;
  fa.d_rgb = h->d_rgb; fa.d_flow = with_flow ? h->d_flow : 0; fa.in16 = in16 ? 1 : 0; fa.kx = kx;
  fa.w1 = (const unsigned short*)h->w1; fa.ld_w1 = din; fa.b1 = h->b1; fa.ln_g = h->ln_g; fa.ln_b = h->ln_b; fa.ln_eps = 1e-5f;
  fa.w_ih = (const unsigned short*)h->w_ih_perm; fa.bias2 = h->bias2_perm; fa.E = E; fa.n3 = 3 * H;      // permuted rows: GI rows in (unit pair, gate) order
  fa.X = X; fa.Y = Y; fa.Eb = Eb; fa.GI = GI; fa.ring_units = rg.ring_units; fa.gi_ring_units = gi_ring;
  fa.total_rows = total; fa.n_units = n_units; fa.xcd_lo = R; fa.chunk_unit_shift = chunk_shift;
  fa.rec_expect = R * h->P * 4; fa.nt1 = E / 256; fa.nt2 = 3 * H / 256;
  static const int lag1 = prego_tune_env("PREGO_SPLIT_LAG1") ? atoi(prego_tune_env("PREGO_SPLIT_LAG1")) : 2;
  static const int lag2 = prego_tune_env("PREGO_SPLIT_LAG2") ? atoi(prego_tune_env("PREGO_SPLIT_LAG2")) : 3;
  static const int lag3 = prego_tune_env("PREGO_SPLIT_LAG3") ? atoi(prego_tune_env("PREGO_SPLIT_LAG3")) : 4;
  static const bool want_stats = prego_tune_env("PREGO_SPLIT_STATS") != nullptr;
  fa.sg = kSplitSg; fa.lag1 = lag1; fa.lag2 = lag2; fa.lag3 = lag3; fa.f16 = h->f16 ? 1 : 0;
  static const int ff_cus = prego_tune_env("PREGO_SPLIT_FF_CUS") ? atoi(prego_tune_env("PREGO_SPLIT_FF_CUS")) : 0;        // debug library only
  fa.max_wg = ff_cus;
  fa.stats = want_stats ? h->stamps : nullptr;
#ifdef PREGO_DEBUG_ABI
  static const int dbg = prego_tune_env("PREGO_SPLIT_DBG") ? atoi(prego_tune_env("PREGO_SPLIT_DBG")) : 0;     // timing experiments (wrong results): debug library only
  fa.dbg = dbg;
#endif
  fa.tick = tick; fa.pack_done = pack_done; fa.l1_cnt = l1_cnt; fa.ln_done = ln_done; fa.wih_cnt = wih_cnt; fa.gi_cnt = gi_cnt;
  fa.rec_cnt = rec_cnt; fa.abort_word = h->abort_word; fa.hs = hs;
  // a job may only ever wait for jobs with earlier tickets: the previous holder of a ring slot (ring / sg super-rounds back) must have been
  // issued before the job that overwrites the slot
  const int ring_sr = kSplitRingPerXcd / kSplitSg;
  if (lag1 < 1 || lag2 <= lag1 || lag3 <= lag2 || lag1 >= ring_sr || lag2 - ring_sr >= lag1 || lag3 - ring_sr >= lag2)
    return fail(PREGO_EINVAL, "split pass: lags %d %d %d", lag1, lag2, lag3);

  GruArgs ga{};
  ga.whh = h->w_hh; ga.b_hn = h->b_hn; ga.gi = GI; ga.gi_bf16 = 1; ga.f16 = h->f16 ? 1 : 0; ga.h_relu_out = HR; ga.h_raw_out = nullptr;
  ga.h_state = h->h_state; ga.hx = h->hx; ga.flags = h->flags; ga.abort_word = h->abort_word;
  ga.rowoff = h->d_rowoff; ga.nact = h->d_nact; ga.t0 = 0; ga.t1 = h->t_max; ga.row_base = 0; ga.rows = 0;
  ga.n_clips = h->n_slots; ga.G = h->G; ga.seg_off = h->plan_single ? nullptr : h->d_seg_off;
  ga.seg_start = h->plan_single ? nullptr : h->d_seg_start; ga.stamps = (h->use_stamps && !want_stats) ? h->stamps : nullptr;
  ga.sync = h->flags; ga.armed = 0; ga.Gd = R;
  ga.gi_cnt = gi_cnt; ga.rec_cnt = rec_cnt; ga.chunk_shift = chunk_shift + 8; ga.n_chunks = n_chunks;
  ga.units_per_chunk = upc; ga.units_last = n_units - upc * (n_chunks - 1); ga.gi_row_mask = (unsigned)gi_ring * 256u - 1u;
  ga.hs = hs;

  // the feed-forward launch goes to the side stream (another hardware queue: it must be resident TOGETHER with the recurrence), forked
  // from and joined to the caller's stream by events
  struct SideJoin {
    prego_miniroad* h; bool pending = false;
    ~SideJoin() { if (pending) (void)hipStreamSynchronize(h->side); }
  } side_join{h};
  // everything the recurrence launch needs done first goes IN FRONT of the fork: once the feed-forward kernel is resident it fills its CUs
  // completely, and an ordinary kernel of the caller's stream (the arm kernel, a memset) would wait for it - with the recurrence queued behind
  launch_gru_arm(true, H, h->G, h->hx, h->flags, s);
  if (h->perm_stale) {                       // in front of the fork, like the arm kernel: nothing of this stream may sit between the two launches
    launch_permute_gi_rows(h->w_ih, h->bias2, h->w_ih_perm, h->bias2_perm, H, E, s);
    h->perm_stale = false;
  }
  std::lock_guard<std::mutex> split_lock(g_split_mu);          // held until this pass is enqueued and its end event recorded
  int dev_ = 0;
  HIPCHK(hipGetDevice(&dev_));
  const bool dev_ok = dev_ >= 0 && dev_ < 64;
  if (dev_ok && g_split_last[dev_]) HIPCHK(hipStreamWaitEvent(s, g_split_last[dev_], 0));
  HIPCHK(hipEventRecord(h->ev_fork, s));
  HIPCHK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
  const bool run_ff = fault != 2 && fault != 4, run_rec = fault != 1 && fault != 3;
  const size_t ev_mark = h->ev_used;
  EventPair* evf = run_ff ? ev_begin(h, 2, h->side) : nullptr;       // timing_read: the feed-forward launch of a split pass is reported in the pack slot
  if (run_ff && launch_ff_pass(fa, h->side)) return fail(PREGO_EINVAL, "split pass: feed-forward shape E=%d kx=%d", E, kx);
  ev_end(evf, h->side);
  side_join.pending = true;
  HIPCHK(hipEventRecord(h->ev_join, h->side));
  EventPair* evr = run_rec ? ev_begin(h, 1, s) : nullptr;
  if (run_rec && launch_gru_recurrence_pass(H, ga, s)) return fail(PREGO_EINVAL, "split pass: recurrence launch");
  ev_end(evr, s);
  HIPCHK(hipStreamWaitEvent(s, h->ev_join, 0));
  side_join.pending = false;
  if (dev_ok) {
    if (!g_split_last[dev_]) HIPCHK(hipEventCreateWithFlags(&g_split_last[dev_], hipEventDisableTiming));
    HIPCHK(hipEventRecord(g_split_last[dev_], s));
  }
  if (fault == 3 || fault == 4) { HIPCHK(hipGetLastError()); return PREGO_OK; }      // replay of one launch: no head, outputs untouched
  // The calling thread waits here until the two launches have met (normally: the moment the stream reaches them).  GO: both are resident,
  // every wait of the pass has a running producer, the head is enqueued behind it.  FAIL: they have left without writing anything
  {
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    unsigned state = 0u; long long polls = 0;
    for (;;) {
      const unsigned v = __atomic_load_n(h->pin_hs, __ATOMIC_ACQUIRE);
      if ((v >> 2) == hs.seq && (v & 3u)) { state = v & 3u; break; }
      if ((++polls & 63) == 0) {
        // both launches gone and nobody decided (cannot happen: every workgroup of either launch votes within its bound): not a pass
        if (hipStreamQuery(s) == hipSuccess) {
          const unsigned v2 = __atomic_load_n(h->pin_hs, __ATOMIC_ACQUIRE);
          state = ((v2 >> 2) == hs.seq && (v2 & 3u)) ? (v2 & 3u) : PREGO_HS_FAIL;
          break;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(300))
          return fail(PREGO_ETIMEOUT, "split pass: the stream did not reach the pass within 300 s");
      }
      std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    if (state != PREGO_HS_GO) {
      h->ev_used = ev_mark;                    // the two launches' timing events do not describe a pass
      h->meas_armed = false;
      *fell_back = true;
      HIPCHK(hipGetLastError());
      return PREGO_OK;
    }
  }
  if (h->timing) { h->gemm_flop += 2.0 * total * ((double)E * kx + 3.0 * H * E); h->split_passes++; h->split_steps += h->t_max; }
  if (launch_head_softmax(true, HR, h->w_c, h->b_c, plan, 0, total, H, h->ncls, (flags & PREGO_FWD_SOFTMAX) ? 1 : 0, d_out_ptrs, d_arg_ptrs, s,
                          RM, h->f16))
    return fail(PREGO_EINVAL, "head: unsupported num_classes %d", h->ncls);
  if (h->meas_armed) { HIPCHK(hipEventRecord(h->ev_meas[1], s)); h->meas_pending = true; h->meas_armed = false; }
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

extern "C" int prego_miniroad_forward(prego_miniroad* h, int n_clips, const int32_t* lens, const float* const* rgb,
                                      const float* const* flow, float* const* out, int32_t* const* argmax,
                                      const float* h0, float* h_last, int flags, void* workspace,
                                      size_t workspace_bytes, prego_stream_t stream) {
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  if (!h->have_weights) return fail(PREGO_EINVAL, "forward before set_weights");
  if (h->layers == 2 && !h->have_layer2) return fail(PREGO_EINVAL, "forward of a 2-layer handle before set_gru_layer(1)");
  if ((flags & PREGO_FWD_KEEP) && !h->bf16 && h->hid == 2048)
    return fail(PREGO_EINVAL, "PREGO_FWD_KEEP (training) with hidden_dim 2048 needs bf16 operands (an fp32 W_hh slice of 2048 does not fit the register file)");
  if (n_clips <= 0 || !lens) return fail(PREGO_EINVAL, "no clips");
  if (n_clips > max_clips_of(h)) return fail(PREGO_EINVAL, "%d clips > max_clips %d per call", n_clips, max_clips_of(h));
  if (h->d_rgb > 0 && !rgb) return fail(PREGO_EINVAL, "rgb pointer array is NULL");
  if (!workspace) return fail(PREGO_EINVAL, "workspace is NULL");
  hipStream_t s = (hipStream_t)stream;
  if (h->f16 && (flags & PREGO_FWD_KEEP))
    return fail(PREGO_EINVAL, "PREGO_FWD_KEEP (training) on an fp16-operand handle: training runs on bf16 / fp32 handles");
  if (h->x2 && (flags & PREGO_FWD_KEEP))
    return fail(PREGO_EINVAL, "PREGO_FWD_KEEP (training) on a split-operand (fp16x2) handle: training runs on bf16 / fp32 handles");
  const bool in16 = (flags & PREGO_FWD_IN16) != 0;
  if (in16 && !h->bf16) return fail(PREGO_EINVAL, "PREGO_FWD_IN16 on an fp32-operand handle (16-bit features go with bf16 / fp16 handles)");
  if (in16 && (flags & PREGO_FWD_KEEP)) return fail(PREGO_EINVAL, "PREGO_FWD_IN16 with PREGO_FWD_KEEP: training takes fp32 features");
  const bool want_single = h0 != nullptr || h_last != nullptr || (flags & PREGO_FWD_KEEP) != 0;
  // link-fed call (prego_miniroad_set_feed_events): the feature arrays are being filled over the host link while this call runs
  const bool hostfeat = !h->feed_ev.empty();
  if (hostfeat && want_single) { h->feed_ev.clear(); return fail(PREGO_EINVAL, "feed events with h0 / h_last / PREGO_FWD_KEEP: link-fed calls are plain inference"); }
  const int host_row_bytes = hostfeat ? h->feed_row_bytes : 0;
  struct FeedClear { prego_miniroad* h; ~FeedClear() { h->feed_ev.clear(); h->feed_upto.clear(); h->feed_pos = 0; } } feed_clear{h};   // one call only
  refresh_placement(h);
  // split pass: the recurrence of the whole call on XCDs 0 .. R - 1 (16 R slots, continuous batching) and its feed-forward on
  // the other XCDs, two persistent launches instead of a chain of launches per chunk.  Plain inference calls of 16-bit handles with
  // enough clips to fill the slots and enough frames to amortise the pipeline fill; needs the verified placement (group := XCD) that an
  // earlier full-width launch of this handle established, so a handle's first call is always the chunked pass.
  int split_r = 0;
  h->meas_armed = false;
  {
    long long frames = 0;
    for (int i = 0; i < n_clips; ++i) frames += lens[i] > 0 ? lens[i] : 0;
    int r_try = h->split_env > 0 ? h->split_env : 3;       // unset: the candidate with the best estimate (below); 3 until estimated
    const bool with_flow_ = flow != nullptr && h->d_flow > 0 && flow[0] != nullptr;
    // everything but the placement (which a handle's first, chunked, call establishes)
    const bool shape_ok = h->split_env != 0 && r_try >= 1 && r_try <= 6 && h->bf16 && h->hid == 1024 && h->layers == 1 && !want_single && !hostfeat && h->G == 8 && !h->no_local &&
                          h->side != nullptr && side_queue_differs(h, s) && n_clips >= 16 * r_try && frames >= 262144 &&
                          frames < (1ll << 31) - 65536 && (out || argmax) && split_workspace_ok(h, r_try, workspace_bytes) &&
                          (h->d_rgb > 0 ? h->d_rgb : h->d_flow) >= 128 &&
                          (size_t)frames * (h->hid * 2 + 8) <= ((size_t)24 << 30) && split_resident_ok(h, frames);
    // a call of this class is worth one wait for the placement word of an earlier launch (the handle's second call otherwise races it)
    if (shape_ok && h->placement < 0 && h->place_pending) { (void)hipEventSynchronize(h->ev_place); refresh_placement(h); }
    bool backing_off = false;                       // a failed start handshake keeps the next eligible calls chunked
    if (shape_ok && h->split_skip > 0) { --h->split_skip; backing_off = true; }
    const bool eligible = shape_ok && h->placement == 1 && !backing_off;
    if (eligible && h->split_env > 0) split_r = r_try;
    else if (shape_ok) {
      // cost model (ms), calibrated on the bench workloads.  Chunked pass: the plan's recurrence estimate + the feed-forward of
      // every row on the whole chip (projections at 1.4 PFLOP/s, 3 ns of LayerNorm + head; the pack hides under the recurrence) + 30 us
      // per chunk.  Split pass: the slower of the 16 R-slot recurrence at 2.0 us per step and the feed-forward on 8 - R of 8 XCDs (pack
      // included, at 5.3 TB/s), + 1.5 ms of pipeline fill and the head behind the pass.  Both are scaled by what passes of that kind
      // took on this device so far (measured / estimated, events around every call of this shape class).
      // while one of the two kinds has never been timed on this handle, the host waits here for the pending measurement (at most the
      // handle's first two calls of this class lose their run-ahead); afterwards measurements are picked up when they happen to be done
      // (only for the FIRST measurement of a kind: a handle whose model never trials the split pass stops waiting after one chunked call)
      if (h->meas_pending && !(h->meas_mode ? h->have_ratio_split : h->have_ratio_chunked)) (void)hipEventSynchronize(h->ev_meas[1]);
      if (h->meas_pending && hipEventQuery(h->ev_meas[1]) == hipSuccess) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->ev_meas[0], h->ev_meas[1]) == hipSuccess && ms > 0 && h->meas_est > 0) {
          const double r = ms / h->meas_est;
          double& ratio = h->meas_mode ? h->ratio_split : h->ratio_chunked;
          bool& have = h->meas_mode ? h->have_ratio_split : h->have_ratio_chunked;
          ratio = have ? 0.25 * ratio + 0.75 * r : r;
          have = true;
        }
        h->meas_pending = false;
      }
      const int key = (with_flow_ ? 1 : 0) | (in16 ? 2 : 0) | (int)((workspace_bytes >> 20) << 2);
      if (!(key == h->split_seen_key && (int)h->split_seen_lens.size() == n_clips && std::equal(lens, lens + n_clips, h->split_seen_lens.begin()))) {
        const double kx_ = h->d_rgb + (with_flow_ ? h->d_flow : 0), E_ = h->emb, H3 = 3.0 * h->hid;
        const double gemm_ns = (2.0 * kx_ * E_ + 2.0 * E_ * H3) / 1.4e15 * 1e9;
        const double pack_ns = kx_ * ((in16 ? 2.0 : 4.0) + 2.0) / 5.3e12 * 1e9;
        int rc0 = build_plan(h, n_clips, lens, false, 0, 0);
        if (rc0) return rc0;
        const RowBytes rb0 = row_bytes(h, with_flow_, flags);
        const double chunk_rows = std::max(1.0, (double)((workspace_bytes - 12 * 256) / rb0.total));
        h->split_seen_est_c = h->plan_cost_us * 1e-3 + frames * (gemm_ns + 3.0) * 1e-6 + 0.03 * std::ceil(frames / chunk_rows);
        // how many XCDs for the recurrence: more slots shorten it (steps = frames / 16 R once every slot is busy), fewer XCDs lengthen the
        // feed-forward: R = 3 balances the rgb + flow workload, a zero-flow call (half of layer1's K) is better off with R = 4
        h->split_seen_est_s = 1e30; h->split_seen_r = r_try;
        for (int r = 3; r <= 4; ++r) {
          if (n_clips < 16 * r || !split_workspace_ok(h, r, workspace_bytes)) continue;
          rc0 = build_plan(h, n_clips, lens, false, 0, 16 * r);
          if (rc0) return rc0;
          const double e = std::max(h->t_max * 2.0e-3, frames * (gemm_ns + pack_ns + 1.5) * 1e-6 * 8.0 / (8 - r)) + 1.5;
          if (e < h->split_seen_est_s) { h->split_seen_est_s = e; h->split_seen_r = r; }
        }
        h->split_seen_lens.assign(lens, lens + n_clips); h->split_seen_key = key;
      }
      r_try = h->split_seen_r;
      if (eligible) {
        // learning order: a chunked pass first (the handle's very first call does not count: kernels are still being loaded, and it ran
        // before the placement was known), then a split trial if the model says it is close, then the corrected comparison
        const double es = h->split_seen_est_s * h->ratio_split, ec = h->split_seen_est_c * h->ratio_chunked;
        if (!h->have_ratio_chunked) split_r = 0;
        else if (!h->have_ratio_split) split_r = es < 1.05 * ec ? r_try : 0;
        else split_r = es < 0.98 * ec ? r_try : 0;
      }
      if (eligible && !h->meas_pending) {       // time this call (one measurement in flight at a time)
        h->meas_armed = true; h->meas_mode = split_r > 0 ? 1 : 0;
        h->meas_est = split_r > 0 ? h->split_seen_est_s : h->split_seen_est_c;
        HIPCHK(hipEventRecord(h->ev_meas[0], s));
      }
    }
  }
  h->split_r = split_r;
  int rc = build_plan(h, n_clips, lens, want_single, host_row_bytes, split_r > 0 ? 16 * split_r : 0);
  if (rc) return rc;
  const SlotPlan plan = device_plan(h);
  const int n_slots = h->n_slots;

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
  rc = stage_tables(h, tab.data(), tab.size(), s);
  if (rc) return rc;
  const float* const* d_rgb_ptrs = h->d_rgb > 0 ? (const float* const*)(h->d_ptrs + 0 * MC) : nullptr;
  const float* const* d_flow_ptrs = any_flow ? (const float* const*)(h->d_ptrs + 1 * MC) : nullptr;
  float* const* d_out_ptrs = out ? (float* const*)(h->d_ptrs + 2 * MC) : nullptr;
  int* const* d_arg_ptrs = argmax ? (int* const*)(h->d_ptrs + 3 * MC) : nullptr;

  // workspace carve
  const bool with_flow = any_flow;
  const int kx = h->d_rgb + (with_flow ? h->d_flow : 0);      // K of the layer1 GEMM actually multiplied
  if (kx == 0) return fail(PREGO_EINVAL, "a model without rgb features (--no_rgb) needs the flow tensors");
  if (split_r > 0) {
    bool fell_back = false;
    rc = forward_split(h, split_r, flags, with_flow, in16, kx, plan, d_rgb_ptrs, d_flow_ptrs, d_out_ptrs, d_arg_ptrs, workspace, workspace_bytes, s,
                       &fell_back);
    if (rc || !fell_back) return rc;
    // The two launches could not run side by side (a profiler that serialises dispatches, another tenant on the XCDs) and left before
    // touching anything: THIS call runs as a chunked pass, right here, behind them in the stream.  Back-off: the next 16, then 64 eligible
    // calls stay chunked, a third failure keeps the handle chunked for good
    h->split_fails++; h->split_fallbacks++;
    if (h->split_fails >= 3) h->split_env = 0;
    else h->split_skip = 16ll << (2 * (h->split_fails - 1));
    return prego_miniroad_forward(h, n_clips, lens, rgb, flow, out, argmax, h0, h_last, flags, workspace, workspace_bytes, stream);
  }
  const int din = h->d_rgb + h->d_flow;
  const RowBytes rb = row_bytes(h, with_flow, flags);
  const int total_rows = h->h_rowoff[h->t_max];
  if (workspace_bytes < 12 * 256 + 128 * rb.total) return fail(PREGO_EWORKSPACE, "workspace %zu B is too small", workspace_bytes);
  long long cap_rows = (long long)((workspace_bytes - 12 * 256) / rb.total) / 128 * 128;
  if (cap_rows < n_slots) return fail(PREGO_EWORKSPACE, "workspace holds %lld rows, need >= %d (one time step)", cap_rows, n_slots);
  if ((flags & PREGO_FWD_KEEP) && cap_rows < total_rows)
    return fail(PREGO_EWORKSPACE, "PREGO_FWD_KEEP needs the whole batch resident: %d rows, workspace holds %lld", total_rows, cap_rows);
  char* wp = (char*)workspace;
  auto carve = [&](size_t bytes) { char* p = wp; wp += align_up(bytes, 256); return (void*)p; };
  void* X = carve((size_t)cap_rows * rb.x);
  void* Y = carve((size_t)cap_rows * rb.y);
  void* Eb = carve((size_t)cap_rows * rb.e);
  void* GI = carve((size_t)cap_rows * rb.gi);
  void* HR = carve((size_t)cap_rows * rb.hr);
  float* HRAW = rb.hraw ? (float*)carve((size_t)cap_rows * rb.hraw) : nullptr;
  // The classifier once per pass.  A pass of many chunks pays the head kernel's launch, its fill and its scatter per chunk (split
  // operands: 71 launches of the fp32 head = 10.3 ms of a 280 ms pass; 16-bit operands: 47 x ~123 us): with relu(h) of the whole call
  // resident - 4 KB (fp32 / fp16x2) or 2 KB per frame in the CALLER's resident buffer (prego_miniroad_set_resident) - ONE launch behind
  // the last chunk does the same work at its HBM rate.  Inference calls of one GRU layer whose rows span four or more chunks; a link-fed
  // call keeps the per-chunk head (its last chunk ends with the link, and a whole-pass head behind it would be pure tail); no resident
  // buffer, or one that is too small = per-chunk head.  Nothing is allocated and nothing is waited for here.
  bool defer_head = false;
  char* HRall = nullptr;
  if (!(flags & PREGO_FWD_KEEP) && !hostfeat && h->layers == 1 && (out || argmax) && (long long)total_rows >= 4 * cap_rows &&
      (size_t)total_rows * rb.hr <= ((size_t)24 << 30)) {
    const size_t need = align_up((size_t)total_rows * rb.hr, 256);
    if (h->res_buf && need <= h->res_bytes) { defer_head = true; HRall = h->res_buf; }
  }
  const bool i16 = inter16(h, flags);
  // projection with fp32 or bf16 output: ping-pong kernel for whole-chip shapes, the 128x128 kernel with a bf16-store epilogue below
  auto proj = [&](const void* A, int lda, const void* Wt, int ldb, const float* bias, void* Cout, int ldc, int M, int N, int K) {
    if (h->x2) {                    // A rows [K hi | K lo] (lda = K), W rows [ldb hi | ldb lo]: three fp16 products, fp32 C
      const float* inv = h->x2_scale + (Wt == h->w1 ? 1 : 3);
      (void)launch_gemm_x2_pingpong(A, 2 * lda, lda, Wt, 2 * ldb, ldb, inv, bias, (float*)Cout, ldc, M, N, K, s);
      return;
    }
    if (!h->bf16) { launch_gemm_f32_nt((const float*)A, lda, (const float*)Wt, ldb, bias, (float*)Cout, ldc, M, N, K, s); return; }
    if (!i16 && !h->f16) { launch_gemm_bf16_nt(A, lda, Wt, ldb, bias, (float*)Cout, ldc, M, N, K, s, false, (flags & PREGO_FWD_KEEP) != 0); return; }
    if (M >= 4096 && launch_gemm_bf16_pingpong_mode(0, A, lda, Wt, ldb, bias, Cout, ldc, M, N, K, i16, s, h->f16) == 0) return;
    GemmEpi epi{};
    epi.f16 = h->f16 ? 1 : 0;
    if (i16) { epi.mode = EPI_STORE_BF16; epi.out_b = Cout; } else epi.mode = EPI_STORE;
    launch_gemm_bf16_nt_epi(A, lda, Wt, ldb, bias, i16 ? nullptr : (float*)Cout, ldc, M, N, K, epi, s);
  };
  float* KR = nullptr; float* KZ = nullptr; float* KN = nullptr; float* KG = nullptr; float* STATS = nullptr;
  const bool keep = (flags & PREGO_FWD_KEEP) != 0;
  if (keep) {
    if (h0) return fail(PREGO_EINVAL, "PREGO_FWD_KEEP (training) runs from h0 = 0 (rnn.py:49,60): h0 must be NULL");
    KR = (float*)carve((size_t)cap_rows * h->hid * 4); KZ = (float*)carve((size_t)cap_rows * h->hid * 4);
    KN = (float*)carve((size_t)cap_rows * h->hid * 4); KG = (float*)carve((size_t)cap_rows * h->hid * 4);
    STATS = (float*)carve((size_t)cap_rows * 8);
    h->kept_kx = kx; h->kept_rows = total_rows;
  }
  // two-layer training: layer 0's h_t [rows][H] (operand type: layer 1's input, and the B operand of dW_ih_l1), layer 1's raw state and gates
  void* HR0 = nullptr; float* HRAW2 = nullptr; float* KR2 = nullptr; float* KZ2 = nullptr; float* KN2 = nullptr; float* KG2 = nullptr;
  if (keep && h->layers == 2) {
    HR0 = carve((size_t)cap_rows * h->hid * (h->bf16 ? 2 : 4));
    HRAW2 = (float*)carve((size_t)cap_rows * h->hid * 4);
    KR2 = (float*)carve((size_t)cap_rows * h->hid * 4); KZ2 = (float*)carve((size_t)cap_rows * h->hid * 4);
    KN2 = (float*)carve((size_t)cap_rows * h->hid * 4); KG2 = (float*)carve((size_t)cap_rows * h->hid * 4);
  }

  char* RM = (char*)carve((size_t)cap_rows * rb.map);        // [2][cap_rows] int2: chunk c uses half c & 1 (the pack of chunk c+1
                                                             // runs under the recurrence of chunk c, before the head of chunk c)
  // initial state (sorted order)
  const int H = h->hid, E = h->emb;
  // state of layer l: h_state + l * slot_stride; h0 / h_last of a 2-layer handle are [layers][n_clips][H] (nn.GRU's h_0 / h_n layout)
  const size_t slot_stride = (size_t)max_slots_of(h) * H;
  for (int l = 0; l < h->layers; ++l) {
    if (h0) launch_permute_rows(h0 + (size_t)l * n_clips * H, h->h_state + l * slot_stride, h->d_sorted, n_slots, H, 1, s);        // one clip per slot here
    else HIPCHK(hipMemsetAsync(h->h_state + l * slot_stride, 0, (size_t)n_slots * H * 4, s));
  }

  const int slots = (n_slots + h->G - 1) / h->G;
  const int nct = (slots + 15) / 16;          // live 16-clip tiles per group (kernels: 1, 2, 4, 8)

  // chunk [t0, t1): the largest t1 with rowoff[t1] - rowoff[t0] <= cap_rows
  auto chunk_end = [&](int t0_) {
    const int base_ = h->h_rowoff[t0_];
    int t1_ = (int)(std::upper_bound(h->h_rowoff.begin() + t0_, h->h_rowoff.end(), base_ + (int)std::min<long long>(cap_rows, total_rows)) -
                    h->h_rowoff.begin()) - 1;
    if (t1_ <= t0_) t1_ = t0_ + 1;
    if (t1_ > h->t_max) t1_ = h->t_max;
    return t1_;
  };
  auto pack_chunk = [&](int t0_, int t1_, hipStream_t st, int ci_) {
    const int base_ = h->h_rowoff[t0_], rows_ = h->h_rowoff[t1_] - base_;
    // link-fed call: this chunk reads rows of steps < t1_; make the packing stream wait for every feed event that covers them
    while (h->feed_pos < h->feed_ev.size() && (h->feed_pos == 0 || h->feed_upto[h->feed_pos - 1] < t1_)) {
      (void)hipStreamWaitEvent(st, h->feed_ev[h->feed_pos], 0);
      ++h->feed_pos;
    }
    EventPair* evp = ev_begin(h, 2, st);
    if (h->x2)
      launch_pack_rows_x2(d_rgb_ptrs, d_flow_ptrs, plan, base_, rows_, h->d_rgb, with_flow ? h->d_flow : 0, X, st,
                          st == s ? 0 : h->prefetch_grid, RM + (size_t)(ci_ & 1) * cap_rows * 8);
    else
    launch_pack_rows(h->bf16, d_rgb_ptrs, d_flow_ptrs, plan, base_, rows_, h->d_rgb, with_flow ? h->d_flow : 0, X, st,
                     st == s ? 0 : h->prefetch_grid, RM + (size_t)(ci_ & 1) * cap_rows * 8, h->f16, in16);
    ev_end(evp, st);
    if (h->timing) h->pack_bytes += (double)rows_ * (kx * (in16 ? 2.0 : 4.0) + rb.x);
  };
  const bool prefetch = h->pack_prefetch && !keep && h->side != nullptr;
  bool packed = false;            // X already holds this chunk (packed on the side stream under the previous recurrence)
  bool l1_done = false;           // Y already holds layer1 of this chunk (XCD overlap: the worker GEMM ran under the previous recurrence)
  refresh_placement(h);
  const bool overlap_ok = h->xcd_overlap && prefetch && i16 && h->bf16 && h->G == 8 && h->placement == 1 && !h->no_local && h->layers == 1 && h->hid == 1024;
  if (overlap_ok) HIPCHK(hipMemsetAsync(h->tile_ctr, 0, 4096 * sizeof(unsigned), s));
  // every exit path after a fork joins the side stream: an error return while the next chunk's pack is still writing X / RM
  // would leave the caller's stream unordered against it (the next forward on this handle could race with that pack)
  struct SideJoin {
    prego_miniroad* h; bool pending = false;
    ~SideJoin() { if (pending) (void)hipStreamSynchronize(h->side); }
  } side_join{h};
  int t0 = 0, ci = 0;             // ci: chunk counter (parity of the row-map half)
  while (t0 < h->t_max) {
    const int base = h->h_rowoff[t0];
    const int t1 = chunk_end(t0);
    const int rows = h->h_rowoff[t1] - base;
    EventPair* ev;
    if (!packed) pack_chunk(t0, t1, s, ci);
    packed = false;

    if (!l1_done) {
      ev = ev_begin(h, 0, s);
      proj(X, kx, h->w1, din, h->b1, Y, E, rows, E, kx);
      ev_end(ev, s);
      if (h->timing) h->gemm_flop += 2.0 * rows * (double)E * kx;
    }
    l1_done = false;
    // PREGO_PACK_EARLY=1 (A/B): X is dead from here on, so the next chunk's pack may run beside LayerNorm + the W_ih GEMM (MFMA-bound,
    // HBM half idle) instead of beside the recurrence (whose L2 hand-off it slows)
    static const bool pack_early = prego_tune_env("PREGO_PACK_EARLY") != nullptr;
    const bool early = pack_early && prefetch && t1 < h->t_max;
    if (early) {
      HIPCHK(hipEventRecord(h->ev_fork, s));
      HIPCHK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
      pack_chunk(t1, chunk_end(t1), h->side, ci + 1);
      side_join.pending = true;
    }
    // the LayerNorm launch also re-arms the recurrence's exchange buffers and rendezvous words (it runs after the previous recurrence
    // launch of this stream and before the next): one launch and one launch gap fewer per chunk (PREGO_NO_ARM_FUSE=1: A/B)
    static const bool arm_fuse = prego_tune_env("PREGO_NO_ARM_FUSE") == nullptr;
    const GruArm arm = h->x2 ? gru_x2_arm_desc(h->hid, h->G, h->hx, h->no_local ? nullptr : h->flags)
                             : gru_arm_desc(h->bf16, h->hid, h->G, h->hx, h->no_local ? nullptr : h->flags);
    if (h->x2) launch_ln_relu_x2((const float*)Y, h->ln_g, h->ln_b, rows, E, 1e-5f, Eb, s, arm_fuse ? &arm : nullptr);
    else
    launch_ln_relu(h->bf16, Y, h->ln_g, h->ln_b, rows, E, 1e-5f, Eb, STATS, keep ? h->drop_p : 0.f, h->drop_seed, base, s, 1, i16, h->f16,
                   arm_fuse ? &arm : nullptr);
    ev = ev_begin(h, 0, s);
    proj(Eb, E, h->w_ih, E, h->bias2, GI, 3 * H, rows, 3 * H, E);
    ev_end(ev, s);
    if (h->timing) h->gemm_flop += 2.0 * rows * 3.0 * H * E;

    GruArgs ga{};
    ga.whh = h->w_hh; ga.b_hn = h->b_hn; ga.gi = GI; ga.gi_bf16 = i16 ? 1 : 0; ga.f16 = h->f16 ? 1 : 0; ga.h_raw_out = HRAW;
    ga.h_relu_out = defer_head ? (void*)(HRall + (size_t)base * rb.hr) : HR;        // the kernels index relu(h) by chunk-relative row
    ga.h_state = h->h_state; ga.hx = h->hx; ga.flags = h->flags; ga.abort_word = h->abort_word;
    ga.rowoff = h->d_rowoff; ga.nact = h->d_nact; ga.t0 = t0; ga.t1 = t1; ga.row_base = base; ga.rows = rows;
    ga.keep_r = KR; ga.keep_z = KZ; ga.keep_n = KN; ga.keep_ghn = KG;
    ga.n_clips = n_slots; ga.G = h->G; ga.seg_off = h->plan_single ? nullptr : h->d_seg_off;
    ga.seg_start = h->plan_single ? nullptr : h->d_seg_start; ga.stamps = h->use_stamps ? h->stamps : nullptr;
    ga.sync = h->no_local ? nullptr : h->flags;   // flags[0..15] double as the rendezvous words
    ga.armed = (arm_fuse && rows > 0) ? 1 : 0;
    ga.no_mt = h->no_mt ? 1 : 0;
    ga.out_floor = h->layers == 2 ? -INFINITY : 0.f;      // 2 layers: layer 0 hands h_t itself to layer 1 (below)
    if (HR0) ga.h_relu_out = HR0;                         // ... and when training, into a buffer of its own (backward needs it again)
    {
      // PREGO_GRU_COMPACT=1 (experiments): live slots packed into the fewest groups, the other XCDs leave at once.  Default
      // off: spreading the live slots over all groups is 2.6 ms per pass faster (the step cost grows with the fullest group's columns)
      static const bool compact = prego_tune_env("PREGO_GRU_COMPACT") != nullptr;
      const int live0 = h->h_nact[t0];
      ga.Gd = (compact && live0 <= 16 * h->G) ? std::max(1, std::min(h->G, (live0 + 15) / 16)) : 0;
    }
    const bool prefetch_next = prefetch && t1 < h->t_max;
    // XCD overlap: when the live slots fit fewer than eight groups, this launch is compacted onto XCDs 0 .. Gd - 1
    // and the NEXT chunk's layer1 GEMM runs as a persistent worker on the side stream behind its pack: its workgroups can only be
    // dispatched where no recurrence workgroup is resident, i.e. on the free XCDs, until this launch ends; the tile queue balances
    bool ov = false;
    if (overlap_ok && prefetch_next && ci + 1 < 4096) {
      const int live0 = h->h_nact[t0];
      const int gd0 = (live0 + 15) / 16;
      const int rows_n = h->h_rowoff[chunk_end(t1)] - h->h_rowoff[t1];
      static const int max_gd = prego_tune_env("PREGO_OVERLAP_MAX_GD") ? atoi(prego_tune_env("PREGO_OVERLAP_MAX_GD")) : 7;     // A/B knob
      if (live0 <= 16 * h->G && gd0 < h->G && gd0 <= max_gd && rows_n >= 4096) {
        // how many groups?  The fewest (gd0) frees the most XCDs; more groups mean fewer columns per group and a faster step
        // (1.67 us + 0.0102 us per live column of the fullest group).  Take the widest spread that still leaves the worker enough
        // XCD-time for the whole layer1 GEMM of the next chunk (13 ns per row on the whole chip, probe: >= proportional on a part)
        const double l1_ms = rows_n * 13.0e-6;
        int pick = std::max(1, gd0);
        static const bool wide = prego_tune_env("PREGO_OVERLAP_NARROW") == nullptr;     // A/B knob: always the fewest groups (same device: 127.2 vs 125.5 ms)
        for (int g2 = h->G - 1; wide && g2 > pick; --g2) {
          const double rec_ms = (t1 - t0) * (1.67 + 0.0102 * ((live0 + g2 - 1) / g2)) * 1e-3;
          if (l1_ms * h->G / (h->G - g2) <= 0.85 * rec_ms) { pick = g2; break; }
        }
        ov = true; ga.Gd = pick;
      }
    }
    ev = ev_begin(h, 1, s);
    if (prefetch_next) HIPCHK(hipEventRecord(h->ev_fork, s));      // fork point: everything before the recurrence launch
    // clip tiles per group that are still alive at this launch's first step (nact never grows): later launches of a pass whose
    // slots have thinned out run the kernels for fewer tiles (fewer registers; one tile = the classic kernel)
    const int live_slots = h->h_nact[t0];
    const int nct_l = std::max(1, std::min(nct, (((live_slots + h->G - 1) / h->G) + 15) / 16));
    if (h->x2 ? launch_gru_recurrence_x2(H, nct_l, ga, h->x2_scale + 5, s) : launch_gru_recurrence(h->bf16, H, nct_l, ga, s))
      return fail(PREGO_EINVAL, "recurrence: unsupported hid=%d nct=%d", H, nct_l);
    if (h->placement < 0 && !h->place_pending && h->xcd_overlap && h->bf16 && h->G == 8 && !h->no_local && ga.Gd == 0) {
      // the first full-width launch writes the verified-placement word: mirror it to the host behind that launch
      HIPCHK(hipMemcpyAsync(h->pin_place, h->flags + 20, sizeof(unsigned), hipMemcpyDeviceToHost, s));
      HIPCHK(hipEventRecord(h->ev_place, s));
      h->place_pending = true;
    }
    ev_end(ev, s);
    if (prefetch_next) {
      // X is dead once the layer1 GEMM of this chunk has run: stream the next chunk's features into it while the recurrence
      // (latency-bound, one wave per SIMD) holds the CUs.  The recurrence is launched FIRST so that its 256 workgroups are
      // resident (placement rendezvous) before the copy's workgroups fill the wave slots
      HIPCHK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
      if (!early) pack_chunk(t1, chunk_end(t1), h->side, ci + 1);
      side_join.pending = true;
      if (ov) {
        const int rows_n = h->h_rowoff[chunk_end(t1)] - h->h_rowoff[t1];
        EventPair* evw = ev_begin(h, 3, h->side);       // kind 3: overlapped worker (its span includes waiting for the recurrence's XCDs)
        if (launch_gemm_bf16_pingpong_worker(X, kx, h->w1, din, h->b1, Y, E, rows_n, E, kx, 0, h->tile_ctr + (ci + 1), 256, h->side, true, h->f16) == 0)
          l1_done = true;
        ev_end(evw, h->side);
      }
      HIPCHK(hipEventRecord(h->ev_join, h->side));
      packed = true;
    }

    if (h->layers == 2) {
      // second GRU layer (nn.GRU num_layers = 2, rnn.py:38,61): its input is layer 0's h_t - the first launch stored h_t itself
      // (out_floor = -inf) in HR -, GI and HR are reused in place: gi' = h W_ih_l1^T + b (GI of layer 0 is dead), then the recurrence of
      // layer 1 over the same steps from its own state, relu(h'_t) -> HR for the classifier
      ev = ev_begin(h, 0, s);
      proj(HR0 ? HR0 : HR, H, h->l2_w_ih, H, h->l2_bias2, GI, 3 * H, rows, 3 * H, H);
      ev_end(ev, s);
      if (h->timing) h->gemm_flop += 2.0 * rows * 3.0 * H * H;
      GruArgs g2 = ga;
      g2.whh = h->l2_w_hh; g2.b_hn = h->l2_b_hn; g2.h_state = h->h_state + slot_stride; g2.out_floor = 0.f;
      if (HR0) { g2.h_relu_out = HR; g2.h_raw_out = HRAW2; g2.keep_r = KR2; g2.keep_z = KZ2; g2.keep_n = KN2; g2.keep_ghn = KG2; }
      g2.armed = 0;                 // the exchange buffers / rendezvous words were used by layer 0's launch: re-arm (launcher)
      g2.Gd = 0;
      ev = ev_begin(h, 1, s);
      if (launch_gru_recurrence(h->bf16, H, nct_l, g2, s)) return fail(PREGO_EINVAL, "recurrence (layer 1): unsupported hid=%d nct=%d", H, nct_l);
      ev_end(ev, s);
    }
    if ((out || argmax) && !defer_head) {
      if (launch_head_softmax(h->bf16, HR, h->w_c, h->b_c, plan, base, rows, H, h->ncls,
                              (flags & PREGO_FWD_SOFTMAX) ? 1 : 0, d_out_ptrs, d_arg_ptrs, s, RM + (size_t)(ci & 1) * cap_rows * 8, h->f16))
        return fail(PREGO_EINVAL, "head: unsupported num_classes %d", h->ncls);
    }
    if (packed) { HIPCHK(hipStreamWaitEvent(s, h->ev_join, 0)); side_join.pending = false; }
    t0 = t1;
    ++ci;
  }
  if (defer_head) {                      // the classifier of the whole call, once (16-bit operands: no row map at hand - the kernel looks rows up in the plan)
    if (launch_head_softmax(h->bf16, HRall, h->w_c, h->b_c, plan, 0, total_rows, H, h->ncls, (flags & PREGO_FWD_SOFTMAX) ? 1 : 0, d_out_ptrs,
                            d_arg_ptrs, s, nullptr, h->f16))
      return fail(PREGO_EINVAL, "head: unsupported num_classes %d", h->ncls);
  }
  if (h_last)
    for (int l = 0; l < h->layers; ++l) launch_permute_rows(h->h_state + l * slot_stride, h_last + (size_t)l * n_clips * H, h->d_sorted, n_slots, H, 0, s);
  if (hostfeat && (h->feed_upto.empty() || h->feed_upto.back() < h->t_max))
    return fail(PREGO_EINVAL, "feed events cover steps < %d, the call has %d", h->feed_upto.empty() ? 0 : h->feed_upto.back(), h->t_max);
  if (h->meas_armed) { HIPCHK(hipEventRecord(h->ev_meas[1], s)); h->meas_pending = true; h->meas_armed = false; }
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// streaming step: one frame for each of n <= 16 streams (stream_step.hip)
extern "C" int prego_miniroad_step(prego_miniroad* h, int n_streams, const float* rgb, const float* flow, float* h_state, float* out,
                                   int32_t* argmax, int flags, prego_stream_t stream) {
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  if (!h->have_weights) return fail(PREGO_EINVAL, "step before set_weights");
  if (!h->bf16) return fail(PREGO_EINVAL, "step: the streaming fast path takes bf16 / fp16 handles (fp32 / fp16x2 operands: use forward() with h0 / h_last)");
  if (h->hid != 1024 || h->layers != 1)
    return fail(PREGO_EINVAL, "step: the streaming kernels are built for hidden_dim 1024, one GRU layer (hidden_dim %d, %d layers: use forward() with h0 / h_last)", h->hid, h->layers);
  if (n_streams < 1 || n_streams > 16) return fail(PREGO_EINVAL, "step: %d streams (1..16 per call)", n_streams);
  if (!h_state) return fail(PREGO_EINVAL, "step: h_state is NULL");
  if (h->d_rgb > 0 && !rgb) return fail(PREGO_EINVAL, "step: rgb is NULL");
  if (h->d_rgb == 0 && !flow) return fail(PREGO_EINVAL, "a model without rgb features (--no_rgb) needs the flow frame");
  hipStream_t s = (hipStream_t)stream;
  const int E = h->emb, H = h->hid, din = h->d_rgb + h->d_flow;
  float* Y = (float*)h->st_scratch;
  void* Eb = h->st_scratch + (size_t)16 * E * 4;
  float* GI = (float*)(h->st_scratch + (size_t)16 * E * 6);
  float* GH = GI + (size_t)16 * 3 * H;
  const bool with_flow = flow != nullptr && h->d_flow > 0;
  // layer1: K = the columns actually present (a zero flow half drops its half of K, as in forward())
  StreamGemv l1{h->w1, rgb, with_flow ? flow : nullptr, h->b1, Y, E, din, h->d_rgb, h->d_rgb, h->d_flow, 0};
  if (launch_stream_gemv(1, &l1, n_streams, s, h->f16)) return fail(PREGO_EINVAL, "step: unsupported layer1 shape %d x %d", E, din);
  // LayerNorm + ReLU: inside the W_ih product for <= 4 streams (three launches per frame), the batched kernel otherwise
  static const bool no_fuse = prego_tune_env("PREGO_STEP_NO_LN_FUSE") != nullptr;
  const bool fuse_ln = n_streams <= 4 && E % 2048 == 0 && !no_fuse;
  if (!fuse_ln) launch_ln_relu(true, Y, h->ln_g, h->ln_b, n_streams, E, 1e-5f, Eb, nullptr, 0.f, 0ull, 0, s, 1, false, h->f16);
  StreamGemv g2[2] = {{h->w_ih, fuse_ln ? (const void*)Y : (const void*)Eb, nullptr, h->bias2, GI, 3 * H, E, E, E, 0, fuse_ln ? 0 : 1},
                      {h->w_hh, h_state, nullptr, nullptr, GH, 3 * H, H, H, H, 0, 0}};
  if (fuse_ln) { g2[0].ln_g = h->ln_g; g2[0].ln_b = h->ln_b; g2[0].ln_eps = 1e-5f; }
  if (launch_stream_gemv(2, g2, n_streams, s, h->f16)) return fail(PREGO_EINVAL, "step: unsupported GRU shape %d / %d", E, H);
  if (launch_stream_gates_head(GI, GH, h->b_hn, h_state, h->w_c, h->b_c, n_streams, H, h->ncls, (flags & PREGO_FWD_SOFTMAX) ? 1 : 0, out,
                               (int*)argmax, s, h->f16)) return fail(PREGO_EINVAL, "step: unsupported head shape %d x %d", h->ncls, H);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

extern "C" int prego_miniroad_check(prego_miniroad* h, prego_stream_t stream) {
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  unsigned ab = 0;
  HIPCHK(hipMemcpy(&ab, h->abort_word, sizeof ab, hipMemcpyDeviceToHost));
  if (ab) {
    (void)hipMemset(h->abort_word, 0, sizeof ab);
    // codes: 1 = the recurrence's gather / rendezvous; 3 = the recurrence of a split pass
    // waiting for its input projection; 0x100 + k = wait k of the feed-forward launch of a split pass (ff_pass.hip)
    if (ab == 0x200u)       // prego_miniroad_set_peer_guard: raised by the guarded AdamW step, not by a kernel of this handle
      return fail(PREGO_ETIMEOUT, "data-parallel training: a recurrence / BPTT kernel of ANOTHER rank timed out; every rank skipped the optimizer "
                  "steps from that one on (weights unchanged since the last good step) [code 0x200]");
    if (ab >= 2) {
      // a wait INSIDE a split pass ran out although its start handshake had seen both launches resident (launches that cannot run side by
      // side never get this far: they leave at the handshake and the call is re-run chunked, prego_miniroad_forward).  A stuck workgroup or a
      // bug: report it, keep this handle on the chunked pass
      h->split_env = 0;
      return fail(PREGO_ETIMEOUT, "split pass: a wait timed out behind a successful start handshake [code 0x%x]; the results of that call are "
                  "invalid, this handle now uses the chunked pass (PREGO_SPLIT_PASS=0 selects it from the start)", ab);
    }
    return fail(PREGO_ETIMEOUT, "GRU recurrence kernel timed out waiting for a producer workgroup (not all %d workgroups resident?) [code 0x%x]",
                h->G * h->P, ab);
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
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  double ms[5] = {0, 0, 0, 0, 0};      // kinds: 0 gemm, 1 recurrence, 2 pack, 3 overlapped layer1 worker, 4 feed-forward launch of a split pass
  int64_t n[5] = {0, 0, 0, 0, 0};
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

extern "C" int prego_miniroad_pass_info(const prego_miniroad* h, int32_t* mode, int32_t* n_steps, int32_t* n_slots) {
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  if (mode) *mode = h->split_r;
  if (n_steps) *n_steps = h->t_max;
  if (n_slots) *n_slots = h->n_slots;
  return PREGO_OK;
}

#ifdef PREGO_DEBUG_ABI
// debug / probe: ONLY the recurrence kernel, one launch over n_steps steps of n_slots equally long slots dealt to `gd` groups
// (0 = all), on caller-supplied gi rows [n_steps * n_slots][3H] (16-bit, the handle's operand type) -> relu(h) [rows][H].
// scripts/probes/xcd_overlap_probe.py runs it beside an XCD-filtered GEMM worker.
extern "C" int prego_debug_recurrence_only(prego_miniroad* h, int n_slots, int n_steps, int gd, const void* gi, void* h_relu,
                                           prego_stream_t stream) {
  HandleScope scope_(h);
  if (!h || !gi || !h_relu) return fail(PREGO_EINVAL, "NULL argument");
  if (!h->have_weights || !h->bf16) return fail(PREGO_EINVAL, "debug recurrence: a bf16 / fp16 handle with weights");
  if (n_slots < 1 || n_slots > 16 * h->G || n_steps < 1) return fail(PREGO_EINVAL, "debug recurrence: %d slots, %d steps", n_slots, n_steps);
  hipStream_t s = (hipStream_t)stream;
  std::vector<int32_t> lens((size_t)n_slots, n_steps);
  int rc = build_plan(h, n_slots, lens.data(), true);
  if (rc) return rc;
  std::vector<const void*> tab((size_t)4 * max_clips_of(h), nullptr);
  rc = stage_tables(h, tab.data(), tab.size(), s);
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(h->h_state, 0, (size_t)n_slots * h->hid * 4, s));
  GruArgs ga{};
  ga.whh = h->w_hh; ga.b_hn = h->b_hn; ga.gi = gi; ga.gi_bf16 = 1; ga.f16 = h->f16 ? 1 : 0; ga.h_relu_out = h_relu; ga.h_raw_out = nullptr;
  ga.h_state = h->h_state; ga.hx = h->hx; ga.flags = h->flags; ga.abort_word = h->abort_word;
  ga.rowoff = h->d_rowoff; ga.nact = h->d_nact; ga.t0 = 0; ga.t1 = n_steps; ga.row_base = 0; ga.rows = n_slots * n_steps;
  ga.n_clips = n_slots; ga.G = h->G; ga.Gd = gd; ga.seg_off = nullptr; ga.seg_start = nullptr; ga.stamps = nullptr;
  ga.sync = h->no_local ? nullptr : h->flags;
  if (launch_gru_recurrence(true, h->hid, 1, ga, s)) return fail(PREGO_EINVAL, "debug recurrence: launch failed");
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// debug / probe: ONLY the head kernel (relu(h) rows -> probabilities + argmax) over n_slots equal slots x n_steps steps;
// out [n_slots][n_steps][C] fp32, argmax [n_slots][n_steps].  scripts/probes/head_probe.py
extern "C" int prego_debug_head_only(prego_miniroad* h, int n_slots, int n_steps, const void* h_relu, float* out, int32_t* argmax,
                                     const void* rowmap, prego_stream_t stream) {
  HandleScope scope_(h);
  if (!h || !h_relu || !out || !argmax) return fail(PREGO_EINVAL, "NULL argument");
  if (!h->have_weights || !h->bf16) return fail(PREGO_EINVAL, "debug head: a bf16 / fp16 handle with weights");
  if (n_slots < 1 || n_slots > max_clips_of(h) || n_steps < 1) return fail(PREGO_EINVAL, "debug head: %d slots, %d steps", n_slots, n_steps);
  hipStream_t s = (hipStream_t)stream;
  std::vector<int32_t> lens((size_t)n_slots, n_steps);
  int rc = build_plan(h, n_slots, lens.data(), true);
  if (rc) return rc;
  const SlotPlan plan = device_plan(h);
  const int MC = max_clips_of(h);
  std::vector<const void*> tab((size_t)4 * MC, nullptr);
  for (int i = 0; i < n_slots; ++i) {
    tab[2 * MC + i] = out + (size_t)i * n_steps * h->ncls;
    tab[3 * MC + i] = argmax + (size_t)i * n_steps;
  }
  rc = stage_tables(h, tab.data(), tab.size(), s);
  if (rc) return rc;
  if (launch_head_softmax(true, h_relu, h->w_c, h->b_c, plan, 0, n_slots * n_steps, h->hid, h->ncls, 1, (float* const*)(h->d_ptrs + 2 * MC),
                          (int* const*)(h->d_ptrs + 3 * MC), s, rowmap, h->f16))
    return fail(PREGO_EINVAL, "debug head: unsupported num_classes %d", h->ncls);
  HIPCHK(hipGetLastError());
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

// fault injection / replay for the NEXT split pass of the handle (one shot; include/prego_amd_debug.h)
extern "C" int prego_debug_split_fault(prego_miniroad* h, int mode) {
  if (!h || mode < 0 || mode > 4) return fail(PREGO_EINVAL, "debug split fault: mode %d", mode);
  h->dbg_fault = mode;
  return PREGO_OK;
}
// unit-test hook: set the handle's timeout word as a kernel that gave up would (stream-ordered)
extern "C" int prego_debug_set_abort(prego_miniroad* h, unsigned value, prego_stream_t stream) {
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  HIPCHK(hipMemsetD32Async((hipDeviceptr_t)h->abort_word, (int)value, 1, (hipStream_t)stream));
  return PREGO_OK;
}
extern "C" int prego_debug_split_state(const prego_miniroad* h, int64_t* fallbacks, int32_t* fails, int64_t* skip, int32_t* split_env) {
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  if (fallbacks) *fallbacks = h->split_fallbacks;
  if (fails) *fails = h->split_fails;
  if (skip) *skip = h->split_skip;
  if (split_env) *split_env = h->split_env;
  return PREGO_OK;
}
#endif  // PREGO_DEBUG_ABI

// ================================================================================================
// training: dropout control, loss, backward
// ================================================================================================
extern "C" int prego_miniroad_set_dropout(prego_miniroad* h, float p, uint64_t seed) {
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  if (!(p >= 0.f && p < 1.f)) return fail(PREGO_EINVAL, "dropout p = %f", (double)p);
  h->drop_p = p;
  h->drop_seed = seed;
  return PREGO_OK;
}

// handle-free: OadLoss is a criterion object of its own in the reference (criterions/loss_builder.py:9-11).
// Scratch for the pointer tables is one small per-device allocation made on first use.
#define LOSS_MAX_CLIPS 4096
extern "C" int prego_oad_loss(int n_clips, const int32_t* lens, const float* const* logits, const float* const* target,
                              int n_classes, float* loss_out, float* const* dlogits, float grad_scale,
                              prego_stream_t stream) {
  return prego_oad_loss_reduce(n_clips, lens, logits, target, n_classes, 0, loss_out, dlogits, grad_scale, stream);
}
extern "C" int prego_oad_loss_reduce(int n_clips, const int32_t* lens, const float* const* logits, const float* const* target,
                                     int n_classes, int reduction, float* loss_out, float* const* dlogits, float grad_scale,
                                     prego_stream_t stream) {
  if (reduction != 0 && reduction != 1) return fail(PREGO_EINVAL, "loss: reduction %d (0 = 'mean', 1 = 'sum')", reduction);
  if (!lens || !logits || !target || !loss_out) return fail(PREGO_EINVAL, "loss: NULL argument");
  if (n_clips <= 0 || n_clips > LOSS_MAX_CLIPS) return fail(PREGO_EINVAL, "loss: %d clips (max %d)", n_clips, LOSS_MAX_CLIPS);
  if (n_classes <= 0 || n_classes > 128) return fail(PREGO_EINVAL, "loss: num_classes %d must be in 1..128", n_classes);
  // per-device scratch of this handle-free op: device pointer tables + a PINNED host staging copy fenced by an event (the
  // async H2D copy reads the staging buffer after this call has returned, so it is neither a stack nor a pageable buffer)
  struct LossScratch { void* dev = nullptr; void* pin = nullptr; hipEvent_t ev = nullptr; bool busy = false; };
  static LossScratch scratch[64];
  static std::mutex scratch_mu;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return fail(PREGO_EINVAL, "device %d", dev);
  const size_t MC = LOSS_MAX_CLIPS;
  std::lock_guard<std::mutex> guard(scratch_mu);
  LossScratch& sc = scratch[dev];
  if (!sc.dev) {
    HIPCHK(hipMalloc(&sc.dev, 4 * MC * sizeof(void*)));
    HIPCHK(hipHostMalloc(&sc.pin, 4 * MC * sizeof(void*), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&sc.ev, hipEventDisableTiming));
  }
  if (sc.busy) { HIPCHK(hipEventSynchronize(sc.ev)); sc.busy = false; }
  void** d = (void**)sc.dev;
  hipStream_t s = (hipStream_t)stream;
  const void** tab = (const void**)sc.pin;
  // the four tables packed one behind the other (n_clips entries each): ONE host -> device copy per call (four copies of 128 bytes
  // were four 5 us blit launches in front of the loss kernel of every training step)
  const size_t n = (size_t)n_clips;
  for (int i = 0; i < n_clips; ++i) {
    if (lens[i] <= 0 || !logits[i] || !target[i]) return fail(PREGO_EINVAL, "loss: clip %d", i);
    tab[0 * n + i] = logits[i]; tab[1 * n + i] = target[i]; tab[2 * n + i] = dlogits ? dlogits[i] : nullptr;
  }
  std::memcpy(&tab[3 * n], lens, n * 4);                       // 4th table doubles as the lens array
  HIPCHK(hipMemcpyAsync(d, tab, 4 * n * sizeof(void*), hipMemcpyHostToDevice, s));
  HIPCHK(hipEventRecord(sc.ev, s));
  sc.busy = true;
  launch_oad_loss((const float* const*)d, (const float* const*)(d + n), (const int*)(d + 3 * n), n_clips, n_classes,
                  loss_out, dlogits ? (float* const*)(d + 2 * n) : nullptr, grad_scale, s, reduction == 1);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

struct BwdLayout {
  size_t total;
  size_t dLp, dLf, dLt, HRt, WcT, dWc, dHR, carry, dhpart, WhhT, dGI, dGH, dGIop, dGHop, part, T1, T2, WihT, dE, dY, dYb, Hprev, vec, bhx, bsync;
};
static BwdLayout bwd_layout(const prego_miniroad* h, int R, int n_clips) {
  const size_t es = h->bf16 ? 2 : 4;
  const size_t Rp = align_up((size_t)R, 64), H = h->hid, E = h->emb, Din = h->d_rgb + h->d_flow, Cp = 128;
  const size_t Bp = align_up((size_t)n_clips, 16);
  BwdLayout L{};
  size_t off = 0;
  auto put = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  L.dLp = put((size_t)R * Cp * es); L.dLf = put((size_t)R * Cp * 4); L.dLt = put(Cp * Rp * es);
  L.HRt = put(H * Rp * es); L.WcT = put(H * Cp * es); L.dWc = put(Cp * H * 4);
  L.dHR = put((size_t)R * H * 4);
  L.carry = put(2 * Bp * H * 4); L.dhpart = put(Bp * H * 4);
  L.WhhT = put(H * 3 * H * es);
  L.dGI = put((size_t)R * 3 * H * 4); L.dGH = put((size_t)R * 3 * H * 4);
  L.dGIop = put((size_t)R * 3 * H * es); L.dGHop = put((size_t)R * 3 * H * es);
  L.part = put(std::max<size_t>(((size_t)R / 64 + 1) * 3 * H, ((size_t)R / 4 + 1) * 2 * E) * 4);
  L.T1 = put(std::max<size_t>(3 * H, E) * Rp * es);           // transposed "A" operand of a wgrad (dGIt / dGHt / dYt)
  L.T2 = put(std::max<size_t>(std::max<size_t>(E, H), Din) * Rp * es);   // transposed "B" operand (Et / Hprev_t / Xt)
  L.WihT = put(std::max(E, H) * 3 * H * es);
  L.dE = put((size_t)R * E * 4); L.dY = put((size_t)R * E * 4);
  L.dYb = put(Rp * E * 2);                                      // bf16 copy of dY: k-major A operand of layer1's wgrad
  L.Hprev = put((size_t)R * H * es);
  L.vec = put(4 * E * 4);
  L.bhx = put(gru_bptt_hx_bytes(h->bf16, h->hid, h->G)); L.bsync = put(1024 * 4);     // persistent BPTT: exchange buffers, step counters
  L.total = off;
  return L;
}

extern "C" int prego_miniroad_set_gru_layer_grads(prego_miniroad* h, int layer, float* g_w_ih, float* g_w_hh, float* g_b_ih, float* g_b_hh) {
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  if (layer != 1 || h->layers != 2) return fail(PREGO_EINVAL, "set_gru_layer_grads: layer %d of a %d-layer handle (layer 0's gradients are prego_miniroad_backward's own arguments)", layer, h->layers);
  if (!g_w_ih || !g_w_hh || !g_b_ih || !g_b_hh) return fail(PREGO_EINVAL, "set_gru_layer_grads: NULL tensor");
  h->g_l2[0] = g_w_ih; h->g_l2[1] = g_w_hh; h->g_l2[2] = g_b_ih; h->g_l2[3] = g_b_hh;
  return PREGO_OK;
}

extern "C" int prego_miniroad_backward_events(prego_miniroad* h, void* ev_head_done, void* ev_gru_done) {
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  h->bwd_ev[0] = (hipEvent_t)ev_head_done;
  h->bwd_ev[1] = (hipEvent_t)ev_gru_done;
  return PREGO_OK;
}

extern "C" int prego_miniroad_backward_callback(prego_miniroad* h, prego_bucket_fn fn, void* user) {
  HandleScope scope_(h);
  if (!h) return fail(PREGO_EINVAL, "handle is NULL");
  h->bwd_cb = fn;
  h->bwd_cb_user = user;
  return PREGO_OK;
}

extern "C" size_t prego_miniroad_backward_workspace_bytes(const prego_miniroad* h, int n_clips, const int32_t* lens) {
  if (!h || n_clips <= 0 || !lens) return 0;
  long long total = 0;
  for (int i = 0; i < n_clips; ++i) total += lens[i];
  return bwd_layout(h, (int)total, n_clips).total;
}

static void gemm_nt(const prego_miniroad* h, const void* A, int lda, const void* B, int ldb, const float* bias, float* C,
                    int ldc, int M, int N, int K, hipStream_t s) {
  if (h->bf16) launch_gemm_bf16_nt(A, lda, B, ldb, bias, C, ldc, M, N, K, s);
  else launch_gemm_f32_nt((const float*)A, lda, (const float*)B, ldb, bias, C, ldc, M, N, K, s);
}

extern "C" int prego_miniroad_backward(prego_miniroad* h, int n_clips, const int32_t* lens, const float* const* dlogits,
                                       float* g_layer1_w, float* g_layer1_b, float* g_ln_w, float* g_ln_b, float* g_w_ih,
                                       float* g_w_hh, float* g_b_ih, float* g_b_hh, float* g_fc_w, float* g_fc_b,
                                       void* fwd_workspace, size_t fwd_bytes, void* bwd_workspace, size_t bwd_bytes,
                                       prego_stream_t stream) {
  HandleScope scope_(h);
  if (!h || !lens || !dlogits || !fwd_workspace || !bwd_workspace) return fail(PREGO_EINVAL, "backward: NULL argument");
  if (!g_layer1_w || !g_layer1_b || !g_ln_w || !g_ln_b || !g_w_ih || !g_w_hh || !g_b_ih || !g_b_hh || !g_fc_w || !g_fc_b)
    return fail(PREGO_EINVAL, "backward: NULL gradient tensor");
  if (h->f16 || h->x2) return fail(PREGO_EINVAL, "backward on an fp16 / fp16x2-operand handle: training runs on bf16 / fp32 handles");
  if ((int)h->plan_lens.size() != n_clips || !std::equal(lens, lens + n_clips, h->plan_lens.begin()) || h->kept_rows == 0)
    return fail(PREGO_EINVAL, "backward must follow a forward(PREGO_FWD_KEEP) of the same clips");
  hipStream_t s = (hipStream_t)stream;
  const bool bf = h->bf16;
  const size_t es = bf ? 2 : 4;
  const int H = h->hid, E = h->emb, C = h->ncls, Cp = 128, din = h->d_rgb + h->d_flow, kx = h->kept_kx;
  const int R = h->h_rowoff[h->t_max];
  const int Rp = (int)align_up((size_t)R, 64);
  // forward workspace carve (must mirror prego_miniroad_forward with PREGO_FWD_KEEP)
  const RowBytes rb = row_bytes(h, kx > h->d_rgb, PREGO_FWD_KEEP);
  const long long cap_rows = (long long)((fwd_bytes - 12 * 256) / rb.total) / 128 * 128;
  if (cap_rows < R) return fail(PREGO_EWORKSPACE, "forward workspace does not hold the kept activations");
  char* wp = (char*)fwd_workspace;
  auto carve = [&](size_t bytes) { char* p = wp; wp += align_up(bytes, 256); return (void*)p; };
  void* X = carve((size_t)cap_rows * rb.x);
  float* Y = (float*)carve((size_t)cap_rows * rb.y);
  void* Eb = carve((size_t)cap_rows * rb.e);
  (void)carve((size_t)cap_rows * rb.gi);
  void* HR = carve((size_t)cap_rows * rb.hr);
  float* HRAW = (float*)carve((size_t)cap_rows * rb.hraw);
  float* KR = (float*)carve((size_t)cap_rows * H * 4); float* KZ = (float*)carve((size_t)cap_rows * H * 4);
  float* KN = (float*)carve((size_t)cap_rows * H * 4); float* KG = (float*)carve((size_t)cap_rows * H * 4);
  float* STATS = (float*)carve((size_t)cap_rows * 8);
  void* HR0 = nullptr; float* HRAW2 = nullptr; float* KR2 = nullptr; float* KZ2 = nullptr; float* KN2 = nullptr; float* KG2 = nullptr;
  if (h->layers == 2) {
    if (!h->g_l2[0] || !h->g_l2[1] || !h->g_l2[2] || !h->g_l2[3])
      return fail(PREGO_EINVAL, "backward of a 2-layer handle before prego_miniroad_set_gru_layer_grads(1, ...)");
    HR0 = carve((size_t)cap_rows * H * es);
    HRAW2 = (float*)carve((size_t)cap_rows * H * 4);
    KR2 = (float*)carve((size_t)cap_rows * H * 4); KZ2 = (float*)carve((size_t)cap_rows * H * 4);
    KN2 = (float*)carve((size_t)cap_rows * H * 4); KG2 = (float*)carve((size_t)cap_rows * H * 4);
  }
  const BwdLayout L = bwd_layout(h, R, n_clips);
  if (bwd_bytes < L.total) return fail(PREGO_EWORKSPACE, "backward workspace %zu < %zu", bwd_bytes, L.total);
  char* bw = (char*)bwd_workspace;
  float* part = (float*)(bw + L.part);

  // dlogits pointer table
  const int MC = max_clips_of(h);
  std::vector<const void*> tab((size_t)MC, nullptr);
  for (int i = 0; i < n_clips; ++i) { if (!dlogits[i]) return fail(PREGO_EINVAL, "dlogits[%d] is NULL", i); tab[i] = dlogits[i]; }
  { const int rc = stage_tables(h, tab.data(), tab.size(), s); if (rc) return rc; }
  const float* const* d_dl = (const float* const*)h->d_ptrs;

  // ---- head: logits = relu(h) Wc^T + bc  (rnn.py:62-64)
  // bf16 handles (round 4): every wgrad / dgrad below runs on the k-major GEMM (gemm_tn.hip: operands staged as they lie in memory,
  // fragments read transposed from LDS, the bias gradient as one more MFMA per k-step) - no transposed copy of any activation or
  // weight, no separate column-sum launches.  K of a wgrad = the packed rows, padded to 64 by reading zeros (k_valid = R).
  // fp32 handles keep the transpose + NT-GEMM + two-stage column-sum path (exact-fp32 MFMA, fixed-order fp32 sums).
  const bool tn = bf;
  launch_gather_dlogits(bf, d_dl, h->d_rowoff, h->d_sorted, h->t_max, R, C, Cp, bw + L.dLp, s);
  if (tn) {
    // dWc [C][H] = dL^T . relu(h), db_c = colsum(dL): straight into the caller's gradient tensors
    if (launch_gemm_bf16_tn(true, true, bw + L.dLp, Cp, HR, H, nullptr, g_fc_w, H, C, H, Rp, R, g_fc_b, s)) return fail(PREGO_EINVAL, "backward: head wgrad shape");
  } else {
    launch_gather_dlogits(false, d_dl, h->d_rowoff, h->d_sorted, h->t_max, R, C, Cp, bw + L.dLf, s);
    launch_colsum((const float*)(bw + L.dLf), R, Cp, part, (float*)(bw + L.vec), s);
    HIPCHK(hipMemcpyAsync(g_fc_b, bw + L.vec, (size_t)C * 4, hipMemcpyDeviceToDevice, s));
    launch_transpose_convert(bf, bf, bw + L.dLp, R, Cp, Cp, bw + L.dLt, Rp, s);            // [Cp][Rp]
    launch_transpose_convert(bf, bf, HR, R, H, H, bw + L.HRt, Rp, s);                       // [H][Rp]
    gemm_nt(h, bw + L.dLt, Rp, bw + L.HRt, Rp, nullptr, (float*)(bw + L.dWc), H, Cp, H, Rp, s);   // dWc[Cp][H]
    HIPCHK(hipMemcpyAsync(g_fc_w, bw + L.dWc, (size_t)C * H * 4, hipMemcpyDeviceToDevice, s));
  }
  if (h->bwd_ev[0]) HIPCHK(hipEventRecord(h->bwd_ev[0], s));            // f_classification gradients are final
  if (h->bwd_cb) h->bwd_cb(h->bwd_cb_user, 0);
  if (tn) {
    // d relu(h) [R][H] = dL [R][Cp] . Wc [ncls_pad][H]: the weight as it is stored ([K][N]); rows >= ncls_pad read as zeros
    if (launch_gemm_bf16_tn(false, true, bw + L.dLp, Cp, h->w_c, H, nullptr, (float*)(bw + L.dHR), H, R, H, Cp, h->ncls_pad, nullptr, s))
      return fail(PREGO_EINVAL, "backward: head dgrad shape");
  } else {
    launch_transpose_convert(bf, bf, h->w_c, h->ncls_pad, H, H, bw + L.WcT, Cp, s);        // [H][Cp] (rows >= ncls_pad zero)
    gemm_nt(h, bw + L.dLp, Cp, bw + L.WcT, Cp, nullptr, (float*)(bw + L.dHR), H, R, H, Cp, s);    // d relu(h)
  }
  launch_relu_mask((const float*)(bw + L.dHR), h->layers == 2 ? HRAW2 : HRAW, (size_t)R * H, (float*)(bw + L.dHR), s);   // the head reads the LAST layer's relu(h)

  // ---- BPTT through the GRU (rnn.py:61), reverse time; a stacked GRU (num_layers 2, rnn.py:32,38) runs its layers last to first:
  // layer 1 from the head's gradient, then dH0 = dGI1 . W_ih_l1 (no relu between the layers), then layer 0 from that
  const size_t Bp = align_up((size_t)n_clips, 16);
  for (int layer = h->layers - 1; layer >= 0; --layer) {
  const void* L_whh = layer == 1 ? h->l2_w_hh : h->w_hh;
  const void* L_wih = layer == 1 ? h->l2_w_ih : h->w_ih;
  const void* L_in = layer == 1 ? HR0 : Eb;                  // the layer's input rows (operand type)
  const int L_k = layer == 1 ? H : E;                        // ... and their width
  float* L_hraw = layer == 1 ? HRAW2 : HRAW;
  float* L_kr = layer == 1 ? KR2 : KR; float* L_kz = layer == 1 ? KZ2 : KZ; float* L_kn = layer == 1 ? KN2 : KN; float* L_kg = layer == 1 ? KG2 : KG;
  float* L_gwih = layer == 1 ? h->g_l2[0] : g_w_ih; float* L_gwhh = layer == 1 ? h->g_l2[1] : g_w_hh;
  float* L_gbih = layer == 1 ? h->g_l2[2] : g_b_ih; float* L_gbhh = layer == 1 ? h->g_l2[3] : g_b_hh;
  const bool last_layer = layer == 0;
  launch_transpose_convert(bf, bf, L_whh, 3 * H, H, H, bw + L.WhhT, 3 * H, s);          // [H][3H]
  float* carry[2] = {(float*)(bw + L.carry), (float*)(bw + L.carry) + Bp * H};
  float* dhpart = (float*)(bw + L.dhpart);
  // one persistent launch (gru_bptt.hip); the step-by-step loop below is the fallback for shapes it does not take and the
  // A/B reference (PREGO_BPTT_STEPWISE=1)
  bool persistent = false;
  {
    static const bool stepwise = prego_tune_env("PREGO_BPTT_STEPWISE") != nullptr;
    const int slots = (h->n_slots + h->G - 1) / h->G;
    const int nct = (slots + 15) / 16;
    if (!stepwise) {
      BpttArgs ba;
      ba.whhT = bw + L.WhhT; ba.dHout = (const float*)(bw + L.dHR); ba.R = L_kr; ba.Z = L_kz; ba.N = L_kn; ba.GHN = L_kg; ba.Hraw = L_hraw;
      // the fp32 copies of dGI / dGH are read by the exact-fp32 path only (column sums, transposes): a bf16 handle's k-major GEMMs take
      // the operand copies, so its BPTT kernel does not store them at all
      ba.dGI = tn ? nullptr : (float*)(bw + L.dGI); ba.dGH = tn ? nullptr : (float*)(bw + L.dGH); ba.dGIop = bw + L.dGIop; ba.dGHop = bw + L.dGHop;
      ba.hx = bw + L.bhx; ba.sync = (unsigned*)(bw + L.bsync); ba.abort_word = h->abort_word;
      ba.rowoff = h->d_rowoff; ba.nact = h->d_nact; ba.t_max = h->t_max; ba.n_clips = h->n_slots; ba.G = h->G;
      ba.force_sc1 = h->no_local ? 1 : 0;
      persistent = launch_gru_bptt(bf, H, nct, ba, s) == 0;
    }
  }
  for (int t = h->t_max - 1; t >= 0 && !persistent; --t) {
    const int na = h->h_nact[t];
    const int na_next = t + 1 < h->t_max ? h->h_nact[t + 1] : 0;
    const int row_t = h->h_rowoff[t], row_tm1 = t > 0 ? h->h_rowoff[t - 1] : 0;
    launch_gru_bwd_step(bf, t, na, na_next, row_t, row_tm1, H, (const float*)(bw + L.dHR), carry[(t + 1) & 1], dhpart, L_kr,
                        L_kz, L_kn, L_kg, L_hraw, carry[t & 1], (float*)(bw + L.dGI), (float*)(bw + L.dGH), bw + L.dGIop,
                        bw + L.dGHop, s);
    if (t > 0)   // dh_{t-1} += dgh_t . W_hh
      gemm_nt(h, bw + L.dGHop + (size_t)row_t * 3 * H * es, 3 * H, bw + L.WhhT, 3 * H, nullptr, dhpart, H, na, H, 3 * H, s);
  }
  launch_build_hprev(bf, L_hraw, h->d_rowoff, h->t_max, R, H, bw + L.Hprev, s);
  if (tn) {
    // dW_ih = dGI^T . e (+ db_ih), dW_hh = dGH^T . h_{t-1} (+ db_hh): bias sums from the bf16 operand copies the BPTT kernel wrote
    if (launch_gemm_bf16_tn(true, true, bw + L.dGIop, 3 * H, L_in, L_k, nullptr, L_gwih, L_k, 3 * H, L_k, Rp, R, L_gbih, s) ||
        launch_gemm_bf16_tn(true, true, bw + L.dGHop, 3 * H, bw + L.Hprev, H, nullptr, L_gwhh, H, 3 * H, H, Rp, R, L_gbhh, s))
      return fail(PREGO_EINVAL, "backward: GRU wgrad shape");
  } else {
    // biases of the GRU
    launch_colsum((const float*)(bw + L.dGI), R, 3 * H, part, L_gbih, s);
    launch_colsum((const float*)(bw + L.dGH), R, 3 * H, part, L_gbhh, s);
    // dW_ih = dGI^T . e
    launch_transpose_convert(bf, bf, bw + L.dGIop, R, 3 * H, 3 * H, bw + L.T1, Rp, s);
    launch_transpose_convert(bf, bf, L_in, R, L_k, L_k, bw + L.T2, Rp, s);
    gemm_nt(h, bw + L.T1, Rp, bw + L.T2, Rp, nullptr, L_gwih, L_k, 3 * H, L_k, Rp, s);
    // dW_hh = dGH^T . h_{t-1}
    launch_transpose_convert(bf, bf, bw + L.dGHop, R, 3 * H, 3 * H, bw + L.T1, Rp, s);
    launch_transpose_convert(bf, bf, bw + L.Hprev, R, H, H, bw + L.T2, Rp, s);
    gemm_nt(h, bw + L.T1, Rp, bw + L.T2, Rp, nullptr, L_gwhh, H, 3 * H, H, Rp, s);
  }
  if (last_layer) {
    if (h->bwd_ev[1]) HIPCHK(hipEventRecord(h->bwd_ev[1], s));          // every GRU gradient is final (layer1 / LayerNorm follow)
    if (h->bwd_cb) h->bwd_cb(h->bwd_cb_user, 1);
  }
  // gradient of the layer's input = dGI . W_ih: d e for layer 0 (LayerNorm's output), dH0 - the next BPTT's dHout, as it is - for layer 1
  float* L_din = last_layer ? (float*)(bw + L.dE) : (float*)(bw + L.dHR);
  if (tn) {
    if (launch_gemm_bf16_tn(false, true, bw + L.dGIop, 3 * H, L_wih, L_k, nullptr, L_din, L_k, R, L_k, 3 * H, 3 * H, nullptr, s))
      return fail(PREGO_EINVAL, "backward: W_ih dgrad shape");
  } else {
    launch_transpose_convert(bf, bf, L_wih, 3 * H, L_k, L_k, bw + L.WihT, 3 * H, s);        // [K][3H]
    gemm_nt(h, bw + L.dGIop, 3 * H, bw + L.WihT, 3 * H, nullptr, L_din, L_k, R, L_k, 3 * H, s);
  }
  }   // layers, last to first

  // ---- Dropout / ReLU / LayerNorm backward (rnn.py:41-43)
  const int nb = launch_ln_relu_bwd((const float*)(bw + L.dE), Y, STATS, h->ln_g, h->ln_b, R, E, h->drop_p, h->drop_seed, 0,
                                    (float*)(bw + L.dY), part, s, 1, 0, tn ? (void*)(bw + L.dYb) : nullptr);
  // d gamma | d beta: the stage-2 sum goes straight into the caller's two tensors when they are adjacent (the flat gradient bucket
  // of prego_amd/engine.py), through a scratch vector and two copies otherwise
  if (g_ln_b == g_ln_w + E) launch_colsum_stage2(part, nb, 2 * E, g_ln_w, s);
  else {
    launch_colsum_stage2(part, nb, 2 * E, (float*)(bw + L.vec), s);
    HIPCHK(hipMemcpyAsync(g_ln_w, bw + L.vec, (size_t)E * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(g_ln_b, bw + L.vec + (size_t)E * 4, (size_t)E * 4, hipMemcpyDeviceToDevice, s));
  }

  // ---- layer1 Linear (rnn.py:40): db = colsum(dY), dW = dY^T . x
  if (kx < din) HIPCHK(hipMemsetAsync(g_layer1_w, 0, (size_t)E * din * 4, s));          // zero-flow columns: zero gradient
  if (tn) {
    if (launch_gemm_bf16_tn(true, true, bw + L.dYb, E, X, kx, nullptr, g_layer1_w, din, E, kx, Rp, R, g_layer1_b, s))
      return fail(PREGO_EINVAL, "backward: layer1 wgrad shape");
  } else {
    launch_colsum((const float*)(bw + L.dY), R, E, part, g_layer1_b, s);
    launch_transpose_convert(false, bf, bw + L.dY, R, E, E, bw + L.T1, Rp, s);
    launch_transpose_convert(bf, bf, X, R, kx, kx, bw + L.T2, Rp, s);
    gemm_nt(h, bw + L.T1, Rp, bw + L.T2, Rp, nullptr, g_layer1_w, din, E, kx, Rp, s);
  }
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// ================================================================================================
// post-processing: utils/aggregate.py:55-72 (the 200-frame majority vote) on the device
// ================================================================================================
extern "C" int prego_window_vote(const int32_t* argmax, int64_t n_frames, int window, int n_classes, int32_t* votes,
                                 prego_stream_t stream) {
  if (!argmax || !votes) return fail(PREGO_EINVAL, "window_vote: NULL argument");
  if (launch_window_vote(argmax, n_frames, window, n_classes, votes, (hipStream_t)stream))
    return fail(PREGO_EINVAL, "window_vote: n_frames %lld, window %d, n_classes %d (1..128)", (long long)n_frames, window, n_classes);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

extern "C" int prego_format_ids(const int32_t* ids, int64_t n, uint32_t* text, int32_t* bad, prego_stream_t stream) {
  if (!ids || !text) return fail(PREGO_EINVAL, "format_ids: NULL argument");
  if (launch_format_ids(ids, n, text, bad, (hipStream_t)stream)) return fail(PREGO_EINVAL, "format_ids: n %lld", (long long)n);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// ================================================================================================
// metric: utils/metrics.py:25-62 (per-class average precision of the per-frame scores) on the device
// ================================================================================================
extern "C" size_t prego_perframe_ap_workspace_bytes(int64_t n_frames, int n_classes) {
  if (n_frames <= 0 || n_classes <= 0) return 0;
  return perframe_ap_workspace_bytes(n_frames, n_classes);
}
static int perframe_ap_common(const float* scores, const float* target, const int32_t* labels, int64_t n_frames, int n_classes, double* ap,
                              int64_t* n_pos, double* score_sum, void* workspace, size_t workspace_bytes, prego_stream_t stream) {
  if (!scores || (!target && !labels) || !ap || !workspace) return fail(PREGO_EINVAL, "perframe_ap: NULL argument");
  if (n_frames <= 0 || n_frames >= (1ll << 31) || n_classes <= 0 || n_classes > 65535)
    return fail(PREGO_EINVAL, "perframe_ap: n_frames %lld, n_classes %d", (long long)n_frames, n_classes);
  if (workspace_bytes < perframe_ap_workspace_bytes(n_frames, n_classes))
    return fail(PREGO_EWORKSPACE, "perframe_ap: workspace %zu < %zu", workspace_bytes, perframe_ap_workspace_bytes(n_frames, n_classes));
  if (launch_perframe_ap(scores, target, (const int*)labels, n_frames, n_classes, ap, (long long*)n_pos, score_sum, workspace, (hipStream_t)stream))
    return fail(PREGO_EINVAL, "perframe_ap: bad arguments");
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}
extern "C" int prego_perframe_ap(const float* scores, const float* target, int64_t n_frames, int n_classes, double* ap, int64_t* n_pos,
                                 double* score_sum, void* workspace, size_t workspace_bytes, prego_stream_t stream) {
  return perframe_ap_common(scores, target, nullptr, n_frames, n_classes, ap, n_pos, score_sum, workspace, workspace_bytes, stream);
}
extern "C" int prego_perframe_ap_labels(const float* scores, const int32_t* labels, int64_t n_frames, int n_classes, double* ap, int64_t* n_pos,
                                        double* score_sum, void* workspace, size_t workspace_bytes, prego_stream_t stream) {
  return perframe_ap_common(scores, nullptr, labels, n_frames, n_classes, ap, n_pos, score_sum, workspace, workspace_bytes, stream);
}

// The feeder's side of that: one-hot target rows (what the reference's dataset yields, dataset.py / eval.py:55 np.argmax(target)) reduced
// to their class id on the host, in the loader's own memory, by a few threads - while the GPU is busy with the features.  HOST function.
extern "C" int prego_onehot_labels(int n_videos, const float* const* targets, const int64_t* n_frames, int n_classes, int32_t* labels,
                                   int32_t* onehot) {
  if (n_videos < 0 || (n_videos && (!targets || !n_frames || !labels || !onehot)) || n_classes <= 0)
    return fail(PREGO_EINVAL, "onehot_labels: bad argument");
  std::vector<int64_t> off((size_t)n_videos + 1, 0);
  for (int v = 0; v < n_videos; ++v) {
    if (n_frames[v] < 0 || (n_frames[v] && !targets[v])) return fail(PREGO_EINVAL, "onehot_labels: video %d", v);
    off[(size_t)v + 1] = off[(size_t)v] + n_frames[v];
    onehot[v] = 1;
  }
  const int64_t total = off[(size_t)n_videos];
  const unsigned hw = std::thread::hardware_concurrency();
  const int nt = (int)std::min<int64_t>(std::max<int64_t>(1, total / 16384), std::min(16u, hw ? hw : 1u));
  auto work = [&](int t) {
    const int64_t a = total * t / nt, b = total * (t + 1) / nt;
    int v = (int)(std::upper_bound(off.begin(), off.end(), a) - off.begin()) - 1;
    for (int64_t i = a; i < b; ++i) {
      while (i >= off[(size_t)v + 1]) ++v;
      const float* row = targets[v] + (size_t)(i - off[(size_t)v]) * n_classes;
      int nz = 0, pos = 0;
      for (int c = 0; c < n_classes; ++c) { nz += row[c] != 0.f; pos += row[c] > 0.f; }
      int best = 0;                                            // np.argmax: the first maximum
      if (nz == 1 && pos == 1) { while (!(row[best] > 0.f)) ++best; }
      else {
        __atomic_store_n(&onehot[v], 0, __ATOMIC_RELAXED);
        for (int c = 1; c < n_classes; ++c) if (row[c] > row[best]) best = c;
      }
      labels[i] = best;
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
  if (total > 0) work(0);
  for (auto& x : th) x.join();
  return PREGO_OK;
}

// ================================================================================================
// optimizer: torch.optim.AdamW of main.py:62-67 on the ABI
// ================================================================================================
extern "C" int prego_adamw_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                                float* const* exp_avg_sq, const int64_t* numel, int64_t step, float lr, float beta1, float beta2,
                                float eps, float weight_decay, prego_stream_t stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel) return fail(PREGO_EINVAL, "adamw: NULL argument");
  std::vector<long long> n(numel, numel + std::max(n_tensors, 0));
  if (launch_adamw(n_tensors, params, grads, exp_avg, exp_avg_sq, nullptr, n.data(), false, step, lr, beta1, beta2, eps, weight_decay,
                   (hipStream_t)stream))
    return fail(PREGO_EINVAL, "adamw: bad tensor list (n = %d, step = %lld)", n_tensors, (long long)step);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// The same step for the ten MiniROAD tensors (prego_miniroad_set_weights' order) that ALSO refreshes the handle's operand copies
// (bf16 / fp32 weights, padded classifier, folded GRU biases) in the same pass: no set_weights call after the step.
extern "C" int prego_miniroad_adamw_step(prego_miniroad* h, float* const* params, const float* const* grads, float* const* exp_avg,
                                         float* const* exp_avg_sq, int64_t step, float lr, float beta1, float beta2, float eps,
                                         float weight_decay, prego_stream_t stream) {
  HandleScope scope_(h);
  if (!h || !params || !grads || !exp_avg || !exp_avg_sq) return fail(PREGO_EINVAL, "adamw: NULL argument");
  if (!h->have_weights) return fail(PREGO_EINVAL, "adamw step before set_weights");
  if (h->f16 || h->x2) return fail(PREGO_EINVAL, "adamw step on an fp16 / fp16x2-operand handle: training runs on bf16 / fp32 handles");
  hipStream_t s = (hipStream_t)stream;
  const long long din = h->d_rgb + h->d_flow, E = h->emb, H = h->hid, C = h->ncls;
  // set_weights order: layer1.0.weight, layer1.0.bias, layer1.1.weight, layer1.1.bias, w_ih, w_hh, b_ih, b_hh, fc.weight, fc.bias
  const long long numel[10] = {E * din, E, E, E, 3 * H * E, 3 * H * H, 3 * H, 3 * H, C * H, C};
  // operand-typed copies (weights) first, fp32 copies (biases, LayerNorm) second: two launches, one element type each
  float* pw[4] = {params[0], params[4], params[5], params[8]};
  const float* gw[4] = {grads[0], grads[4], grads[5], grads[8]};
  float* mw[4] = {exp_avg[0], exp_avg[4], exp_avg[5], exp_avg[8]};
  float* vw[4] = {exp_avg_sq[0], exp_avg_sq[4], exp_avg_sq[5], exp_avg_sq[8]};
  void* cw[4] = {h->w1, h->w_ih, h->w_hh, h->w_c};            // w_c: rows >= n_classes stay zero (same linear index below them)
  const long long nw[4] = {numel[0], numel[4], numel[5], numel[8]};
  float* pb[6] = {params[1], params[2], params[3], params[6], params[7], params[9]};
  const float* gb[6] = {grads[1], grads[2], grads[3], grads[6], grads[7], grads[9]};
  float* mb[6] = {exp_avg[1], exp_avg[2], exp_avg[3], exp_avg[6], exp_avg[7], exp_avg[9]};
  float* vb[6] = {exp_avg_sq[1], exp_avg_sq[2], exp_avg_sq[3], exp_avg_sq[6], exp_avg_sq[7], exp_avg_sq[9]};
  void* cb[6] = {h->b1, h->ln_g, h->ln_b, nullptr, nullptr, h->b_c};
  const long long nb[6] = {numel[1], numel[2], numel[3], numel[6], numel[7], numel[9]};
  for (int i = 0; i < 10; ++i) if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]) return fail(PREGO_EINVAL, "adamw: tensor %d is NULL", i);
  // guarded by the handle's timeout word: after a forward / backward that gave up, the step changes nothing (the add / copy below then
  // rebuild the same derived vectors from the unchanged biases)
  if (launch_adamw(4, pw, gw, mw, vw, cw, nw, h->bf16, step, lr, beta1, beta2, eps, weight_decay, s, h->abort_word, h->peer_guard) ||
      launch_adamw(6, pb, gb, mb, vb, cb, nb, false, step, lr, beta1, beta2, eps, weight_decay, s, h->abort_word, h->peer_guard))
    return fail(PREGO_EINVAL, "adamw: bad step %lld", (long long)step);
  launch_add_vec(params[6], params[7], h->bias2, (int)(3 * H), (int)(2 * H), s);      // r,z rows: b_ih + b_hh ; n rows: b_ih
  h->perm_stale = true;
  HIPCHK(hipMemcpyAsync(h->b_hn, params[7] + 2 * H, (size_t)H * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

#ifdef PREGO_DEBUG_ABI
// probe: the ping-pong GEMM as a persistent worker that only runs on XCDs >= xcd_lo and claims tiles from `counter`
// (device word, zeroed by the caller in stream order); grid = workgroups launched (256 = one per CU)
extern "C" int prego_debug_gemm_worker(const void* A, const void* B, const float* bias, float* C, int M, int N, int K, int xcd_lo,
                                       unsigned* counter, int grid, prego_stream_t stream) {
  if (!A || !B || !bias || !C || !counter || M <= 0 || grid <= 0) return fail(PREGO_EINVAL, "debug gemm worker: bad arguments");
  if (launch_gemm_bf16_pingpong_worker(A, K, B, K, bias, C, N, M, N, K, xcd_lo, counter, grid, (hipStream_t)stream))
    return fail(PREGO_EINVAL, "debug gemm worker: unsupported shape");
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// debug / microbenchmark: C[M,N] (fp32) = A[M,K] . B[N,K]^T + bias with a chosen bf16 kernel variant
// (0 = 128x128 two-stage, 1 = 256x128 three-stage counted-vmcnt, 9 = 256x256 two-stage, 12 = ping-pong), scripts/gemm_bench.py
void launch_gemm_bf16_variant(int variant, const void* A, int lda, const void* B, int ldb, const float* bias, float* C, int ldc,
                              int M, int N, int K, hipStream_t s);
extern "C" int prego_debug_gemm_bf16(int variant, const void* A, const void* B, const float* bias, float* C, int M, int N, int K,
                                     prego_stream_t stream) {
  if (!A || !B || !C || M <= 0 || N % 128 || K % 64) return fail(PREGO_EINVAL, "debug gemm: bad arguments");
  launch_gemm_bf16_variant(variant, A, K, B, K, bias, C, N, M, N, K, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}
#endif  // PREGO_DEBUG_ABI
