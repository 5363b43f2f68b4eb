// Host-visible declarations of the kernel launchers (implemented in the .hip files).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>
#include <mutex>

// Tuning / A-B / diagnostic environment knobs (PREGO_SPLIT_LAG*, PREGO_PLAN_SLOTS, PREGO_GRU_STAMPS, PREGO_ATTN_NW, ...): only the
// DEBUG library (libprego_amd_debug.so, -DPREGO_DEBUG_ABI) reads them; in the product library this returns nullptr and every one of
// them stays at its default.  The product library reads exactly four switches between code paths that are both held by the driver-run
// tests: PREGO_SPLIT_PASS, PREGO_NO_XCD_OVERLAP, PREGO_GRU_NO_LOCAL, PREGO_GRU_NO_MT (include/prego_amd.h).  Defined in miniroad.cpp.
const char* prego_tune_env(const char* name);

// One-time per-DEVICE setup at a launch site (hipFuncSetAttribute, symbol addresses): function attributes and __device__
// symbols belong to the device that was current when they were set / resolved, so a process that drives several GPUs
// needs them once per device, and the first calls may race between host threads.
struct DeviceOnce {
  std::mutex mu;
  bool done[64] = {};
  template <class F> int run(F&& f) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) { f(); return 0; }
    std::lock_guard<std::mutex> g(mu);
    if (!done[dev]) { f(); done[dev] = true; }
    return dev;
  }
};

// host-visible copy of the slot schedule descriptor (layout identical to common.h's SlotPlan)
#ifndef PREGO_HAVE_SLOTPLAN
struct SlotPlan {
  const int* rowoff; const int* nact; const int* seg_off; const int* seg_clip; const int* seg_start; const int* blk_step;
  int s_max; int n_slots;
};
#define PREGO_HAVE_SLOTPLAN 1
#endif

// Start handshake of a split pass (DESIGN 5b "fail-safe"): both persistent launches must be resident TOGETHER, or neither may touch
// memory.  `word` (device, zeroed per pass) is decided ONCE by compare-and-swap: 1 = GO (the recurrence's leader has seen P workgroups on
// each of its XCDs and at least one feed-forward workgroup on each of the others), 2 = FAIL (somebody's bounded wait ran out: a
// dispatch-serialising profiler, another tenant on the XCDs).  Whoever decides also writes (seq << 2) | state to the pinned host word
// the calling thread polls: on FAIL both launches leave without having written anything and the host runs the chunked pass instead.
struct PassHandshake {
  unsigned* word;               // device: 0 undecided, 1 GO, 2 FAIL
  unsigned* ff_here;            // device [8]: feed-forward workgroups that have started, per XCD
  unsigned* host;               // pinned host memory, nullable
  unsigned seq;
  unsigned ticks_lead;          // s_memrealtime ticks (100 MHz) the recurrence's leader waits for the other launch
  unsigned ticks_all;           // ... everyone waits for a decision before voting FAIL
};
#define PREGO_HS_GO 1u
#define PREGO_HS_FAIL 2u

struct GruArgs {
  const void* whh;       // [3H][H] bf16 or f32, reference layout of gru.weight_hh_l0 (rows r|z|n)
  const float* b_hn;     // [H]  = gru.bias_hh_l0[2H:3H]
  const void* gi;        // [rows][3H] fp32 (or bf16 when gi_bf16), chunk-relative packed rows
  void* h_relu_out;      // [rows][H] WT  relu(h_t)  (operand of the classification GEMM), nullable
  float* h_raw_out;      // [rows][H] fp32 h_t (kept for BPTT), nullable
  float* h_state;        // [n_clips][H] fp32, indexed by SORTED clip position; in: h_{t0-1}, out: h_{t1-1}
  void* hx;              // exchange buffers [G][2][16*NCT][H] WT
  unsigned* flags;       // [G*P] one word per producer workgroup, zeroed before every launch
  unsigned* abort_word;  // set to 1 on spin timeout
  const int* rowoff;     // absolute packed row offsets, [t_max+1]
  const int* nact;       // [t_max]
  int t0, t1;            // time steps [t0, t1) handled by this launch
  int row_base;          // rowoff[t0]: chunk-relative row = rowoff[t] - row_base + sorted_index
  int n_clips;                 // number of SLOTS (one or more clips each)
  int G;
  const int* seg_off; const int* seg_start;   // slot schedule: where each slot's next clip starts (h := 0 there)
  float* keep_r; float* keep_z; float* keep_n; float* keep_ghn;   // [rows][H] gate activations for BPTT, nullable
  unsigned* sync;              // [16] placement rendezvous words (8 per-XCD tickets + total), zeroed per launch; nullable
  unsigned long long* stamps;  // debug: per-phase cycle sums of block 0 / wave 0 (nullable)
  int gi_bf16;                 // gi rows are bf16 (inference path with bf16 intermediates)
  int f16;                     // 16-bit operands / intermediates are IEEE fp16 instead of bf16 (launch_gru_recurrence picks the instantiation)
  int Gd;                      // groups the slots are dealt to (0 = G); one-tile kernel only
  int rows;                    // packed rows this launch covers (rowoff[t1] - row_base); 0 = unknown
  int armed;                   // 1: hx / sync were re-armed by an earlier kernel of this stream (launch_ln_relu with a GruArm): no arm launch
  int no_mt;                   // 1: never the software-pipelined multi-tile kernel (handle created under PREGO_GRU_NO_MT=1: A/B and the bit-identity test)
  // split pass (PASS instantiation, launch_gru_recurrence_pass): gi is a ring of (gi_row_mask + 1) rows indexed by absolute packed row,
  // filled by ff_pass.hip while this launch runs; h_relu_out is indexed by absolute packed row
  const unsigned* gi_cnt;      // [n_chunks] completed 256-row units of chunk c (chunk = rows >> chunk_shift)
  unsigned* rec_cnt;           // [n_chunks] += 1 per wave that has consumed the chunk
  int chunk_shift, n_chunks;
  int units_per_chunk, units_last;   // what gi_cnt[c] must reach (last chunk: units_last)
  unsigned gi_row_mask;
  PassHandshake hs;            // pass mode only
  float out_floor;             // h_relu_out = max(h_t, out_floor): 0 = relu(h_t) (the classifier's operand), -inf = h_t (next GRU layer's input)
};
// ---- split pass (ff_pass.hip + the PASS instantiation of gru_recurrence.hip): the feed-forward of a whole pass as one persistent
// kernel on XCDs xcd_lo .. 7 beside one persistent recurrence launch on XCDs 0 .. xcd_lo - 1
struct FfPassArgs {
  const float* const* rgb_ptrs; const float* const* flow_ptrs;   // per-clip feature arrays (device pointer tables), flow nullable
  SlotPlan plan;
  void* rowmap;                 // int2 [total_rows]: (clip, frame) of every packed row (written by the PACK jobs, read by the head); nullable
  int d_rgb, d_flow;            // d_flow = 0: no flow half
  int in16;                     // feature arrays hold the 16-bit operand type already
  int kx;                       // K of layer1 = d_rgb + d_flow
  const unsigned short* w1; int ld_w1; const float* b1;          // [E][ld_w1]
  const float* ln_g; const float* ln_b; float ln_eps;
  const unsigned short* w_ih; const float* bias2;                // [3H][E]
  int E, n3;                    // embedding width, 3 H
  unsigned short* X; unsigned short* Y; unsigned short* Eb;      // rings of ring_units x 256 rows
  unsigned short* GI;           // ring of gi_ring_units x 256 rows x n3
  int ring_units, gi_ring_units;
  int total_rows, n_units;      // n_units = ceil(total_rows / 256)
  int xcd_lo;                   // XCDs below belong to the recurrence
  int chunk_unit_shift;         // log2(units per chunk): chunk of unit u = u >> shift
  int rec_expect;               // recurrence waves that signal a chunk (groups x P x 4)
  int nt1, nt2;                 // tiles per unit of the two GEMMs (E / 256, 3H / 256)
  int sg;                       // units per super-round of an XCD's ticket order
  int max_wg;                   // debug library (PREGO_SPLIT_FF_CUS): workgroups per feed-forward XCD that take jobs (0 = all 32); the others leave after the handshake
  int lag1, lag2, lag3;         // super-rounds between PACK and L1 / LN / WIH of a unit
  int dbg;                      // timing experiments only (PREGO_SPLIT_DBG; wrong results): 1 skip the pack copies, 2 skip the LayerNorm rows, 4 / 8 skip the layer1 / W_ih tiles
  unsigned long long* stats;    // debug, nullable: [8] 10 ns tick sums (pack, l1, ln, wih, waits, ticket), [6] shader-clock cycles and [7] ticks of the workgroups' lifetime
  int f16;
  unsigned* tick;               // [8] per-XCD job tickets
  unsigned* pack_done; unsigned* l1_cnt; unsigned* ln_done; unsigned* wih_cnt;   // [n_units]
  unsigned* gi_cnt;             // [n_chunks] completed units per chunk (read by the recurrence)
  const unsigned* rec_cnt;      // [n_chunks] recurrence waves done with the chunk
  unsigned* abort_word;
  PassHandshake hs;
};
int launch_ff_pass(const FfPassArgs& a, hipStream_t s);
// the recurrence of a whole pass as one launch on XCDs 0 .. a.Gd - 1 (GruArgs pass fields); 16-bit operands, 16-bit GI ring, one tile
int launch_gru_recurrence_pass(int hid, GruArgs a, hipStream_t s);
int gru_group_size(bool bf16, int hid);            // workgroups per recurrence group
bool gru_hidden_supported(bool bf16, int hid);     // 1024; 512; 2048 with 16-bit operands
void launch_gru_arm(bool bf16, int hid, int G, void* hx, unsigned* sync, hipStream_t s);   // must precede it in the stream (see there)

// what a recurrence launch needs re-armed before it starts (gru_recurrence.hip: buffer 0 := tag 1 everywhere, buffer 1 := 0,
// sync[0..15] := 0).  A LayerNorm launch that runs between two recurrence launches of a stream can do it on the side (one launch
// and one launch gap fewer per chunk)
struct GruArm { unsigned* hx; unsigned long long words_per_buf; unsigned pattern; unsigned* sync; };
GruArm gru_arm_desc(bool bf16, int hid, int G, void* hx, unsigned* sync);

// persistent reverse-time recurrence of BPTT (gru_bptt.hip); all row indices are absolute packed rows of the kept forward
struct BpttArgs {
  const void* whhT;            // [H][3H] operand dtype: W_hh transposed (the contraction runs over the 3H gate rows)
  const float* dHout;          // [rows][H] dL/dh_t from the head (after the relu mask)
  const float* R; const float* Z; const float* N; const float* GHN;   // [rows][H] kept gate activations of the forward
  const float* Hraw;           // [rows][H] h_t of the forward
  float* dGI; float* dGH;      // [rows][3H] fp32 gradients of the two pre-activation terms
  void* dGIop; void* dGHop;    // the same in the operand dtype (wgrad GEMM operands)
  void* hx;                    // exchange buffers, gru_bptt_hx_bytes()
  unsigned* sync;              // [1024] placement words + one epoch word per producer workgroup (zeroed by the launcher)
  unsigned* abort_word;
  const int* rowoff; const int* nact;
  int t_max, n_clips, G;
  int force_sc1;               // test knob: skip the XCD-local fast path
};
size_t gru_bptt_hx_bytes(bool bf16, int hid, int G);
int launch_gru_bptt(bool bf16, int hid, int nct, BpttArgs a, hipStream_t s);

// GEMM epilogues (bf16 kernel): what happens to acc + bias
enum { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_GELU_BF16 = 2, EPI_STORE_BF16 = 3, EPI_QKV = 4, EPI_TOKENS = 5 };
struct GemmEpi {
  int mode;
  void* out_b;                 // bf16 output (EPI_GELU_BF16 / EPI_STORE_BF16), leading dim = ldc
  void* q; void* k;            // EPI_QKV destinations [B,h,n_tok,dh]
  int n_tok, heads, dh, emb;
  float q_scale;               // softmax scale folded into Q
  void* vn;                    // EPI_QKV: V [B,h,n_tok,dh] (row-major; the attention kernels read it transposed from LDS)
  int which0;                  // EPI_QKV: index of the first output block (0 = q|k|v; 1 = the GEMM computes k|v only)
  // training-mode nn.Dropout on the epilogue's result (EPI_RESIDUAL: on acc + bias before the residual add, Transformer.py:31,46;
  // EPI_GELU_BF16: on gelu(.), Transformer.py:41): stateless hash mask of (seed, m * ldc + n); thresh = 0 -> no dropout
  unsigned drop_thresh; float drop_scale; unsigned long long drop_seed;
  unsigned drop2_thresh; float drop2_scale; unsigned long long drop2_seed;   // a second, independent mask on the same value (proj_drop, Attention.py:19,40)
  float* pre_f32;              // EPI_GELU_BF16, training: the pre-activation W1 x + b1 in fp32 (gelu'), leading dim = ldc; nullable
  // EPI_TOKENS (ViT.py:125-129 in the encoding GEMM's epilogue): GEMM row m = b n_tok + t goes to row m + b of C (one cls row per
  // window is left for the caller) with the learned positional row pe[t] added: x[b, t] = W_enc f + b_enc + pe[t]
  const float* pe;
  int f16;                     // operands (and every 16-bit output) are IEEE fp16 instead of bf16
  int xcd_lo; unsigned* counter;   // WORKER instantiation (probe): XCDs below xcd_lo leave at once; tiles are claimed from *counter
  int split_a_lo, split_b_lo;      // SPLIT instantiation (fp16x2 operands): element column of the lo half in the A / B rows
  const float* acc_scale;          // SPLIT: device pointer to 1 / (power-of-two scale of the weights); NULL = 1
};
void launch_gemm_bf16_nt_epi(const void* A, int lda, const void* B, int ldb, const float* bias, float* C, int ldc, int M,
                             int N, int K, GemmEpi epi, hipStream_t s);

void launch_pack_rows(bool bf16, const float* const* rgb_ptrs, const float* const* flow_ptrs, const SlotPlan& plan,
                      int row0, int nrows, int d_rgb, int d_flow, void* X, hipStream_t s, int grid_limit = 0,
                      void* rowmap = nullptr /* int2 [nrows]: (clip, frame) of every packed row, for the head kernel */,
                      bool f16 = false /* with bf16 = true: the 16-bit operand type is IEEE fp16 */,
                      bool in16 = false /* the feature arrays hold the 16-bit operand type already (PREGO_FWD_IN16) */);
void launch_ln_relu(bool bf16, const void* Y, const float* gamma, const float* beta, int nrows, int E, float eps,
                    void* out, float* stats, float drop_p, unsigned long long seed, int row0_abs, hipStream_t s, int relu = 1,
                    bool in_bf16 = false, bool f16 = false, const GruArm* arm = nullptr);
void launch_f32_to_bf16(const float* src, void* dst, size_t n, hipStream_t s);
void launch_pad_convert(bool bf16, const float* src, int rows_src, int cols_src, int ld_src, void* dst, int rows_dst,
                        int cols_dst, hipStream_t s, bool f16 = false);
// 256x256x64 ping-pong (8-phase) kernel, csrc/gemm_pp.hip; -1 = shape not supported (N % 256, K % 64, K >= 128)
int launch_gemm_bf16_pingpong_mode(int mode, const void* A, int lda, const void* B, int ldb, const float* bias, void* C, int ldc,
                                   int M, int N, int K, bool out_bf16, hipStream_t s, bool f16 = false);
int launch_gemm_bf16_pingpong_epi(const void* A, int lda, const void* B, int ldb, const float* bias, void* C, int ldc, int M, int N,
                                  int K, GemmEpi epi, hipStream_t s);
int launch_gemm_bf16_pingpong_worker(const void* A, int lda, const void* B, int ldb, const float* bias, void* C, int ldc, int M, int N,
                                     int K, int xcd_lo, unsigned* counter, int grid, hipStream_t s, bool out16 = false, bool f16 = false);
int launch_gemm_x2_pingpong(const void* A, int lda, int a_lo, const void* B, int ldb, int b_lo, const float* inv_scale, const float* bias,
                            float* C, int ldc, int M, int N, int K, hipStream_t s);
void launch_x2_weight_split(const float* src, int rows, int cols, void* dst, float* scale2 /* device [2]: scale, 1 / scale */, hipStream_t s);
void launch_pack_rows_x2(const float* const* rgb_ptrs, const float* const* flow_ptrs, const SlotPlan& plan, int row0, int nrows,
                         int d_rgb, int d_flow, void* X, hipStream_t s, int grid_limit = 0, void* rowmap = nullptr);
void launch_ln_relu_x2(const float* Y, const float* gamma, const float* beta, int nrows, int E, float eps, void* out, hipStream_t s,
                       const GruArm* arm = nullptr);
// split-operand recurrence (gru_recurrence_x2.hip): a.whh = [3H][2H] split rows of W_hh * scale, a.gi fp32, a.h_relu_out fp32
int launch_gru_recurrence_x2(int hid, int nct, GruArgs a, const float* inv_scale, hipStream_t s);
size_t gru_x2_hx_bytes(int hid, int G);
GruArm gru_x2_arm_desc(int hid, int G, void* hx, unsigned* sync);
int launch_gemm_bf16_pingpong(const void* A, int lda, const void* B, int ldb, const float* bias, float* C, int ldc, int M, int N,
                              int K, hipStream_t s);
void launch_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, const float* bias, float* C, int ldc, int M,
                         int N, int K, hipStream_t s, bool f16 = false,
                         bool train_splitk = false /* bf16, at most 256 tiles of 128 x 128: the eight-wave split-K workgroup of gemm_tn.hip
                                                      (another summation order over k than every other NT kernel: training forward only) */);
void launch_gemm_f32_nt(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc, int M,
                        int N, int K, hipStream_t s);
int launch_gru_recurrence(bool bf16, int hid, int nct, GruArgs a, hipStream_t s);
size_t gru_hx_bytes(bool bf16, int hid, int G);
int gru_max_tiles();
int launch_head_softmax(bool bf16, const void* Hrelu, const void* Wc, const float* bc, const SlotPlan& plan,
                        int row0, int nrows, int hid, int C, int apply_softmax,
                        float* const* out_ptrs, int* const* argmax_ptrs, hipStream_t s, const void* rowmap = nullptr, bool f16 = false);
void launch_permute_rows(const float* src, float* dst, const int* sorted_clip, int n, int width, int to_sorted,
                         hipStream_t s);
void launch_add_vec(const float* a, const float* b, float* out, int n, int n_add, hipStream_t s);

// training path (train.hip)
void launch_oad_loss(const float* const* logit_ptrs, const float* const* target_ptrs, const int* lens, int n_clips, int C,
                     float* loss_out, float* const* dlogit_ptrs, float grad_scale, hipStream_t s, bool sum = false);
void launch_gather_dlogits(bool bf16, const float* const* dl_ptrs, const int* rowoff, const int* sorted_clip, int t_max,
                           int nrows, int C, int Cpad, void* out, hipStream_t s);
void launch_transpose_convert(bool in_bf16, bool out_bf16, const void* src, int M, int N, int ld_src, void* dst, int Mpad,
                              hipStream_t s);
void launch_colsum(const float* src, int M, int N, float* part, float* out, hipStream_t s);
void launch_colsum_bf16(const void* src, int M, int N, float* part, float* out, hipStream_t s);
void launch_colsum_stage2(const float* part, int nb, int N, float* out, hipStream_t s);
void launch_gru_bwd_step(bool bf16, int t, int na, int na_next, int row_t, int row_tm1, int H, const float* dHout,
                         const float* carry_in, const float* dhpart, const float* R, const float* Z, const float* Nn,
                         const float* GHN, const float* Hraw, float* carry_out, float* dGI, float* dGH, void* dGIop,
                         void* dGHop, hipStream_t s);
void launch_relu_mask(const float* dHrelu, const float* Hraw, size_t n, float* out, hipStream_t s);
void launch_build_hprev(bool bf16, const float* Hraw, const int* rowoff, int t_max, int nrows, int H, void* out,
                        hipStream_t s);
// relu = 0: plain LayerNorm backward (Transformer path); accumulate = 1: dY += result (residual stream)
int launch_ln_relu_bwd(const float* dE, const float* Y, const float* stats, const float* gamma, const float* beta, int nrows,
                       int E, float drop_p, unsigned long long seed, int row0_abs, float* dY, float* part, hipStream_t s,
                       int relu = 1, int accumulate = 0, void* dYb = nullptr /* bf16 copy of dY (wgrad operand), nullable */);
// training GEMMs on k-major operands (gemm_tn.hip): C[M,N] fp32 = op(A) . op(B) + bias; ta: A stored [K][M]; tb (required): B stored [K][N];
// colsum_out (ta only): [M] column sums of A = the bias gradient of a wgrad; k_valid: contraction rows present in memory
int launch_gemm_bf16_tn(bool ta, bool tb, const void* A, int lda, const void* B, int ldb, const float* bias, float* C, int ldc, int M, int N,
                        int K, int k_valid, float* colsum_out, hipStream_t s, void* C16 = nullptr /* bf16 C instead of fp32 */);

// post-processing (postproc.hip)
int launch_window_vote(const int* pred, long long n_frames, int window, int n_classes, int* votes, hipStream_t s);

// per-frame average precision (metrics.hip): segmented radix sort + scan, one segment per class
int launch_format_ids(const int* ids, long long n, unsigned* text, int* bad, hipStream_t s);
size_t perframe_ap_workspace_bytes(long long n_frames, int n_classes);
int launch_perframe_ap(const float* scores, const float* target, const int* labels, long long n_frames, int n_classes, double* ap, long long* n_pos,
                       double* score_sum, void* workspace, hipStream_t s);

// split pass (round 6): rows gate * H + u of a 16-bit [3H][E] matrix and an fp32 [3H] vector -> rows (u / 2) * 6 + 2 * gate + u % 2 (rowwise.hip)
void launch_permute_gi_rows(const void* w, const float* bias, void* w_perm, float* bias_perm, int H, int E, hipStream_t s);

// fused multi-tensor AdamW (optim.hip)
int launch_adamw(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 void* const* copies, const long long* numel, bool copy_bf16, long long step, float lr, float b1, float b2, float eps,
                 float wd, hipStream_t s, unsigned* guard = nullptr /* device word: non-zero = change nothing */,
                 const float* peer = nullptr /* device float: non-zero = change nothing and raise *guard (data-parallel runs) */);

// Transformer path (attention.hip, vit.hip)
// attention forward (attention.hip): query on the lane, V row-major [B,h,N,dh], Nq queries against N keys
int launch_flash_attention_v2(const void* Q, const void* K, const void* V, void* out, int B, int Nq, int N, int heads, int dh,
                              int causal, hipStream_t s, float* lse = nullptr, unsigned drop_thresh = 0, float drop_scale = 1.f,
                              unsigned long long drop_seed = 0, bool f16 = false /* Q, K, V, P and the output are IEEE fp16 */);
// attention backward (attention_bwd.hip): dqkv [B*N, 3*heads*dh] bf16; delta: scratch fp32 [B*heads*N]
int launch_attention_bwd(const void* Qs, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                         float* delta, void* dqkv, int B, int N, int heads, int dh, int causal, float q_scale, hipStream_t s,
                         unsigned drop_thresh = 0, float drop_scale = 1.f, unsigned long long drop_seed = 0);
// ViTEnc training glue (vit_train.hip)
// dropout arguments (thresh = p * 2^32, scale = 1 / (1 - p), seed): thresh = 0 means none
void launch_gelu_bwd(const float* df, const float* u, size_t n, float* du, void* du_bf16, hipStream_t s, unsigned drop_thresh = 0,
                     float drop_scale = 1.f, unsigned long long drop_seed = 0);
// out_f32 (nullable) / out_bf16 := src * dropout mask(seed, element index) * scale
void launch_mask_convert(const float* src, size_t n, float* out_f32, void* out_bf16, unsigned drop_thresh, float drop_scale,
                         unsigned long long drop_seed, hipStream_t s, unsigned drop2_thresh = 0, float drop2_scale = 1.f,
                         unsigned long long drop2_seed = 0);
void launch_vit_head_bwd(const float* x, const float* dlogits, int B, int N, int E, int C, const float* lnw, const float* lnb,
                         const float* hw, float* dx, float* scratch /*[3][B][E]*/, float* g_lnw, float* g_lnb, float* g_hw,
                         float* g_hb, hipStream_t s);
void launch_vit_tokens_bwd(const float* dx, int B, int T, int E, float* denc, float* g_pe, float* g_cls, hipStream_t s,
                           unsigned drop_thresh = 0, float drop_scale = 1.f, unsigned long long drop_seed = 0);
void launch_cat_convert(const float* rgb, const float* flow, int rows, int d_rgb, int d_flow, void* out_bf16, hipStream_t s, bool f16 = false);
void launch_vit_tokens(const float* enc, const float* cls, const float* pe, int B, int T, int E, float* x, hipStream_t s,
                       unsigned drop_thresh = 0, float drop_scale = 1.f, unsigned long long drop_seed = 0);
void launch_vit_cls_rows(const float* cls, const float* pe, int B, int T, int E, float* x, hipStream_t s);
void launch_vit_head(const float* x, int B, int N, int E, const float* lnw, const float* lnb, const float* hw,
                     const float* hb, int C, float* out, hipStream_t s, int* argmax = nullptr);
// sliding windows over one video: token rows of windows ending at frames t0 .. t0 + B - 1 from the per-frame encoding (vit.hip)
void launch_vit_sliding_tokens(const float* enc, const float* enc_b, const float* cls, const float* pe, int t0, int B, int T, int E,
                               float* x, const float* ln_w, const float* ln_b, void* xn, float* x0, hipStream_t s, bool f16 = false);
void launch_add_bias_rows(float* x, const float* bias, int rows, int n, hipStream_t s);
// fp32-operand mode of the Transformer path (vit_f32.hip): attention over fp32 rows (q / k / v at column offsets of one [B*N, ld]
// array, Nq <= N queries per sequence), exact-erf GELU in place, x += y, [rgb | flow] rows
int launch_attention_f32(const float* qkv, int ld, int q_off, int k_off, int v_off, float* out, int B, int N, int Nq, int heads,
                         int dh, int causal, float scale, hipStream_t s);
void launch_gelu_f32(float* u, size_t n, hipStream_t s);
void launch_add_rows(float* x, const float* y, size_t n, hipStream_t s);
void launch_cat_rows_f32(const float* rgb, const float* flow, int rows, int d_rgb, int d_flow, float* out, hipStream_t s);

// streaming step (stream_step.hip): skinny products for n <= 16 rows, one frame per stream
struct StreamGemv {
  const void* W;        // [Nout][K] bf16
  const void* X;        // input columns [0, kx1): [n][ldx], fp32 or bf16 (x_bf16)
  const void* X2;       // input columns [kx1, K): [n][ldx2]; nullptr = zeros
  const float* bias;    // nullable
  float* Y;             // [n][Nout]
  int Nout, K, kx1, ldx, ldx2, x_bf16;
  const float* ln_g = nullptr;   // non-null (with ln_b): X is the fp32 pre-LayerNorm row, normalised + ReLU inside the kernel (n <= 4, K % 2048 == 0)
  const float* ln_b = nullptr;
  float ln_eps = 1e-5f;
};
int launch_stream_gemv(int nprob, const StreamGemv* pr, int n, hipStream_t s, bool f16 = false);
int launch_stream_gates_head(const float* gi, const float* gh, const float* b_hn, float* h_state, const void* wc, const float* bc, int n,
                              int H, int C, int softmax, float* out, int* argmax, hipStream_t s, bool f16 = false);
