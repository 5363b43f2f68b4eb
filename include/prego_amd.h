/* prego_amd.h - C ABI of the MI355X-native PREGO step_recognition hot path.
 *
 * The reference (aleflabo/PREGO) is pure Python and has no FFI; its plug-in API for this path is the
 * string registry (step_recognition/utils/registry.py:6-20, model/model_builder.py:5-9).  The entry points
 * below are what a binding of that plug-in boundary needs; each one cites the reference interface it
 * replaces.  INTEGRATION.md shows the ctypes stub a PREGO maintainer would add.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only; no torch / HIP types in signatures
 *    (`prego_stream_t` is a `hipStream_t` passed as void*; NULL = the default stream).
 *  - every pointer called "device" is HBM memory of the current HIP device, fp32, row-major contiguous.
 *  - functions return 0 on success, a negative PREGO_E* code otherwise; prego_last_error() gives the text.
 *  - calls only enqueue work on the caller's stream; nothing waits for the END of device work except prego_miniroad_check().  One
 *    documented wait for a START: a prego_miniroad_forward() that runs the split pass returns once the stream has reached its two
 *    persistent launches and they have confirmed each other resident (normally at once; behind earlier work of the stream otherwise).
 *  - the caller owns inputs, outputs, the workspace and the resident buffer (prego_miniroad_set_resident); they must stay valid until
 *    the stream reaches the end of the call.  The handle owns converted weight copies, the plan tables (pre-sized at create for clips
 *    of up to 131 072 frames: forward() allocates nothing below that; a longer clip or more clips than any call before grows them
 *    once, behind a stream synchronisation) and a pinned staging buffer for the per-call pointer tables (host pointer arrays passed
 *    to a call may be freed as soon as the call returns).  Nothing else is allocated by a hot call: every per-call buffer whose size
 *    depends on the call is the caller's, sized by a query (workspace_bytes, resident_bytes, backward_workspace_bytes).
 *  - one handle per (device, stream); different handles are independent and re-entrant.
 */
#ifndef PREGO_AMD_H
#define PREGO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 1: round 1.  2: round 2 entry points (step, adamw, vit, attention layer, window vote) and the grown forward workspace.
 * 3: round 3 (fp16 operand mode, device AP, window_vote marks windows with an id outside [0, n_classes) as -1).
 * 4: round 4 (PREGO_F16X2 split-operand mode; the prego_debug_* / _debug_stamps entry points left this header and the product library:
 *    prego_amd_debug.h / libprego_amd_debug.so).
 * 5: round 4, later (prego_miniroad_pass_info; the split pass behind prego_miniroad_forward).
 * 6: round 5.  Behaviour: a forward() that runs the split pass returns once its two launches have met - see prego_miniroad_forward -,
 *    a pass that cannot run side by side is re-run chunked inside the same call instead of being reported as PREGO_ETIMEOUT by
 *    prego_miniroad_check; tuning environment knobs are read by the debug library only.  Added (existing signatures unchanged):
 *    prego_miniroad_create_layers / _set_gru_layer (num_layers 2), prego_oad_loss_reduce (reduction 'sum'),
 *    prego_attention_layer_set_dropout, prego_perframe_ap_labels, prego_onehot_labels (host), prego_format_ids.
 * 7: round 6.  prego_miniroad_forward no longer allocates or synchronises for the whole-call relu(h) buffer: the caller sizes it with
 *    prego_miniroad_resident_bytes and hands it over with prego_miniroad_set_resident (without one, every call runs the chunked pass
 *    with the per-chunk classifier - same results).  Added: prego_miniroad_resident_bytes / _set_resident, prego_miniroad_guard_publish /
 *    _set_peer_guard (data-parallel training: a timeout on one rank stops the optimizer step of every rank), prego_miniroad_set_gru_layer_grads
 *    (training of a two-layer GRU; PREGO_FWD_KEEP now takes hidden_dim 512 / 1024 / 2048 and num_layers 1 / 2).  Existing signatures unchanged. */
#define PREGO_ABI_VERSION 7

enum {
  PREGO_OK = 0,
  PREGO_EINVAL = -1,       /* bad argument / unsupported dimension */
  PREGO_EHIP = -2,         /* a HIP runtime call failed */
  PREGO_EWORKSPACE = -3,   /* workspace too small */
  PREGO_ETIMEOUT = -4      /* the persistent recurrence kernel gave up waiting (reported by _check) */
};

/* MFMA operand type of a handle.  Accumulators, GRU state, gate math, LayerNorm statistics and softmax are fp32 in every mode.
 * PREGO_F16: IEEE fp16 operands and 16-bit intermediates - the same matrix rate and bytes as bf16 with 8x less operand rounding
 * (values beyond +-65504 saturate); inference entry points only (forward without PREGO_FWD_KEEP, step). */
enum { PREGO_F32 = 0, PREGO_BF16 = 1, PREGO_F16 = 2, PREGO_F16X2 = 3 };
/* PREGO_F16X2 ("fp16x2", round 4): split operands - every matrix operand travels as two fp16 numbers (hi + lo, ~22 mantissa bits) and
 * every product is three fp16 MFMA products with fp32 accumulation; intermediates, state and the classifier stay fp32.  The
 * argmax-identical mode of the north star (rnn.py:58-70: fp32-class results) at 3/16 of the matrix cost of PREGO_F32.  MiniROAD
 * inference entry point only (prego_miniroad_forward without PREGO_FWD_KEEP / PREGO_FWD_IN16). */

/* forward() flags */
enum {
  PREGO_FWD_SOFTMAX = 1,   /* eval branch of MROAD.forward: out = softmax(logits) (rnn.py:66-70); else raw logits */
  PREGO_FWD_KEEP = 2,      /* keep activations for backward() in the training workspace */
  PREGO_FWD_IN16 = 4       /* rgb[i] / flow[i] hold the handle's 16-bit operand type (bf16 or IEEE fp16 bits) instead of fp32: a feeder that
                            * keeps 16-bit features in pinned host memory ships half the bytes per frame; bf16 / fp16 handles, inference only */
};

typedef void* prego_stream_t;
typedef struct prego_miniroad prego_miniroad;

int prego_abi_version(void);
const char* prego_last_error(void);     /* most recent error text of the calling thread (any handle, or handle-free calls) */

/* ---- MiniROAD (MROAD, registry name "MiniROAD"): step_recognition/model/rnn/rnn.py:18-71 ---------------- */

/* MROAD.__init__ (rnn.py:21-49): d_rgb/d_flow = FEATURE_SIZES of cfg['rgb_type'/'flow_type'] (0 when
 * --no_rgb/--no_flow), emb = cfg['embedding_dim'], hid = cfg['hidden_dim'], n_classes = cfg['num_classes'].
 * Supported on gfx950: hid in {512, 1024, 2048} (see prego_miniroad_create_layers), emb % 512 == 0 (<= 4096), d_rgb % 64 == 0,
 * d_flow % 64 == 0, n_classes <= 128.  One GRU layer. */
int prego_miniroad_create(prego_miniroad** out, int d_rgb, int d_flow, int emb, int hid, int n_classes,
                          int compute_dtype);
/* The same with cfg['num_layers'] (rnn.py:32,38: nn.GRU(embedding_dim, hidden_dim, num_layers)): 1 or 2.  Hidden sizes (rnn.py:31): 512,
 * 1024, 2048 with 16-bit operands; 512, 1024 with PREGO_F32; 1024 with PREGO_F16X2 (the recurrence keeps its slice of W_hh in registers:
 * what does not fit is refused here with a message).  Two layers: PREGO_F32 / PREGO_BF16 / PREGO_F16.  Inference (forward without
 * PREGO_FWD_KEEP) covers all of these, and since ABI 7 so does training (PREGO_FWD_KEEP + prego_miniroad_backward; PREGO_BF16 / PREGO_F32
 * handles, hidden_dim 2048: PREGO_BF16; two layers: prego_miniroad_set_gru_layer_grads); prego_miniroad_step and the split pass stay with
 * hidden_dim 1024 / one layer. */
int prego_miniroad_create_layers(prego_miniroad** out, int d_rgb, int d_flow, int emb, int hid, int n_classes, int num_layers,
                                 int compute_dtype);
void prego_miniroad_destroy(prego_miniroad* h);
/* text of the last error raised by an entry point of THIS handle (handles are independent: one per (device, stream)) */
const char* prego_miniroad_last_error(const prego_miniroad* h);

/* load_state_dict (main.py:48): device fp32 tensors with the reference's state_dict shapes
 *   layer1.0.weight [emb, d_rgb+d_flow]  layer1.0.bias [emb]   layer1.1.weight/bias [emb]   (rnn.py:39-44)
 *   gru.weight_ih_l0 [3*hid, emb]  gru.weight_hh_l0 [3*hid, hid]  gru.bias_ih_l0/bias_hh_l0 [3*hid] (rnn.py:38)
 *   f_classification.0.weight [n_classes, hid]  f_classification.0.bias [n_classes]          (rnn.py:45-47)
 * The handle keeps its own (converted) copies: call again after every optimizer step. */
int prego_miniroad_set_weights(prego_miniroad* h, const float* layer1_w, const float* layer1_b, const float* ln_w,
                               const float* ln_b, const float* w_ih, const float* w_hh, const float* b_ih,
                               const float* b_hh, const float* fc_w, const float* fc_b, prego_stream_t stream);

/* Layer 1 of a two-layer handle (state_dict keys gru.weight_ih_l1 [3*hid, hid], gru.weight_hh_l1 [3*hid, hid], gru.bias_ih_l1,
 * gru.bias_hh_l1 [3*hid]); layer 0 comes with prego_miniroad_set_weights.  h0 / h_last of such a handle are [2][n_clips][hid]. */
int prego_miniroad_set_gru_layer(prego_miniroad* h, int layer, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                                 prego_stream_t stream);

/* Largest number of clips one forward() call accepts (8192).  The call packs them, longest first, into its recurrence
 * slots (continuous batching: a slot runs several clips back to back, h restarts from 0 at every clip boundary), so
 * the number of sequential steps is max(longest clip, frames / slots).  Calls that pass h0 or h_last, or keep
 * activations for backward, need one clip per slot: at most 512 clips (bf16) / 256 (fp32). */
int prego_miniroad_max_clips(const prego_miniroad* h);

/* Workspace size (bytes) that lets forward() process `rows_per_chunk` packed rows (frames) per pipeline
 * pass; any size >= the value for rows_per_chunk = n_clips works, larger chunks run faster.
 * flags: PREGO_FWD_KEEP adds the activations backward() needs (whole batch resident). */
size_t prego_miniroad_workspace_bytes(const prego_miniroad* h, int n_clips, const int32_t* lens,
                                      int64_t rows_per_chunk, int flags);

/* Whole-call resident buffer (ABI 7).  Long inference calls run the classifier ONCE behind the pass instead of once per chunk, and the
 * split pass (two persistent launches, prego_miniroad_pass_info) keeps its row map and counters beside it: both need relu(h) of every
 * frame of the call resident - 2 KB per frame with 16-bit operands, 4 KB with fp32 / fp16x2 operands (4.7 GB for a 2.3 M-frame eval
 * set).  resident_bytes: the size that lets a call of these clips and flags use those passes (0 = such a call never would: training
 * calls, num_layers 2, fewer than 65 536 frames, more than 24 GB).  set_resident: a device buffer (256-byte aligned) the handle may use
 * for this in every following forward() until it is replaced (NULL, 0 removes it); it is caller-owned, read and written only between the
 * start and the end of a forward() call in stream order, and holds nothing between calls.  A call whose size exceeds the registered
 * buffer simply runs the chunked pass with the per-chunk classifier: same result bits, ~20 % slower on the eval-set workload.
 * forward() itself never allocates device memory for it and never waits for the stream because of it. */
size_t prego_miniroad_resident_bytes(const prego_miniroad* h, int n_clips, const int32_t* lens, int flags);
int prego_miniroad_set_resident(prego_miniroad* h, void* device_buffer, size_t bytes);

/* MROAD.forward (rnn.py:51-71) for a ragged batch, and the device half of Evaluate.eval (trainer/eval.py:36-56).
 *   lens[i]            frames of clip i (host array)
 *   rgb[i], flow[i]    device fp32 [lens[i], d_rgb] / [lens[i], d_flow]; flow == NULL or flow[i] == NULL means an
 *                      all-zero flow half (datasets/dataset.py:69) and skips its half of layer1's K dimension
 *   out[i]             device fp32 [lens[i], n_classes]: probabilities (PREGO_FWD_SOFTMAX) or logits; nullable
 *   argmax[i]          device int32 [lens[i]]: np.argmax(prob, axis=1) of eval.py:53, first max wins; nullable
 *   h0 / h_last        device fp32 [n_clips, hid] GRU state before frame 0 / after the last frame of each clip;
 *                      NULL h0 = zeros (rnn.py:49,60).  Chaining h_last -> h0 gives streaming inference.
 * The pointer arrays themselves are host arrays (copied during the call).
 * Which pass runs (prego_miniroad_pass_info reports it) is the library's choice per call and never changes a result bit.  The split pass
 * needs its two persistent launches resident together: they start with a bounded handshake, the call returns once it has succeeded, and
 * if it fails (a profiler that serialises kernel dispatches, another tenant holding the XCDs) both launches leave before either has
 * written anything and THIS call runs the chunked pass instead - no call is ever lost to the choice of pass. */
int prego_miniroad_forward(prego_miniroad* h, int n_clips, const int32_t* lens, const float* const* rgb,
                           const float* const* flow, float* const* out, int32_t* const* argmax, const float* h0,
                           float* h_last, int flags, void* workspace, size_t workspace_bytes,
                           prego_stream_t stream);

/* Streaming inference, the online use of the model: ONE new frame for each of n_streams <= 16 independent streams -
 * MROAD.forward (rnn.py:51-71) with T = 1 and h0 = the state the previous call left.
 *   rgb / flow     device fp32 [n_streams, d_rgb] / [n_streams, d_flow], one frame per stream; flow == NULL = zero flow half
 *                  (rgb is ignored by a --no_rgb model)
 *   h_state        device fp32 [n_streams, hid], read and OVERWRITTEN with the new state (zeros before a stream's first frame)
 *   out            device fp32 [n_streams, n_classes]: probabilities (PREGO_FWD_SOFTMAX in flags) or logits; nullable
 *   argmax         device int32 [n_streams]; nullable
 * Three kernel launches for <= 4 streams (LayerNorm inside the W_ih product), four above; no plan, no workspace, no host staging (the general forward() with n_clips = 1, lens = {1}, h0, h_last is
 * the same arithmetic in eight launches plus table staging).  bf16 handles only (PREGO_EINVAL otherwise: use forward()).
 * Projection outputs stay fp32 here (forward()'s inference path rounds them to bf16), so the two paths agree to the operand
 * rounding, not bit for bit. */
int prego_miniroad_step(prego_miniroad* h, int n_streams, const float* rgb, const float* flow, float* h_state, float* out,
                        int32_t* argmax, int flags, prego_stream_t stream);

/* Synchronises `stream` and reports a recurrence timeout (PREGO_ETIMEOUT) or HIP error since the last check. */
int prego_miniroad_check(prego_miniroad* h, prego_stream_t stream);

/* Link-fed inference (the eval loop of trainer/eval.py:36-56 with features in host memory): let the H2D copy of a batch run UNDER its
 * forward instead of in front of it.
 *   plan_starts       the slot schedule the next forward of exactly these clips will use, costed for features that arrive at link speed
 *                     (link_row_bytes = bytes one frame moves over the link; 0 = features already in HBM): start_step[i] = the step at
 *                     which frame 0 of clip i is consumed, so frame a of clip i is needed at step start_step[i] + a; *n_steps = steps.
 *   set_feed_events   for the NEXT forward only: rows needed at steps < upto_step[j] are valid in the rgb / flow arrays once
 *                     events[0..j] (hipEvent_t recorded by the caller behind its copies, upto_step ascending) have fired; the last
 *                     upto_step must be >= *n_steps.  The library makes its packing stream wait for the events a chunk needs - the
 *                     caller copies in need order and never waits on the host.  Plain inference calls only (no h0 / h_last / KEEP). */
int prego_miniroad_plan_starts(prego_miniroad* h, int n_clips, const int32_t* lens, int link_row_bytes, int32_t* start_step, int32_t* n_steps);
int prego_miniroad_set_feed_events(prego_miniroad* h, int n_events, const int32_t* upto_step, void* const* events, int link_row_bytes);

/* Kernel-level timing hooks for bench.py's roofline leg: when enabled, forward() brackets its GEMM launches and
 * its recurrence launches with HIP events on the caller's stream; read() synchronises and returns the summed
 * milliseconds and launch counts since enable. */
int prego_miniroad_timing_enable(prego_miniroad* h, int enable);
/* (a split pass reports its recurrence launch in the gru slot and its feed-forward launch - pack, both projections, LayerNorm - in the
 * pack slot; gemm_flop still counts the projections' flops, gemm_ms / gemm_launches stay 0) */
int prego_miniroad_timing_read(prego_miniroad* h, double* gemm_ms, int64_t* gemm_launches, double* gemm_flop,
                               double* gru_ms, int64_t* gru_launches, double* pack_ms, int64_t* pack_launches,
                               double* pack_bytes);
/* What the last prego_miniroad_forward() of this handle ran (any pointer may be NULL): *mode = 0: the chunked pass (a chain of launches
 * per chunk); R > 0: the split pass - the recurrence of the whole call as ONE launch on R XCDs (16 R slots) beside ONE feed-forward
 * launch on the other XCDs (plain inference calls of 16-bit handles with >= 16 R clips and >= 262 144 frames, once an earlier
 * call has verified the workgroup placement; PREGO_SPLIT_PASS=R selects it).  *n_steps sequential recurrence steps, *n_slots slots. */
int prego_miniroad_pass_info(const prego_miniroad* h, int32_t* mode, int32_t* n_steps, int32_t* n_slots);

/* ---- training: trainer/train.py:6-29 (fwd, loss, backward); criterions/loss.py:15-34 -------------------------- */

/* nn.Dropout(p=cfg['dropout']) after layer1's ReLU (rnn.py:43): applied by forward() calls that carry
 * PREGO_FWD_KEEP (training mode); the mask is a stateless hash of (seed, element index), regenerated in backward. */
int prego_miniroad_set_dropout(prego_miniroad* h, float p, uint64_t seed);

/* OadLoss ("NONUNIFORM", loss.py:15-34): loss = mean_b sum_k -(y/||y||_2)_k * log_softmax(logits[b,-1,:])_k with
 * F.normalize's 1e-12 clamp.  logits[i], target[i]: device fp32 [lens[i], n_classes]; loss_out: device fp32 scalar;
 * dlogits[i] (array nullable): device fp32 [lens[i], n_classes] := grad_scale * dloss/dlogits (zero except last frame). */
int prego_oad_loss(int n_clips, const int32_t* lens, const float* const* logits, const float* const* target,
                   int n_classes, float* loss_out, float* const* dlogits, float grad_scale, prego_stream_t stream);
/* OadLoss(cfg, reduction) (loss.py:8-11,30-33): reduction 0 = 'mean' (what main.py builds; prego_oad_loss), 1 = 'sum' over the batch. */
int prego_oad_loss_reduce(int n_clips, const int32_t* lens, const float* const* logits, const float* const* target,
                          int n_classes, int reduction, float* loss_out, float* const* dlogits, float grad_scale, prego_stream_t stream);

/* loss.backward() through MROAD (train.py:23).  Must follow a forward() with PREGO_FWD_KEEP of the same clips whose
 * workspace is passed back as fwd_workspace (untouched in between).  dlogits[i]: device fp32 [lens[i], n_classes]
 * (any values: all frames are honoured).  The ten gradient tensors have the shapes of set_weights' arguments and are
 * OVERWRITTEN.  All column sums are fixed-order (deterministic). */
/* Data-parallel training (trainer/train.py:20-24 under clip sharding): hipEvent_t handles (NULL = none) that every following
 * prego_miniroad_backward records on its stream at two milestones - f_classification gradients final; all four GRU gradients
 * final - so the caller can all-reduce those buckets on another stream under the rest of the backward (layer1's weight gradient,
 * 47 % of the bytes, is final only when backward returns).  The events stay owned by the caller. */
int prego_miniroad_backward_events(prego_miniroad* h, void* ev_head_done, void* ev_gru_done);
/* Optional host callback of backward(): fn(user, bucket) is called on the calling thread, from inside prego_miniroad_backward, right after
 * the launches that make a group of gradient tensors final have been ENQUEUED and its event (above) recorded - bucket 0: f_classification,
 * bucket 1: the four GRU tensors.  A data-parallel caller enqueues that bucket's all-reduce there (behind the event, on its own
 * stream), i.e. ahead of the ~30 launches of the rest of the backward instead of behind them.  The callback must not call back into
 * this handle.  fn = NULL removes it. */
typedef void (*prego_bucket_fn)(void* user, int bucket);
int prego_miniroad_backward_callback(prego_miniroad* h, prego_bucket_fn fn, void* user);
/* A stacked GRU (num_layers 2, rnn.py:32,38) under loss.backward() (ABI 7): where the gradients of gru.weight_ih_l1 [3H, H],
 * gru.weight_hh_l1 [3H, H], gru.bias_ih_l1 [3H], gru.bias_hh_l1 [3H] go (device fp32, OVERWRITTEN by every following
 * prego_miniroad_backward; layer 0's are that call's own arguments).  Required before the backward of a 2-layer handle. */
int prego_miniroad_set_gru_layer_grads(prego_miniroad* h, int layer, float* g_w_ih, float* g_w_hh, float* g_b_ih, float* g_b_hh);
size_t prego_miniroad_backward_workspace_bytes(const prego_miniroad* h, int n_clips, const int32_t* lens);
int prego_miniroad_backward(prego_miniroad* h, int n_clips, const int32_t* lens, const float* const* dlogits,
                            float* g_layer1_w, float* g_layer1_b, float* g_ln_w, float* g_ln_b, float* g_w_ih,
                            float* g_w_hh, float* g_b_ih, float* g_b_hh, float* g_fc_w, float* g_fc_b,
                            void* fwd_workspace, size_t fwd_workspace_bytes, void* bwd_workspace,
                            size_t bwd_workspace_bytes, prego_stream_t stream);

/* utils/aggregate.py:55-72 on the device: the per-frame argmax of ONE video (device int32 [n_frames], as prego_miniroad_forward
 * writes it) is cut into consecutive windows of `window` frames (the reference uses 200; the last one may be shorter) and every
 * window votes for its most frequent class, the lowest class id winning a tie (np.argmax(np.bincount(.))).
 * votes: device int32 [ceil(n_frames / window)].  n_classes <= 128; a window that holds an id outside [0, n_classes) votes -1
 * (np.bincount raises on a negative id; the host wrapper turns the marker into an error).  The de-duplication / change lists of aggregate.py:75-78
 * then run over one value per window instead of one per frame. */
int prego_window_vote(const int32_t* argmax, int64_t n_frames, int window, int n_classes, int32_t* votes, prego_stream_t stream);

/* trainer/eval.py:59-65 writes the per-frame predicted and ground-truth class ids of every video as JSON text.  ids: device int32 [n]
 * (0 <= id <= 999); text: device uint32 [n], element i = the four bytes "%3d," of ids[i] in memory order (blanks in front of a number
 * are JSON whitespace), so the host cuts the text per video and turns every list's last comma into its bracket.  bad (nullable):
 * device int32, set to 1 when an id is out of range (its text is then "  0,": the caller must not use the text). */
int prego_format_ids(const int32_t* ids, int64_t n, uint32_t* text, int32_t* bad, prego_stream_t stream);

/* utils/metrics.py:25-62 on the device: sklearn.metrics.average_precision_score of every class column of the per-frame score
 * matrix the eval loop collects (trainer/eval.py:48-57; main.py:101 runs it after every epoch).  scores / target: device fp32
 * [n_frames][n_classes] row-major (target != 0 marks a positive).  Thresholds are the distinct score values (ties share one),
 * AP = sum_k (R_k - R_{k-1}) P_k.  ap: device double [n_classes] (NaN for a class without positives); n_pos (nullable): device
 * int64 [n_classes] positives per class; score_sum (nullable): device double [n_classes] column sums of the scores (the "pred:"
 * figure of metrics.py:52).  The caller applies the reference's "ignore class 0" rule (metrics.py:44-48) when averaging.
 * Exact integer ranks, fp64 sum: only thresholds that hold a positive contribute, so the positives of every class are sorted
 * (segmented radix sort) and every score is counted against them (csrc/metrics.hip).  Workspace:
 * prego_perframe_ap_workspace_bytes (16 B per score + histograms). */
size_t prego_perframe_ap_workspace_bytes(int64_t n_frames, int n_classes);
int prego_perframe_ap(const float* scores, const float* target, int64_t n_frames, int n_classes, double* ap, int64_t* n_pos,
                      double* score_sum, void* workspace, size_t workspace_bytes, prego_stream_t stream);
/* The same metric with the positives given as ONE class id per frame (device int32 [n_frames]; an id outside [0, n_classes) = a
 * frame without a positive): what a one-hot target matrix says, in 4 bytes per frame instead of 4 x n_classes.  Same results, bit for bit. */
int prego_perframe_ap_labels(const float* scores, const int32_t* labels, int64_t n_frames, int n_classes, double* ap, int64_t* n_pos,
                             double* score_sum, void* workspace, size_t workspace_bytes, prego_stream_t stream);
/* HOST function (no device work): the reference's loader yields fp32 target rows [n_frames][n_classes] per video (one-hot for both
 * shipped configs).  targets: n_videos host pointers, n_frames: their row counts; labels: host int32 [sum of n_frames], the videos one
 * after the other, labels[i] = np.argmax(row i) (eval.py:55, the first maximum); onehot[v] = 1 iff every row of video v holds exactly one
 * nonzero entry and it is positive - then the labels say everything the matrix does and prego_perframe_ap_labels replaces
 * prego_perframe_ap (the eval loop sends 4 bytes per frame over the link instead of 4 x n_classes).  Up to 16 threads. */
int prego_onehot_labels(int n_videos, const float* const* targets, const int64_t* n_frames, int n_classes, int32_t* labels,
                        int32_t* onehot);

/* torch.optim.AdamW as main.py:62-67 builds it (amsgrad off, maximize off), fused over a tensor list in one launch:
 *   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)
 * params/grads/exp_avg/exp_avg_sq: host arrays of n_tensors device fp32 pointers; numel: host array; step is 1-based. */
int prego_adamw_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                     float* const* exp_avg_sq, const int64_t* numel, int64_t step, float lr, float beta1, float beta2, float eps,
                     float weight_decay, prego_stream_t stream);
/* The same step for MiniROAD's ten tensors (prego_miniroad_set_weights' order, reference state_dict shapes) that ALSO rewrites the
 * handle's converted operand copies from the updated values in the same pass: the training loop (train.py:24 optimizer.step())
 * needs no prego_miniroad_set_weights after it.  The launch is GUARDED by the handle's timeout word on the device: while a recurrence /
 * BPTT timeout of this handle is pending (set by the kernel that gave up, cleared by prego_miniroad_check, which reports it), the step
 * changes nothing - a training loop may enqueue it without synchronising first and still never applies garbage gradients. */
int prego_miniroad_adamw_step(prego_miniroad* h, float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, int64_t step, float lr, float beta1, float beta2, float eps,
                              float weight_decay, prego_stream_t stream);
/* The same guard across data-parallel ranks (ABI 7; train.py:20-24 under clip sharding).  A rank whose kernels gave up still takes part
 * in the gradient all-reduce, so its garbage reaches every rank: the guard has to be collective.
 *   guard_publish: enqueue dst[0] = (this handle's timeout word is set) ? 1.0f : 0.0f.  dst is one fp32 element of the gradient bucket
 *                  the ranks sum (the caller reserves it; call it after prego_miniroad_backward, before that bucket's all-reduce).
 *   set_peer_guard: the address of that element (after the reduction it is non-zero iff ANY rank gave up), NULL = none.  While it
 *                  holds a non-zero value prego_miniroad_adamw_step changes nothing AND raises this handle's own timeout word, so the
 *                  following steps are skipped as well and prego_miniroad_check reports PREGO_ETIMEOUT on every rank at the same step. */
int prego_miniroad_guard_publish(prego_miniroad* h, float* dst, prego_stream_t stream);
int prego_miniroad_set_peer_guard(prego_miniroad* h, const float* reduced_word);

/* ---- "Transformer" (ViTEnc): step_recognition/model/transformer_models/ViT.py:25-143 ------------------------------ */
typedef struct prego_vit prego_vit;

/* ViTEnc.__init__ (ViT.py:26-90) with patch_dim = 1: emb = cfg['embedding_dim'], mlp = cfg['hidden_dim'] (ViT.py:75),
 * heads = cfg['num_heads'], layers = cfg['num_layers'], window = cfg['window_size'] (the learned positional table pins
 * T == window, PositionalEncoding.py:25-41).  Supported: head dim 64/128/256, emb % 512 == 0, mlp % 128 == 0. */
int prego_vit_create(prego_vit** out, int d_rgb, int d_flow, int emb, int mlp, int heads, int layers, int window,
                     int n_classes);
void prego_vit_destroy(prego_vit* h);
/* number of tensors set_weights expects: 4 + 11 * layers + 4 */
int prego_vit_num_tensors(const prego_vit* h);
/* device fp32 tensors in state_dict order:
 *   linear_encoding.weight [emb, d_in], linear_encoding.bias, cls_token [emb], position_encoding.pe.weight [window+1, emb],
 *   per layer l (a = 2l, f = 2l+1): encoder.net.a.fn.norm.{weight,bias}, encoder.net.a.fn.fn.qkv.weight [3emb, emb],
 *     encoder.net.a.fn.fn.proj.{weight [emb,emb], bias}, encoder.net.f.fn.norm.{weight,bias},
 *     encoder.net.f.fn.fn.net.0.{weight [mlp,emb], bias}, encoder.net.f.fn.fn.net.3.{weight [emb,mlp], bias},
 *   pre_head_ln.{weight,bias}, mlp_head.{weight [n_classes, emb], bias} */
int prego_vit_set_weights(prego_vit* h, const float* const* tensors, int n_tensors, prego_stream_t stream);
size_t prego_vit_workspace_bytes(const prego_vit* h, int batch);
/* MFMA operand type of the handle: PREGO_BF16 (default) or PREGO_F16 (IEEE fp16 operands and 16-bit activations: the same rate
 * and bytes, 8x less operand rounding - logits within 1e-3 of the fp32 reference where bf16 gives 2.5e-3; inference entry points
 * only: forward, forward_frames), or PREGO_F32 (parity mode: the matrices stay fp32, the projections run on the exact-fp32 MFMA
 * GEMM, attention / GELU / residuals in fp32, every block on every token; prego_vit_forward only - logits within 2e-5 of the
 * reference; the north star's "1e-3 fp32" figure is checked on it).  Changing the type invalidates the ingested weights (call
 * set_weights again); prego_vit_workspace_bytes follows the handle's type. */
int prego_vit_set_compute_dtype(prego_vit* h, int compute_dtype);

/* ViTEnc.forward (ViT.py:117-143): rgb/flow device fp32 [batch, window, d_rgb/d_flow] (flow NULL = zeros);
 * out_logits device fp32 [batch, n_classes] (the reference returns it as [batch, 1, n_classes], raw logits in both
 * modes).  flags bit 0: causal self-attention (extension; the reference module has no mask, Attention.py:21-41).
 * The reference reads the encoder output at token 0 only (ViT.py:136), so the LAST block computes its query, attention output,
 * projection and FFN for token 0 of every window only (keys / values for all tokens): exact, 44 % fewer FLOPs at num_layers = 1.
 * flags bit 1 (debug / A-B): run the last block on every token as the reference does. */
int prego_vit_forward(prego_vit* h, int batch, const float* rgb, const float* flow, float* out_logits, int flags,
                      void* workspace, size_t workspace_bytes, prego_stream_t stream);

/* Per-frame inference of the `Transformer` entry over ONE whole video - what trainer/eval.py:36-56 needs from a model whose forward
 * emits one logit vector per window (ViT.py:136-141) and requires T == window_size: out_logits[t] = ViTEnc(window ending at frame
 * t), zero feature rows in front of the video (the training loader's windows, datasets/dataset.py:53-55,96-103, at stride 1).
 * linear_encoding (ViT.py:124) runs once per frame, not once per (window, position); windows go through the encoder
 * `windows_per_batch` at a time.  rgb / flow: device fp32 [n_frames][d_rgb | d_flow] (flow NULL = zeros); out_logits: device fp32
 * [n_frames][n_classes] raw logits (ViTEnc applies no softmax); out_argmax (nullable): device int32 [n_frames], np.argmax of each
 * row (trainer/eval.py:53).  flags bit 0: causal attention.  Workspace: prego_vit_frames_workspace_bytes. */
size_t prego_vit_frames_workspace_bytes(const prego_vit* h, int n_frames, int windows_per_batch);
int prego_vit_forward_frames(prego_vit* h, int n_frames, const float* rgb, const float* flow, float* out_logits, int32_t* out_argmax,
                             int windows_per_batch, int flags, void* workspace, size_t workspace_bytes, prego_stream_t stream);

/* Training of the "Transformer" registry entry: trainer/train.py:20-24 (fwd, loss, backward) over ViTEnc (ViT.py:117-143,
 * Transformer.py:5-82, Attention.py:21-41).  forward_train is ViTEnc.forward in training mode with every dropout rate 0
 * (cfg['dropout'] == cfg['attn_dropout_rate'] == 0; non-zero rates are rejected by the host module) and keeps the activations
 * the backward needs in `workspace` (caller-owned, prego_vit_train_workspace_bytes, untouched between the two calls).
 * backward: dlogits device fp32 [batch, n_classes] = dLoss/dlogits (OadLoss: prego_oad_loss on the [batch,1,C] logits);
 * grads: host array of n_tensors device fp32 tensors in prego_vit_set_weights' order and shapes, OVERWRITTEN.  All sums over
 * rows / windows run in a fixed order (bit-reproducible); the attention backward recomputes the probabilities from Q, K and the
 * forward's log-sum-exp (no [B,h,N,N] tensor) and uses no atomics.  flags bit 0: causal attention (as in forward). */
/* The nn.Dropout layers of the training-mode forward.  p = cfg['dropout']: pe_dropout (ViT.py:130), PreNormDrop after the
 * attention block (Transformer.py:24-32) and the two Dropouts of FeedForward (Transformer.py:41,46).  attn_p =
 * cfg['attn_dropout_rate']: the attention probabilities inside the attention kernel (Attention.py:17,36) and proj_drop
 * (Attention.py:19,40).  All are stateless hash masks of (seed, site, element), regenerated by backward. */
int prego_vit_set_dropout(prego_vit* h, float p, float attn_p, uint64_t seed);
size_t prego_vit_train_workspace_bytes(const prego_vit* h, int batch);
int prego_vit_forward_train(prego_vit* h, int batch, const float* rgb, const float* flow, float* out_logits, int flags,
                            void* workspace, size_t workspace_bytes, prego_stream_t stream);
int prego_vit_backward(prego_vit* h, int batch, const float* dlogits, float* const* grads, int n_tensors, int flags,
                       void* workspace, size_t workspace_bytes, prego_stream_t stream);
/* optimizer.step() (train.py:24, AdamW of main.py:62-67) for the handle's tensors (prego_vit_set_weights' order and shapes):
 * prego_adamw_step's arithmetic, and the handle's converted copies (bf16 matrices, fp32 vectors) are rewritten from the updated
 * values in the same pass - no prego_vit_set_weights (5 + 4 per layer conversions and 10 + 7 per layer copies) after the step. */
int prego_vit_adamw_step(prego_vit* h, float* const* params, const float* const* grads, float* const* exp_avg,
                         float* const* exp_avg_sq, int n_tensors, int64_t step, float lr, float beta1, float beta2, float eps,
                         float weight_decay, prego_stream_t stream);

/* AttentionLayer(FullAttention(mask_flag=causal)) of attn.py:139-170,35-57,10-18 as a stateless op (BASELINE config 4:
 * long-window causal attention).  x, out: device fp32 [batch, len, d_model]; projection weights [d_model, d_model] and
 * biases [d_model] in nn.Linear layout.  scores = softmax(mask(q k^T) / sqrt(d_model/heads)); never materialised. */
size_t prego_attention_layer_workspace_bytes(int batch, int len, int d_model);
int prego_attention_layer_forward(int batch, int len, int d_model, int heads, int causal, const float* x, const float* wq,
                                  const float* bq, const float* wk, const float* bk, const float* wv, const float* bv,
                                  const float* wo, const float* bo, float* out, void* workspace, size_t workspace_bytes,
                                  prego_stream_t stream);

/* The same layer as a handle: the projection weights are converted once by set_weights (nn.Linear layout, as above), forward
 * then only moves activations.  x, out: device fp32 [batch, len, d_model]. */
typedef struct prego_attn_layer prego_attn_layer;
int prego_attention_layer_create(prego_attn_layer** out, int d_model, int heads);
void prego_attention_layer_destroy(prego_attn_layer* h);
/* PREGO_BF16 (default), PREGO_F16 or PREGO_F32 (parity mode: fp32 projections and fp32 attention, handle_forward only) operands
 * for the handle's forward; changing the type invalidates the ingested weights; handle_workspace_bytes follows the type. */
int prego_attention_layer_set_compute_dtype(prego_attn_layer* h, int compute_dtype);
int prego_attention_layer_set_weights(prego_attn_layer* h, const float* wq, const float* bq, const float* wk, const float* bk,
                                      const float* wv, const float* bv, const float* wo, const float* bo, prego_stream_t stream);
size_t prego_attention_layer_handle_workspace_bytes(const prego_attn_layer* h, int batch, int len);
int prego_attention_layer_handle_forward(prego_attn_layer* h, int batch, int len, int causal, const float* x, float* out,
                                         void* workspace, size_t workspace_bytes, prego_stream_t stream);
/* The layer under autograd (attn.py:151-170 inside loss.backward(), trainer/train.py:20-24): forward_train is handle_forward that
 * keeps x, q, k, v, the attention output (bf16) and the row log-sum-exp in `workspace`; backward reads them from the SAME
 * workspace and overwrites grads[0..7] (fp32, set_weights order: wq, bq, wk, bk, wv, bv, wo, bo) and, if not NULL,
 * dx [batch, len, d_model] (queries = keys = values = x: the three input gradients summed).  bf16 handles only; the
 * attention_dropout of attn.py:39,54 (nn.Dropout on A = softmax(scale * scores), active under module.train()): prego_attention_layer_set_dropout
 * below, p = 0 by default (the state the reference's modules are in under .eval()). */
/* FullAttention(attention_dropout = p) in training mode: every forward_train / backward pair that follows drops each attention probability
 * with probability p and scales the survivors by 1 / (1 - p) (the softmax denominator sums the undropped row), by a stateless hash of
 * (seed, element) - pass a fresh seed per step; the backward regenerates the mask of its forward.  0 <= p < 1. */
int prego_attention_layer_set_dropout(prego_attn_layer* h, float p, uint64_t seed);
size_t prego_attention_layer_train_workspace_bytes(const prego_attn_layer* h, int batch, int len);
int prego_attention_layer_forward_train(prego_attn_layer* h, int batch, int len, int causal, const float* x, float* out,
                                        void* workspace, size_t workspace_bytes, prego_stream_t stream);
int prego_attention_layer_backward(prego_attn_layer* h, int batch, int len, int causal, const float* dout, float* dx,
                                   float* const* grads, int n_tensors, void* workspace, size_t workspace_bytes,
                                   prego_stream_t stream);

/* Environment switches THIS library reads (each once, when a handle is created; none is needed in production).  Four, each between two
 * code paths that the driver-run GPU tests hold bit-identical to each other:
 *   PREGO_SPLIT_PASS      0 = never the split pass, R = on R XCDs whenever a call is eligible; unset = per call (cost model)   tests/test_gpu_split.py
 *   PREGO_NO_XCD_OVERLAP  no layer1 worker on the XCDs a thinned-out recurrence has left (the serial chunked pass)            tests/test_gpu_fullsize.py
 *   PREGO_GRU_NO_LOCAL    never the XCD-local hand-off of the recurrence (sc1 stores / loads everywhere)                      tests/test_gpu_miniroad.py
 *   PREGO_GRU_NO_MT       multi-tile steps on the classic recurrence kernel                                                   tests/test_gpu_fullsize.py
 * Every tuning / calibration / diagnostic knob (PREGO_SPLIT_LAG1..3, PREGO_SPLIT_CHUNK_SHIFT, PREGO_SPLIT_GI_RING, PREGO_SPLIT_STATS, PREGO_PLAN_SLOTS,
 * PREGO_SIDE_PRIO, PREGO_PACK_*, PREGO_GRU_STAMPS, PREGO_GRU_MT_SPEC, PREGO_GEMM_NO_*, PREGO_HEAD_V1, PREGO_ATTN_NW, ...) is read by
 * libprego_amd_debug.so ONLY (csrc/kernels.h: prego_tune_env returns NULL in this library), like the probe / unit-test entry points
 * (prego_debug_*, prego_miniroad_debug_stamps: include/prego_amd_debug.h; the same sources built with -DPREGO_DEBUG_ABI). */

#ifdef __cplusplus
}
#endif
#endif /* PREGO_AMD_H */
