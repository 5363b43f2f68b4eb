/* Probe / unit-test entry points of the MI355X PREGO library - NOT part of the product ABI (include/prego_amd.h).
 * They exist only in libprego_amd_debug.so = the same sources built with -DPREGO_DEBUG_ABI (python -m prego_amd.build builds both),
 * and serve the kernel-level GPU tests (tests/test_gpu_gemm.py, the attention kernel tests) and the measurement scripts under
 * scripts/.  Nothing under prego_amd/ calls them. */
#ifndef PREGO_AMD_DEBUG_H
#define PREGO_AMD_DEBUG_H
#include "prego_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Debug / probe: only the head kernel over n_slots equal slots x n_steps steps of caller-supplied relu(h) rows (16-bit, the handle's
 * operand type, time-major packed [n_steps][n_slots][hidden]); out [n_slots][n_steps][n_classes], argmax [n_slots][n_steps]. */
int prego_debug_head_only(prego_miniroad* h, int n_slots, int n_steps, const void* h_relu, float* out, int32_t* argmax,
                          const void* rowmap /* nullable: int32 (clip, frame) per row, as the pack kernel writes it */, prego_stream_t stream);

/* Debug only (env PREGO_GRU_STAMPS=1 at create): per-phase shader-cycle sums of workgroup 0 / wave 0 of the
 * recurrence kernel: out8[0..4] = rest of gather + mfma, step top -> first gather segment valid, reduce+barrier, gates+publish, outputs; [5] = gather retry rounds;
 * [6] = time steps.  Synchronises the device. */
int prego_miniroad_debug_stamps(prego_miniroad* h, unsigned long long* out8);

/* Probes of DESIGN.md section 5c (scripts/probes/xcd_overlap_probe.py), not product entry points: ONLY the recurrence kernel over
 * n_steps steps of n_slots equal slots dealt to gd groups (0 = all; gi: device 16-bit [n_steps * n_slots][3 H], h_relu: device 16-bit
 * [rows][H]); and the projection GEMM as a persistent worker that leaves XCDs below xcd_lo at once and claims 256 x 256 tiles from
 * *counter (device word, zero at launch). */
int prego_debug_recurrence_only(prego_miniroad* h, int n_slots, int n_steps, int gd, const void* gi, void* h_relu, prego_stream_t stream);
int prego_debug_gemm_worker(const void* A, const void* B, const float* bias, float* C, int M, int N, int K, int xcd_lo,
                            unsigned* counter, int grid, prego_stream_t stream);

/* The split pass's start handshake (DESIGN 5b "fail-safe") under test, and the replay entries counter collection needs.  One shot: applies
 * to the NEXT split pass of the handle.
 *   mode 1: the recurrence launch is withheld - the feed-forward launch's handshake wait runs out, it leaves without having written anything
 *           and prego_miniroad_forward re-runs the call as a chunked pass (tests/test_gpu_split.py)
 *   mode 2: the feed-forward launch is withheld (the recurrence's leader gives up; same outcome)
 *   mode 3: ONLY the feed-forward launch, handshake pre-decided and the recurrence's chunk counters pre-armed: the launch runs alone at full
 *           length (rocprofv3 --pmc serialises dispatches, so the pair can never be profiled together).  No head, outputs untouched
 *   mode 4: ONLY the recurrence launch on whatever (finite) rows an earlier pass left in the GI ring; no head, outputs untouched
 * split_state: fallbacks = calls re-run chunked behind a failed handshake, fails = failed handshakes (3 = chunked for good),
 * skip = eligible calls still to be kept chunked by the back-off, split_env = the handle's PREGO_SPLIT_PASS state (-1 auto, 0 never, R). */
int prego_debug_split_fault(prego_miniroad* h, int mode);
int prego_debug_split_state(const prego_miniroad* h, int64_t* fallbacks, int32_t* fails, int64_t* skip, int32_t* split_env);
/* unit-test hook: sets the handle's timeout word on the device (stream-ordered), as a recurrence / BPTT kernel that gave up would:
 * prego_miniroad_check then reports PREGO_ETIMEOUT and clears it; until then prego_miniroad_adamw_step changes nothing. */
int prego_debug_set_abort(prego_miniroad* h, unsigned value, prego_stream_t stream);
/* unit-test hook: how many hipMalloc calls / host-side stream or event waits the MiniROAD host code of this library has made so far in
 * this process (prego_miniroad_check's own synchronisation counts).  A test calls it around a hot call to hold "forward() allocates
 * nothing and waits for nothing" to zero (include/prego_amd.h, conventions). */
int prego_debug_alloc_count(int64_t* device_mallocs, int64_t* host_waits);
/* probe: a synthetic neighbour on XCDs >= xcd_lo for `ms` milliseconds - kind 1: back-to-back MFMAs on registers, no
 * memory traffic; kind 2: streaming reads of read_buf (+ one write per eight reads into write_buf), no matrix work; kind 3: both.
 * Launched on its own stream beside a replayed recurrence launch (prego_debug_split_fault mode 4) it separates what that launch loses to
 * power / clocks from what it loses to the fabric.  bytes: size of each of the two buffers (>= 1 MiB).  sink: one device float. */
int prego_debug_hog(int kind, int xcd_lo, int ms, const void* read_buf, void* write_buf, size_t bytes, float* sink, prego_stream_t stream);

/* Debug / microbenchmark only: C[M,N] fp32 = A[M,K] bf16 . B[N,K]^T bf16 + bias with a chosen kernel variant
 * (0 = 128x128, 1 = 256x128 three-stage, 9 = 256x256 two-stage, 12 = the ping-pong kernel = the production kernel of the projections; scripts/gemm_bench.py).  N % 128 == 0 (256 for variants >= 9), K % 64 == 0. */
int prego_debug_gemm_bf16(int variant, const void* A, const void* B, const float* bias, float* C, int M, int N, int K,
                          prego_stream_t stream);

/* Debug / unit test only: the attention backward kernels alone.  qs (= q * dh^-0.5), k, v: device bf16 [batch, heads, len, dh];
 * o, dout: device bf16 [batch, len, heads*dh]; lse: device fp32 [batch, heads, len] log-sum-exp of the scaled scores;
 * dqkv: device bf16 [batch*len, 3*heads*dh] (dq | dk | dv, dq wrt the unscaled q).  Synchronises the stream. */
int prego_debug_attention_bwd(int batch, int len, int heads, int dh, int causal, const void* qs, const void* k, const void* v,
                              const void* o, const void* dout, const float* lse, void* dqkv, prego_stream_t stream);

/* Debug / unit test only: the attention forward kernel alone (every head-dim / shape variant is reachable from here).
 * qs (= q * dh^-0.5): device bf16 [batch, heads, n_query, dh], queries at sequence positions 0 .. n_query-1; k, v: device bf16
 * [batch, heads, len, dh]; out: device bf16 [batch, n_query, heads*dh]; lse: device fp32 [batch, heads, n_query] or NULL.
 * Synchronises the stream. */
int prego_debug_attention_fwd(int batch, int n_query, int len, int heads, int dh, int causal, const void* qs, const void* k,
                              const void* v, void* out, float* lse, prego_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PREGO_AMD_DEBUG_H */
