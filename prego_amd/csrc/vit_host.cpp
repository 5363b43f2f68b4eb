// C ABI of the "Transformer" (ViTEnc) path and of the causal AttentionLayer op (include/prego_amd.h).
// bf16 (or, inference only, IEEE fp16) MFMA operands, fp32 accumulation / residual stream / LayerNorm / softmax.
#include "../../include/prego_amd.h"
#ifdef PREGO_DEBUG_ABI
#include "../../include/prego_amd_debug.h"
#endif
#include "kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

extern "C" const char* prego_last_error(void);
// error plumbing shared with miniroad.cpp
int prego_fail_(int code, const char* fmt, ...);
#define HIPCHK(x)                                                                                          \
  do {                                                                                                     \
    hipError_t e_ = (x);                                                                                   \
    if (e_ != hipSuccess) return prego_fail_(PREGO_EHIP, "%s failed: %s", #x, hipGetErrorString(e_));      \
  } while (0)
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct VitLayer {
  float *ln1_w, *ln1_b, *proj_b, *ln2_w, *ln2_b, *ff1_b, *ff2_b;
  void *qkv_w, *proj_w, *ff1_w, *ff2_w;    // bf16
  float *qkv_w32 = nullptr, *proj_w32 = nullptr, *ff1_w32 = nullptr, *ff2_w32 = nullptr;   // PREGO_F32 handles only
};
struct prego_vit {
  int d_rgb, d_flow, emb, mlp, heads, layers, window, ncls;
  void* enc_w = nullptr; float* enc_b = nullptr; float* cls = nullptr; float* pe = nullptr;
  std::vector<VitLayer> L;
  float *lnf_w = nullptr, *lnf_b = nullptr, *head_w = nullptr, *head_b = nullptr;
  std::vector<void*> allocs;
  bool have_weights = false;
  bool f16 = false;                // IEEE fp16 operands / 16-bit activations instead of bf16 (prego_vit_set_compute_dtype; inference only)
  bool f32 = false;                // fp32 operands everywhere (parity mode, prego_vit_forward only): the matrices are kept in fp32
  float* enc_w32 = nullptr;
  // training-mode dropout (cfg['dropout']; ViT.py:130 pe_dropout, Transformer.py:31 PreNormDrop, Transformer.py:41,46 FeedForward)
  float drop_p = 0.f;
  float attn_drop_p = 0.f;         // cfg['attn_dropout_rate']: attention probabilities (Attention.py:17,36) and proj_drop (Attention.py:19,40)
  unsigned long long drop_seed = 0;
};
// per-site mask seeds: site 0 = positional dropout, per layer: 1 = attention branch, 2 = after GELU, 3 = FFN output,
// 4 = attention probabilities, 5 = proj_drop
static inline unsigned long long site_seed(const prego_vit* h, int layer, int site) {
  return h->drop_seed + 0x1000ull * (unsigned long long)(layer + 1) * (site ? 1 : 0) + (unsigned long long)site;
}
static inline unsigned drop_thresh_of(const prego_vit* h) { return h->drop_p > 0.f ? (unsigned)((double)h->drop_p * 4294967296.0) : 0u; }
static inline float drop_scale_of(const prego_vit* h) { return h->drop_p > 0.f ? 1.f / (1.f - h->drop_p) : 1.f; }
static inline unsigned athr_of(const prego_vit* h) { return h->attn_drop_p > 0.f ? (unsigned)((double)h->attn_drop_p * 4294967296.0) : 0u; }
static inline float asc_of(const prego_vit* h) { return h->attn_drop_p > 0.f ? 1.f / (1.f - h->attn_drop_p) : 1.f; }
extern "C" int prego_vit_set_dropout(prego_vit* h, float p, float attn_p, uint64_t seed) {
  if (!h) return prego_fail_(PREGO_EINVAL, "handle is NULL");
  if (!(p >= 0.f && p < 1.f) || !(attn_p >= 0.f && attn_p < 1.f)) return prego_fail_(PREGO_EINVAL, "dropout p = %f, attn p = %f", (double)p, (double)attn_p);
  h->drop_p = p;
  h->attn_drop_p = attn_p;
  h->drop_seed = seed;
  return PREGO_OK;
}

static int dmalloc(prego_vit* h, void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) return prego_fail_(PREGO_EHIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
  h->allocs.push_back(*p);
  return 0;
}

extern "C" int prego_vit_create(prego_vit** out, int d_rgb, int d_flow, int emb, int mlp, int heads, int layers, int window,
                                int n_classes) {
  if (!out) return prego_fail_(PREGO_EINVAL, "out is NULL");
  *out = nullptr;
  const int din = d_rgb + d_flow;
  if (din <= 0 || din % 64) return prego_fail_(PREGO_EINVAL, "feature size %d must be a multiple of 64", din);
  if (emb % 512 || emb > 4096) return prego_fail_(PREGO_EINVAL, "embedding_dim %d must be a multiple of 512, <= 4096", emb);
  if (mlp % 128) return prego_fail_(PREGO_EINVAL, "hidden_dim (mlp) %d must be a multiple of 128", mlp);
  if (heads <= 0 || emb % heads) return prego_fail_(PREGO_EINVAL, "num_heads %d must divide embedding_dim", heads);
  const int dh = emb / heads;
  if (dh != 64 && dh != 128 && dh != 256) return prego_fail_(PREGO_EINVAL, "head dim %d: supported 64, 128, 256", dh);
  if (layers <= 0 || window <= 0 || n_classes <= 0) return prego_fail_(PREGO_EINVAL, "bad layers/window/classes");
  prego_vit* h = new prego_vit();
  h->d_rgb = d_rgb; h->d_flow = d_flow; h->emb = emb; h->mlp = mlp; h->heads = heads; h->layers = layers;
  h->window = window; h->ncls = n_classes;
  const size_t E = emb;
  int rc = 0;
  rc |= dmalloc(h, &h->enc_w, E * din * 2); rc |= dmalloc(h, (void**)&h->enc_b, E * 4);
  rc |= dmalloc(h, (void**)&h->cls, E * 4); rc |= dmalloc(h, (void**)&h->pe, (size_t)(window + 1) * E * 4);
  h->L.resize(layers);
  for (auto& l : h->L) {
    rc |= dmalloc(h, (void**)&l.ln1_w, E * 4); rc |= dmalloc(h, (void**)&l.ln1_b, E * 4);
    rc |= dmalloc(h, &l.qkv_w, 3 * E * E * 2); rc |= dmalloc(h, &l.proj_w, E * E * 2); rc |= dmalloc(h, (void**)&l.proj_b, E * 4);
    rc |= dmalloc(h, (void**)&l.ln2_w, E * 4); rc |= dmalloc(h, (void**)&l.ln2_b, E * 4);
    rc |= dmalloc(h, &l.ff1_w, (size_t)mlp * E * 2); rc |= dmalloc(h, (void**)&l.ff1_b, (size_t)mlp * 4);
    rc |= dmalloc(h, &l.ff2_w, E * mlp * 2); rc |= dmalloc(h, (void**)&l.ff2_b, E * 4);
  }
  rc |= dmalloc(h, (void**)&h->lnf_w, E * 4); rc |= dmalloc(h, (void**)&h->lnf_b, E * 4);
  rc |= dmalloc(h, (void**)&h->head_w, (size_t)n_classes * E * 4); rc |= dmalloc(h, (void**)&h->head_b, (size_t)n_classes * 4);
  if (rc) { for (void* p : h->allocs) (void)hipFree(p); delete h; return PREGO_EHIP; }
  *out = h;
  return PREGO_OK;
}

extern "C" void prego_vit_destroy(prego_vit* h) {
  if (!h) return;
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
}

extern "C" int prego_vit_set_compute_dtype(prego_vit* h, int compute_dtype) {
  if (!h) return prego_fail_(PREGO_EINVAL, "handle is NULL");
  if (compute_dtype != PREGO_BF16 && compute_dtype != PREGO_F16 && compute_dtype != PREGO_F32)
    return prego_fail_(PREGO_EINVAL, "ViTEnc compute_dtype %d: PREGO_BF16, PREGO_F16 or PREGO_F32", compute_dtype);
  if ((compute_dtype == PREGO_F16) != h->f16 || (compute_dtype == PREGO_F32) != h->f32) h->have_weights = false;   // other copies: re-ingest
  h->f16 = compute_dtype == PREGO_F16;
  h->f32 = compute_dtype == PREGO_F32;
  if (h->f32 && !h->enc_w32) {          // fp32 copies of the matrices, allocated with the first switch to this mode
    const size_t E = h->emb, din = h->d_rgb + h->d_flow, mlp = h->mlp;
    int rc = dmalloc(h, (void**)&h->enc_w32, E * din * 4);
    for (auto& l : h->L) {
      rc |= dmalloc(h, (void**)&l.qkv_w32, 3 * E * E * 4); rc |= dmalloc(h, (void**)&l.proj_w32, E * E * 4);
      rc |= dmalloc(h, (void**)&l.ff1_w32, mlp * E * 4); rc |= dmalloc(h, (void**)&l.ff2_w32, E * mlp * 4);
    }
    if (rc) { h->f32 = false; return PREGO_EHIP; }
  }
  return PREGO_OK;
}

extern "C" int prego_vit_num_tensors(const prego_vit* h) { return h ? 4 + 11 * h->layers + 4 : 0; }

extern "C" int prego_vit_set_weights(prego_vit* h, const float* const* t, int n_tensors, prego_stream_t stream) {
  if (!h || !t) return prego_fail_(PREGO_EINVAL, "NULL");
  if (n_tensors != prego_vit_num_tensors(h)) return prego_fail_(PREGO_EINVAL, "expected %d tensors, got %d", prego_vit_num_tensors(h), n_tensors);
  for (int i = 0; i < n_tensors; ++i) if (!t[i]) return prego_fail_(PREGO_EINVAL, "tensor %d is NULL", i);
  hipStream_t s = (hipStream_t)stream;
  const int E = h->emb, din = h->d_rgb + h->d_flow, mlp = h->mlp;
  int k = 0;
  auto f32 = [&](float* dst, size_t n) { return hipMemcpyAsync(dst, t[k++], n * 4, hipMemcpyDeviceToDevice, s); };
  if (h->f32) {                                    // parity mode: every tensor is kept as it comes
    HIPCHK(f32(h->enc_w32, (size_t)E * din));
    HIPCHK(f32(h->enc_b, E)); HIPCHK(f32(h->cls, E)); HIPCHK(f32(h->pe, (size_t)(h->window + 1) * E));
    for (auto& l : h->L) {
      HIPCHK(f32(l.ln1_w, E)); HIPCHK(f32(l.ln1_b, E));
      HIPCHK(f32(l.qkv_w32, (size_t)3 * E * E)); HIPCHK(f32(l.proj_w32, (size_t)E * E));
      HIPCHK(f32(l.proj_b, E)); HIPCHK(f32(l.ln2_w, E)); HIPCHK(f32(l.ln2_b, E));
      HIPCHK(f32(l.ff1_w32, (size_t)mlp * E)); HIPCHK(f32(l.ff1_b, mlp));
      HIPCHK(f32(l.ff2_w32, (size_t)E * mlp)); HIPCHK(f32(l.ff2_b, E));
    }
    HIPCHK(f32(h->lnf_w, E)); HIPCHK(f32(h->lnf_b, E)); HIPCHK(f32(h->head_w, (size_t)h->ncls * E)); HIPCHK(f32(h->head_b, h->ncls));
    HIPCHK(hipGetLastError());
    h->have_weights = true;
    return PREGO_OK;
  }
  launch_pad_convert(true, t[k++], E, din, din, h->enc_w, E, din, s, h->f16);
  HIPCHK(f32(h->enc_b, E)); HIPCHK(f32(h->cls, E)); HIPCHK(f32(h->pe, (size_t)(h->window + 1) * E));
  for (auto& l : h->L) {
    HIPCHK(f32(l.ln1_w, E)); HIPCHK(f32(l.ln1_b, E));
    launch_pad_convert(true, t[k++], 3 * E, E, E, l.qkv_w, 3 * E, E, s, h->f16);
    launch_pad_convert(true, t[k++], E, E, E, l.proj_w, E, E, s, h->f16);
    HIPCHK(f32(l.proj_b, E)); HIPCHK(f32(l.ln2_w, E)); HIPCHK(f32(l.ln2_b, E));
    launch_pad_convert(true, t[k++], mlp, E, E, l.ff1_w, mlp, E, s, h->f16);
    HIPCHK(f32(l.ff1_b, mlp));
    launch_pad_convert(true, t[k++], E, mlp, mlp, l.ff2_w, E, mlp, s, h->f16);
    HIPCHK(f32(l.ff2_b, E));
  }
  HIPCHK(f32(h->lnf_w, E)); HIPCHK(f32(h->lnf_b, E)); HIPCHK(f32(h->head_w, (size_t)h->ncls * E)); HIPCHK(f32(h->head_b, h->ncls));
  HIPCHK(hipGetLastError());
  h->have_weights = true;
  return PREGO_OK;
}

// optimizer.step() on the handle's tensors: fused AdamW that also rewrites the handle's operand copies (csrc/optim.hip)
extern "C" int prego_vit_adamw_step(prego_vit* h, float* const* params, const float* const* grads, float* const* exp_avg,
                                    float* const* exp_avg_sq, int n_tensors, int64_t step, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, prego_stream_t stream) {
  if (h && (h->f16 || h->f32)) return prego_fail_(PREGO_EINVAL, "prego_vit_adamw_step on an fp16- / fp32-operand handle: training runs on bf16 handles");
  if (!h || !params || !grads || !exp_avg || !exp_avg_sq) return prego_fail_(PREGO_EINVAL, "vit adamw: NULL argument");
  if (!h->have_weights) return prego_fail_(PREGO_EINVAL, "vit adamw step before set_weights");
  if (n_tensors != prego_vit_num_tensors(h)) return prego_fail_(PREGO_EINVAL, "expected %d tensors, got %d", prego_vit_num_tensors(h), n_tensors);
  const long long E = h->emb, din = h->d_rgb + h->d_flow, mlp = h->mlp;
  // set_weights order; matrices (bf16 copies) and vectors (fp32 copies) go in two launches, one copy element type each
  std::vector<float*> pm, pv_; std::vector<const float*> gm, gv; std::vector<float*> mm, mv, vm, vv; std::vector<void*> cm, cv;
  std::vector<long long> nm, nv;
  int k = 0;
  auto mat = [&](void* copy, long long n) { pm.push_back(params[k]); gm.push_back(grads[k]); mm.push_back(exp_avg[k]); vm.push_back(exp_avg_sq[k]); cm.push_back(copy); nm.push_back(n); ++k; };
  auto vec = [&](void* copy, long long n) { pv_.push_back(params[k]); gv.push_back(grads[k]); mv.push_back(exp_avg[k]); vv.push_back(exp_avg_sq[k]); cv.push_back(copy); nv.push_back(n); ++k; };
  for (int i = 0; i < n_tensors; ++i) if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]) return prego_fail_(PREGO_EINVAL, "vit adamw: tensor %d is NULL", i);
  mat(h->enc_w, E * din); vec(h->enc_b, E); vec(h->cls, E); vec(h->pe, (long long)(h->window + 1) * E);
  for (auto& l : h->L) {
    vec(l.ln1_w, E); vec(l.ln1_b, E); mat(l.qkv_w, 3 * E * E); mat(l.proj_w, E * E); vec(l.proj_b, E); vec(l.ln2_w, E); vec(l.ln2_b, E);
    mat(l.ff1_w, mlp * E); vec(l.ff1_b, mlp); mat(l.ff2_w, E * mlp); vec(l.ff2_b, E);
  }
  vec(h->lnf_w, E); vec(h->lnf_b, E); vec(h->head_w, (long long)h->ncls * E); vec(h->head_b, h->ncls);
  hipStream_t s = (hipStream_t)stream;
  if (launch_adamw((int)pm.size(), pm.data(), gm.data(), mm.data(), vm.data(), cm.data(), nm.data(), true, step, lr, beta1, beta2, eps, weight_decay, s) ||
      launch_adamw((int)pv_.size(), pv_.data(), gv.data(), mv.data(), vv.data(), cv.data(), nv.data(), false, step, lr, beta1, beta2, eps, weight_decay, s))
    return prego_fail_(PREGO_EINVAL, "vit adamw: bad step %lld", (long long)step);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

struct VitWs { size_t xb, enc, x, xn, q, k, vn, ao, f, x0, q0, ao0, xn0, f0, total; };
static VitWs vit_ws(const prego_vit* h, int B) {
  const size_t E = h->emb, T = h->window, N = T + 1, din = h->d_rgb + h->d_flow;
  VitWs w{};
  size_t off = 0;
  auto put = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  w.xb = put((size_t)B * T * din * 2); w.enc = put((size_t)B * T * E * 4); w.x = put((size_t)B * N * E * 4);
  w.xn = put((size_t)B * N * E * 2); w.q = put((size_t)B * N * E * 2); w.k = put((size_t)B * N * E * 2);
  w.vn = put((size_t)B * N * E * 2); w.ao = put((size_t)B * N * E * 2); w.f = put((size_t)B * N * h->mlp * 2);
  // last block, token 0 only (ViT.py:136 reads x[:, 0]): one row per window
  w.x0 = put((size_t)B * E * 4); w.q0 = put((size_t)B * E * 2); w.ao0 = put((size_t)B * E * 2); w.xn0 = put((size_t)B * E * 2);
  w.f0 = put((size_t)B * h->mlp * 2);
  w.total = off;
  return w;
}
// fp32-operand mode: fp32 rows everywhere, every block on every token
struct VitWs32 { size_t xc, enc, x, xn, qkv, ao, tmp, total; };
static VitWs32 vit_ws32(const prego_vit* h, int B) {
  const size_t E = h->emb, T = h->window, N = T + 1, din = h->d_rgb + h->d_flow, M = (size_t)B * N, mlp = h->mlp;
  VitWs32 w{};
  size_t off = 0;
  auto put = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  w.xc = put((size_t)B * T * din * 4); w.enc = put((size_t)B * T * E * 4); w.x = put(M * E * 4); w.xn = put(M * E * 4);
  w.qkv = put(M * 3 * E * 4); w.ao = put(M * E * 4); w.tmp = put(M * std::max(E, mlp) * 4);
  w.total = off;
  return w;
}
extern "C" size_t prego_vit_workspace_bytes(const prego_vit* h, int batch) {
  if (!h || batch <= 0) return 0;
  return h->f32 ? vit_ws32(h, batch).total : vit_ws(h, batch).total;
}
// ViTEnc.forward (ViT.py:117-143) with fp32 operands: the projections on the exact-fp32 MFMA GEMM, everything else in fp32 too
static int vit_forward_f32(prego_vit* h, int B, const float* rgb, const float* flow, float* out_logits, int causal, char* ws,
                           hipStream_t s) {
  const VitWs32 w = vit_ws32(h, B);
  const int T = h->window, N = T + 1, E = h->emb, din = h->d_rgb + h->d_flow, M = B * N, dh = E / h->heads, mlp = h->mlp;
  float* xc = (float*)(ws + w.xc); float* enc = (float*)(ws + w.enc); float* x = (float*)(ws + w.x); float* xn = (float*)(ws + w.xn);
  float* qkv = (float*)(ws + w.qkv); float* ao = (float*)(ws + w.ao); float* tmp = (float*)(ws + w.tmp);
  launch_cat_rows_f32(rgb, flow, B * T, h->d_rgb, h->d_flow, xc, s);
  launch_gemm_f32_nt(xc, din, h->enc_w32, din, h->enc_b, enc, E, B * T, E, din, s);                     // ViT.py:124
  launch_vit_tokens(enc, h->cls, h->pe, B, T, E, x, s);                                                  // ViT.py:125-129
  for (int li = 0; li < h->layers; ++li) {
    const VitLayer& l = h->L[li];
    launch_ln_relu(false, x, l.ln1_w, l.ln1_b, M, E, 1e-5f, xn, nullptr, 0.f, 0, 0, s, 0);
    launch_gemm_f32_nt(xn, E, l.qkv_w32, E, nullptr, qkv, 3 * E, M, 3 * E, E, s);                       // Attention.py:23-27
    if (launch_attention_f32(qkv, 3 * E, 0, E, 2 * E, ao, B, N, N, h->heads, dh, causal, 1.0f / sqrtf((float)dh), s))
      return prego_fail_(PREGO_EINVAL, "attention launch failed");
    launch_gemm_f32_nt(ao, E, l.proj_w32, E, l.proj_b, tmp, E, M, E, E, s);
    launch_add_rows(x, tmp, (size_t)M * E, s);                                                           // Transformer.py:24-32
    launch_ln_relu(false, x, l.ln2_w, l.ln2_b, M, E, 1e-5f, xn, nullptr, 0.f, 0, 0, s, 0);
    launch_gemm_f32_nt(xn, E, l.ff1_w32, E, l.ff1_b, tmp, mlp, M, mlp, E, s);
    launch_gelu_f32(tmp, (size_t)M * mlp, s);                                                            // Transformer.py:40
    launch_gemm_f32_nt(tmp, mlp, l.ff2_w32, mlp, l.ff2_b, xn, E, M, E, mlp, s);
    launch_add_rows(x, xn, (size_t)M * E, s);
  }
  launch_vit_head(x, B, N, E, h->lnf_w, h->lnf_b, h->head_w, h->head_b, h->ncls, out_logits, s);       // token 0, ViT.py:134-141
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// one pre-norm encoder block on the fp32 residual stream x [M = B*N, E] (Transformer.py:60-77)
static int encoder_block(const prego_vit* h, const VitLayer& l, float* x, char* ws, const VitWs& w, int B, int N, int causal,
                         hipStream_t s) {
  const int E = h->emb, M = B * N, dh = E / h->heads;
  launch_ln_relu(true, x, l.ln1_w, l.ln1_b, M, E, 1e-5f, ws + w.xn, nullptr, 0.f, 0, 0, s, 0, false, h->f16);
  GemmEpi e{}; e.f16 = h->f16 ? 1 : 0;
  e.mode = EPI_QKV; e.q = ws + w.q; e.k = ws + w.k; e.vn = ws + w.vn; e.n_tok = N; e.heads = h->heads;
  e.dh = dh; e.emb = E; e.q_scale = 1.0f / sqrtf((float)dh);                      // Attention.py:14 (dh^-0.5)
  launch_gemm_bf16_nt_epi(ws + w.xn, E, l.qkv_w, E, nullptr, nullptr, 0, M, 3 * E, E, e, s);
  if (launch_flash_attention_v2(ws + w.q, ws + w.k, ws + w.vn, ws + w.ao, B, N, N, h->heads, dh, causal, s, nullptr, 0, 1.f, 0, h->f16)) return -1;
  GemmEpi r{}; r.f16 = h->f16 ? 1 : 0; r.mode = EPI_RESIDUAL;
  launch_gemm_bf16_nt_epi(ws + w.ao, E, l.proj_w, E, l.proj_b, x, E, M, E, E, r, s);          // x += proj(attn)
  launch_ln_relu(true, x, l.ln2_w, l.ln2_b, M, E, 1e-5f, ws + w.xn, nullptr, 0.f, 0, 0, s, 0, false, h->f16);
  GemmEpi g{}; g.f16 = h->f16 ? 1 : 0; g.mode = EPI_GELU_BF16; g.out_b = ws + w.f;
  launch_gemm_bf16_nt_epi(ws + w.xn, E, l.ff1_w, E, l.ff1_b, nullptr, h->mlp, M, h->mlp, E, g, s);   // gelu(W1 x + b1)
  launch_gemm_bf16_nt_epi(ws + w.f, h->mlp, l.ff2_w, h->mlp, l.ff2_b, x, E, M, E, h->mlp, r, s);     // x += W2 . + b2
  return 0;
}

// The LAST encoder block when only token 0 of its output is read (ViT.py:136 `x[:, 0]`, then pre_head_ln and mlp_head): keys and
// values are needed for every token, but the query, the attention output, the projection, the residual stream and the whole FFN
// only for token 0 of each window - B rows instead of B*N.  Exact (the skipped rows never reach the logits); with num_layers = 1
// it removes 44 % of a window's FLOPs.  Leaves the block's token-0 output in x0 [B, E].
// have_xn: the caller already wrote LayerNorm1(x) to w.xn and token 0 of every window to w.x0 (the sliding-window token kernel
// does both): x is not read at all.
static int encoder_block_token0(const prego_vit* h, const VitLayer& l, const float* x, char* ws, const VitWs& w, int B, int N,
                                int causal, hipStream_t s, bool have_xn = false) {
  const int E = h->emb, M = B * N, dh = E / h->heads;
  if (!have_xn) launch_ln_relu(true, x, l.ln1_w, l.ln1_b, M, E, 1e-5f, ws + w.xn, nullptr, 0.f, 0, 0, s, 0, false, h->f16);
  GemmEpi e{}; e.f16 = h->f16 ? 1 : 0;
  e.mode = EPI_QKV; e.q = ws + w.q0; e.k = ws + w.k; e.vn = ws + w.vn; e.n_tok = N; e.heads = h->heads; e.dh = dh; e.emb = E;
  e.q_scale = 1.0f / sqrtf((float)dh);
  e.which0 = 1;                                                                                // k | v for every token
  launch_gemm_bf16_nt_epi(ws + w.xn, E, (const char*)l.qkv_w + (size_t)E * E * 2, E, nullptr, nullptr, 0, M, 2 * E, E, e, s);
  e.which0 = 0; e.n_tok = 1;                                                                   // q for token 0: rows b * N of xn
  launch_gemm_bf16_nt_epi(ws + w.xn, N * E, l.qkv_w, E, nullptr, nullptr, 0, B, E, E, e, s);
  if (launch_flash_attention_v2(ws + w.q0, ws + w.k, ws + w.vn, ws + w.ao0, B, 1, N, h->heads, dh, causal, s, nullptr, 0, 1.f, 0, h->f16)) return -1;
  float* x0 = (float*)(ws + w.x0);
  if (!have_xn && hipMemcpy2DAsync(x0, (size_t)E * 4, x, (size_t)N * E * 4, (size_t)E * 4, B, hipMemcpyDeviceToDevice, s) != hipSuccess) return -1;
  GemmEpi r{}; r.f16 = h->f16 ? 1 : 0; r.mode = EPI_RESIDUAL;
  launch_gemm_bf16_nt_epi(ws + w.ao0, E, l.proj_w, E, l.proj_b, x0, E, B, E, E, r, s);
  launch_ln_relu(true, x0, l.ln2_w, l.ln2_b, B, E, 1e-5f, ws + w.xn0, nullptr, 0.f, 0, 0, s, 0, false, h->f16);
  GemmEpi g{}; g.f16 = h->f16 ? 1 : 0; g.mode = EPI_GELU_BF16; g.out_b = ws + w.f0;
  launch_gemm_bf16_nt_epi(ws + w.xn0, E, l.ff1_w, E, l.ff1_b, nullptr, h->mlp, B, h->mlp, E, g, s);
  launch_gemm_bf16_nt_epi(ws + w.f0, h->mlp, l.ff2_w, h->mlp, l.ff2_b, x0, E, B, E, h->mlp, r, s);
  return 0;
}

extern "C" int prego_vit_forward(prego_vit* h, int batch, const float* rgb, const float* flow, float* out_logits, int flags,
                                 void* workspace, size_t workspace_bytes, prego_stream_t stream) {
  if (!h || !out_logits || !workspace) return prego_fail_(PREGO_EINVAL, "NULL argument");
  if (!h->have_weights) return prego_fail_(PREGO_EINVAL, "forward before set_weights");
  if (batch <= 0) return prego_fail_(PREGO_EINVAL, "batch %d", batch);
  if ((h->d_rgb > 0 && !rgb) || (h->d_flow > 0 && !flow && h->d_rgb == 0)) return prego_fail_(PREGO_EINVAL, "missing input");
  if (h->f32) {
    if (workspace_bytes < vit_ws32(h, batch).total) return prego_fail_(PREGO_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, vit_ws32(h, batch).total);
    return vit_forward_f32(h, batch, rgb, flow, out_logits, (flags & 1) ? 1 : 0, (char*)workspace, (hipStream_t)stream);
  }
  const VitWs w = vit_ws(h, batch);
  if (workspace_bytes < w.total) return prego_fail_(PREGO_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, w.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int B = batch, T = h->window, N = T + 1, E = h->emb, din = h->d_rgb + h->d_flow;
  const int causal = (flags & 1) ? 1 : 0;
  const bool all_rows = (flags & 2) != 0;             // bit 1 (debug / A-B): run the last block on every token as well
  launch_cat_convert(rgb, flow, B * T, h->d_rgb, h->d_flow, ws + w.xb, s, h->f16);
  // ViT.py:125-129: the encoding GEMM writes the residual stream itself (frame rows + positional rows in its epilogue: no fp32
  // encoding tensor, no token kernel pass); the cls row of every window is B short rows
  static const bool no_epi_tokens = prego_tune_env("PREGO_VIT_TOKENS_KERNEL") != nullptr;       // A/B: the separate token kernel
  if (no_epi_tokens || B * T < 4096) {       // small batches: the 128 x 128 kernel's per-element epilogue costs more than the token kernel
    launch_gemm_bf16_nt(ws + w.xb, din, h->enc_w, din, h->enc_b, (float*)(ws + w.enc), E, B * T, E, din, s, h->f16);
    launch_vit_tokens((const float*)(ws + w.enc), h->cls, h->pe, B, T, E, (float*)(ws + w.x), s);
  } else {
    GemmEpi te{}; te.f16 = h->f16 ? 1 : 0;
    te.mode = EPI_TOKENS; te.pe = h->pe; te.n_tok = T;
    launch_gemm_bf16_nt_epi(ws + w.xb, din, h->enc_w, din, h->enc_b, (float*)(ws + w.x), E, B * T, E, din, te, s);
    launch_vit_cls_rows(h->cls, h->pe, B, T, E, (float*)(ws + w.x), s);
  }
  for (int li = 0; li < h->layers; ++li) {
    const bool last = li + 1 == h->layers && !all_rows;
    const int rc = last ? encoder_block_token0(h, h->L[li], (const float*)(ws + w.x), ws, w, B, N, causal, s)
                        : encoder_block(h, h->L[li], (float*)(ws + w.x), ws, w, B, N, causal, s);
    if (rc) return prego_fail_(PREGO_EINVAL, "encoder block launch failed");
  }
  if (all_rows) launch_vit_head((const float*)(ws + w.x), B, N, E, h->lnf_w, h->lnf_b, h->head_w, h->head_b, h->ncls, out_logits, s);
  else launch_vit_head((const float*)(ws + w.x0), B, 1, E, h->lnf_w, h->lnf_b, h->head_w, h->head_b, h->ncls, out_logits, s);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// ================================================================================================
// per-frame inference over a whole video (the eval loop of trainer/eval.py:36-56 for model: 'Transformer')
// ================================================================================================
// ViTEnc emits one logit vector per window (ViT.py:136-141) and needs T == window_size; the per-frame output the eval loop
// collects is the window ENDING at every frame, zero feature rows in front of the video - the windows the training loader cuts
// (dataset.py:53-55,96-103) at stride 1.  Frame f sits in up to `window` windows, and linear_encoding (ViT.py:124, half of a
// window's FLOPs at one layer) does not depend on the position inside the window (the positional table is added afterwards,
// ViT.py:129): it runs ONCE per frame here.  Windows are processed `wb` at a time through the same blocks as prego_vit_forward.
struct VitFramesWs { size_t xb, enc, win; VitWs w; size_t total; };
static VitFramesWs vit_frames_ws(const prego_vit* h, int n_frames, int wb) {
  VitFramesWs f{};
  const size_t E = h->emb, din = h->d_rgb + h->d_flow;
  size_t off = 0;
  auto put = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  f.xb = put((size_t)n_frames * din * 2); f.enc = put((size_t)n_frames * E * 4);
  f.w = vit_ws(h, wb);
  f.win = put(f.w.total);
  f.total = off;
  return f;
}
extern "C" size_t prego_vit_frames_workspace_bytes(const prego_vit* h, int n_frames, int windows_per_batch) {
  return (h && n_frames > 0 && windows_per_batch > 0) ? vit_frames_ws(h, n_frames, windows_per_batch).total : 0;
}
extern "C" int prego_vit_forward_frames(prego_vit* h, int n_frames, const float* rgb, const float* flow, float* out_logits,
                                        int32_t* out_argmax, int windows_per_batch, int flags, void* workspace, size_t workspace_bytes,
                                        prego_stream_t stream) {
  if (!h || !out_logits || !workspace) return prego_fail_(PREGO_EINVAL, "NULL argument");
  if (h->f32) return prego_fail_(PREGO_EINVAL, "prego_vit_forward_frames on an fp32-operand handle: the parity mode covers prego_vit_forward");
  if (!h->have_weights) return prego_fail_(PREGO_EINVAL, "forward before set_weights");
  if (n_frames <= 0 || windows_per_batch <= 0) return prego_fail_(PREGO_EINVAL, "n_frames %d, windows_per_batch %d", n_frames, windows_per_batch);
  if ((h->d_rgb > 0 && !rgb) || (h->d_rgb == 0 && !flow)) return prego_fail_(PREGO_EINVAL, "missing input");
  const int wb = windows_per_batch;
  const VitFramesWs f = vit_frames_ws(h, n_frames, wb);
  if (workspace_bytes < f.total) return prego_fail_(PREGO_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, f.total);
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)workspace;
  char* ws = base + f.win;                      // the per-batch arena, laid out as prego_vit_forward's
  const VitWs& w = f.w;
  const int T = h->window, N = T + 1, E = h->emb, din = h->d_rgb + h->d_flow;
  const int causal = (flags & 1) ? 1 : 0;
  float* enc = (float*)(base + f.enc);
  launch_cat_convert(rgb, flow, n_frames, h->d_rgb, h->d_flow, base + f.xb, s, h->f16);
  launch_gemm_bf16_nt(base + f.xb, din, h->enc_w, din, h->enc_b, enc, E, n_frames, E, din, s, h->f16);       // ViT.py:124, once per frame
  const bool fused = h->layers == 1;            // one layer: the token kernel writes LayerNorm1(x) and x0; x is never materialised
  for (int t0 = 0; t0 < n_frames; t0 += wb) {
    const int B = std::min(wb, n_frames - t0);
    const VitLayer& l0 = h->L[0];
    launch_vit_sliding_tokens(enc, h->enc_b, h->cls, h->pe, t0, B, T, E, fused ? nullptr : (float*)(ws + w.x), l0.ln1_w, l0.ln1_b,
                              fused ? ws + w.xn : nullptr, fused ? (float*)(ws + w.x0) : nullptr, s, h->f16);
    for (int li = 0; li < h->layers; ++li) {
      const bool last = li + 1 == h->layers;
      const int rc = last ? encoder_block_token0(h, h->L[li], (const float*)(ws + w.x), ws, w, B, N, causal, s, fused)
                          : encoder_block(h, h->L[li], (float*)(ws + w.x), ws, w, B, N, causal, s);
      if (rc) return prego_fail_(PREGO_EINVAL, "encoder block launch failed");
    }
    launch_vit_head((const float*)(ws + w.x0), B, 1, E, h->lnf_w, h->lnf_b, h->head_w, h->head_b, h->ncls,
                    out_logits + (size_t)t0 * h->ncls, s, out_argmax ? (int*)out_argmax + t0 : nullptr);
  }
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// ================================================================================================
// training: forward that keeps activations + backward (trainer/train.py:20-24 over ViT.py:117-143)
// ================================================================================================
struct VitLayerKeep { size_t x_in, st1, xn1, q, k, vn, lse, ao, x_mid, st2, xn2, u, f; };
struct VitTrainWs {
  size_t xb, enc, x;                     // inputs as bf16, encoding GEMM output, residual stream (final value after forward)
  std::vector<VitLayerKeep> L;
  // backward scratch
  size_t dx, dxm, dxb, tmp, du, dub, dO, dqkv, delta, T1, T2, WT, part, head, denc, vec;
  size_t total;
  int npad, Mp, MTp;
};
static VitTrainWs vit_train_ws(const prego_vit* h, int B) {
  const size_t E = h->emb, T = h->window, N = T + 1, din = h->d_rgb + h->d_flow, mlp = h->mlp, M = (size_t)B * N;
  VitTrainWs w{};
  w.npad = (int)align_up(N, 64);
  w.Mp = (int)align_up(M, 64);
  w.MTp = (int)align_up((size_t)B * T, 64);
  size_t off = 0;
  auto put = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  w.xb = put((size_t)B * T * din * 2); w.enc = put((size_t)B * T * E * 4); w.x = put(M * E * 4);
  w.L.resize(h->layers);
  for (auto& l : w.L) {
    l.x_in = put(M * E * 4); l.st1 = put(M * 8); l.xn1 = put(M * E * 2);
    l.q = put(M * E * 2); l.k = put(M * E * 2); l.vn = put(M * E * 2);
    l.lse = put((size_t)B * h->heads * N * 4); l.ao = put(M * E * 2);
    l.x_mid = put(M * E * 4); l.st2 = put(M * 8); l.xn2 = put(M * E * 2);
    l.u = put(M * mlp * 4); l.f = put(M * mlp * 2);
  }
  const size_t wide = std::max<size_t>(std::max<size_t>(3 * E, mlp), din);
  w.dx = put(M * E * 4); w.dxm = put(M * E * 4); w.dxb = put(M * E * 2); w.tmp = put(M * std::max<size_t>(E, mlp) * 4);
  w.du = put(M * mlp * 4); w.dub = put(M * mlp * 2); w.dO = put(M * E * 2); w.dqkv = put(M * 3 * E * 2);
  w.delta = put((size_t)B * h->heads * N * 4);
  w.T1 = put(wide * (size_t)w.Mp * 2); w.T2 = put(wide * (size_t)w.Mp * 2);
  w.WT = put(3 * E * E * 2);
  w.part = put(std::max<size_t>(((size_t)M / 64 + 2) * std::max<size_t>(mlp, E), ((size_t)M / 4 + 2) * 2 * E) * 4);
  w.head = put((size_t)3 * B * E * 4); w.denc = put((size_t)B * T * E * 4); w.vec = put(4 * std::max<size_t>(E, mlp) * 4);
  w.total = off;
  return w;
}
extern "C" size_t prego_vit_train_workspace_bytes(const prego_vit* h, int batch) {
  return (h && batch > 0) ? vit_train_ws(h, batch).total : 0;
}

extern "C" int prego_vit_forward_train(prego_vit* h, int batch, const float* rgb, const float* flow, float* out_logits, int flags,
                                       void* workspace, size_t workspace_bytes, prego_stream_t stream) {
  if (h && (h->f16 || h->f32)) return prego_fail_(PREGO_EINVAL, "prego_vit_forward_train on an fp16- / fp32-operand handle: training runs on bf16 handles");
  if (!h || !out_logits || !workspace) return prego_fail_(PREGO_EINVAL, "NULL argument");
  if (!h->have_weights) return prego_fail_(PREGO_EINVAL, "forward before set_weights");
  if (batch <= 0) return prego_fail_(PREGO_EINVAL, "batch %d", batch);
  if ((h->d_rgb > 0 && !rgb) || (h->d_rgb == 0 && !flow)) return prego_fail_(PREGO_EINVAL, "missing input");
  const VitTrainWs w = vit_train_ws(h, batch);
  if (workspace_bytes < w.total) return prego_fail_(PREGO_EWORKSPACE, "training workspace %zu < %zu", workspace_bytes, w.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int B = batch, T = h->window, N = T + 1, E = h->emb, din = h->d_rgb + h->d_flow, M = B * N, dh = E / h->heads;
  const int causal = (flags & 1) ? 1 : 0;
  float* x = (float*)(ws + w.x);
  launch_cat_convert(rgb, flow, B * T, h->d_rgb, h->d_flow, ws + w.xb, s);
  launch_gemm_bf16_nt(ws + w.xb, din, h->enc_w, din, h->enc_b, (float*)(ws + w.enc), E, B * T, E, din, s);      // the eval forward's kernel: the keeping forward is the same arithmetic, bit for bit (test G5b)
  const unsigned dthr = drop_thresh_of(h);
  const float dsc = drop_scale_of(h);
  launch_vit_tokens((const float*)(ws + w.enc), h->cls, h->pe, B, T, E, x, s, dthr, dsc, site_seed(h, 0, 0));
  for (int li = 0; li < h->layers; ++li) {
    const VitLayer& l = h->L[li];
    const VitLayerKeep& k = w.L[li];
    HIPCHK(hipMemcpyAsync(ws + k.x_in, x, (size_t)M * E * 4, hipMemcpyDeviceToDevice, s));
    launch_ln_relu(true, x, l.ln1_w, l.ln1_b, M, E, 1e-5f, ws + k.xn1, (float*)(ws + k.st1), 0.f, 0, 0, s, 0);
    GemmEpi e{};
    e.mode = EPI_QKV; e.q = ws + k.q; e.k = ws + k.k; e.vn = ws + k.vn; e.n_tok = N;
    e.heads = h->heads; e.dh = dh; e.emb = E; e.q_scale = 1.0f / sqrtf((float)dh);
    launch_gemm_bf16_nt_epi(ws + k.xn1, E, l.qkv_w, E, nullptr, nullptr, 0, M, 3 * E, E, e, s);
    if (launch_flash_attention_v2(ws + k.q, ws + k.k, ws + k.vn, ws + k.ao, B, N, N, h->heads, dh, causal, s, (float*)(ws + k.lse),
                                  athr_of(h), asc_of(h), site_seed(h, li, 4)))
      return prego_fail_(PREGO_EINVAL, "attention launch failed");
    GemmEpi r{}; r.mode = EPI_RESIDUAL; r.drop_thresh = dthr; r.drop_scale = dsc; r.drop_seed = site_seed(h, li, 1);
    r.drop2_thresh = athr_of(h); r.drop2_scale = asc_of(h); r.drop2_seed = site_seed(h, li, 5);        // proj_drop, then PreNormDrop
    launch_gemm_bf16_nt_epi(ws + k.ao, E, l.proj_w, E, l.proj_b, x, E, M, E, E, r, s);              // x += drop(proj(attn))
    HIPCHK(hipMemcpyAsync(ws + k.x_mid, x, (size_t)M * E * 4, hipMemcpyDeviceToDevice, s));
    launch_ln_relu(true, x, l.ln2_w, l.ln2_b, M, E, 1e-5f, ws + k.xn2, (float*)(ws + k.st2), 0.f, 0, 0, s, 0);
    GemmEpi g{}; g.mode = EPI_GELU_BF16; g.out_b = ws + k.f; g.pre_f32 = (float*)(ws + k.u);
    g.drop_thresh = dthr; g.drop_scale = dsc; g.drop_seed = site_seed(h, li, 2);
    launch_gemm_bf16_nt_epi(ws + k.xn2, E, l.ff1_w, E, l.ff1_b, nullptr, h->mlp, M, h->mlp, E, g, s);   // f = drop(gelu(.))
    r.drop_seed = site_seed(h, li, 3); r.drop2_thresh = 0;
    launch_gemm_bf16_nt_epi(ws + k.f, h->mlp, l.ff2_w, h->mlp, l.ff2_b, x, E, M, E, h->mlp, r, s);     // x += drop(W2 f + b2)
  }
  launch_vit_head(x, B, N, E, h->lnf_w, h->lnf_b, h->head_w, h->head_b, h->ncls, out_logits, s);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// Weight gradient C[Mo, No] = A^T . B over the rows (the reduction dimension of every weight gradient is the row index) and, when
// bias_out is given, the bias gradient colsum(A): the k-major GEMM of csrc/gemm_tn.hip reads A_rows [R, Mo] and B_rows [R, No] (bf16) as
// they lie in memory - round 3 transposed both first and summed the bias in two more launches.  Rp = R padded to 64 (zeros are read).
static int wgrad(const void* a_rows, int Mo, const void* b_rows, int No, int R, int Rp, float* out, float* bias_out, hipStream_t s) {
  return launch_gemm_bf16_tn(true, true, a_rows, Mo, b_rows, No, nullptr, out, No, Mo, No, Rp, R, bias_out, s);
}
// Input gradient C[R, No] = A [R, Ko] . W [Ko, No]: the weight as nn.Linear stores it IS the [K][N] operand (no transposed copy)
static int dgrad(const void* a_rows, int Ko, const void* w, int No, int R, float* out, hipStream_t s, void* out16 = nullptr) {
  return launch_gemm_bf16_tn(false, true, a_rows, Ko, w, No, nullptr, out, No, R, No, Ko, Ko, nullptr, s, out16);
}

extern "C" int prego_vit_backward(prego_vit* h, int batch, const float* dlogits, float* const* grads, int n_tensors, int flags,
                                  void* workspace, size_t workspace_bytes, prego_stream_t stream) {
  if (h && (h->f16 || h->f32)) return prego_fail_(PREGO_EINVAL, "prego_vit_backward on an fp16- / fp32-operand handle: training runs on bf16 handles");
  if (!h || !dlogits || !grads || !workspace) return prego_fail_(PREGO_EINVAL, "NULL argument");
  if (n_tensors != prego_vit_num_tensors(h)) return prego_fail_(PREGO_EINVAL, "expected %d gradient tensors, got %d", prego_vit_num_tensors(h), n_tensors);
  for (int i = 0; i < n_tensors; ++i) if (!grads[i]) return prego_fail_(PREGO_EINVAL, "gradient tensor %d is NULL", i);
  const VitTrainWs w = vit_train_ws(h, batch);
  if (workspace_bytes < w.total) return prego_fail_(PREGO_EWORKSPACE, "training workspace %zu < %zu", workspace_bytes, w.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int B = batch, T = h->window, N = T + 1, E = h->emb, din = h->d_rgb + h->d_flow, M = B * N, dh = E / h->heads, mlp = h->mlp;
  const int causal = (flags & 1) ? 1 : 0;
  const int Mp = w.Mp;
  const unsigned dthr = drop_thresh_of(h);
  const float dsc = drop_scale_of(h);
  float* dx = (float*)(ws + w.dx);
  float* tmp = (float*)(ws + w.tmp);
  float* part = (float*)(ws + w.part);
  float* vec = (float*)(ws + w.vec);
  char* T1 = ws + w.T1; char* T2 = ws + w.T2; char* WT = ws + w.WT;
  // gradient tensors in state_dict order (prego_vit_set_weights)
  float* g_enc_w = grads[0]; float* g_enc_b = grads[1]; float* g_cls = grads[2]; float* g_pe = grads[3];
  float* const* gl = grads + 4;
  float* const* gt = grads + 4 + 11 * h->layers;          // pre_head_ln.weight/bias, mlp_head.weight/bias

  // ---- head + final LayerNorm (token 0 only, ViT.py:134-138)
  HIPCHK(hipMemsetAsync(dx, 0, (size_t)M * E * 4, s));
  launch_vit_head_bwd((const float*)(ws + w.x), dlogits, B, N, E, h->ncls, h->lnf_w, h->lnf_b, h->head_w, dx, (float*)(ws + w.head),
                      gt[0], gt[1], gt[2], gt[3], s);

  for (int li = h->layers - 1; li >= 0; --li) {
    const VitLayer& l = h->L[li];
    const VitLayerKeep& k = w.L[li];
    float* const* g = gl + 11 * li;      // ln1 w,b | qkv w | proj w,b | ln2 w,b | ff1 w,b | ff2 w,b
    // ---- FFN: x += drop(W2 drop(gelu(W1 LN2(x) + b1)) + b2)   (Transformer.py:35-47)
    // gradient entering a branch = dx through that branch's output dropout (same stateless mask as the forward)
    const float* dbr = dx;
    if (dthr) { launch_mask_convert(dx, (size_t)M * E, (float*)(ws + w.dxm), ws + w.dxb, dthr, dsc, site_seed(h, li, 3), s); dbr = (const float*)(ws + w.dxm); }
    else launch_f32_to_bf16(dx, ws + w.dxb, (size_t)M * E, s);
    (void)dbr;                                                                              // the bf16 copy w.dxb carries the same values
    if (wgrad(ws + w.dxb, E, ws + k.f, mlp, M, Mp, g[9], g[10], s) ||                        // d W2 [E, mlp], d b2
        dgrad(ws + w.dxb, E, l.ff2_w, mlp, M, tmp, s)) return prego_fail_(PREGO_EINVAL, "backward: FFN-2 GEMM shape");   // d f = dx . W2
    launch_gelu_bwd(tmp, (const float*)(ws + k.u), (size_t)M * mlp, (float*)(ws + w.du), ws + w.dub, s, dthr, dsc, site_seed(h, li, 2));
    if (wgrad(ws + w.dub, mlp, ws + k.xn2, E, M, Mp, g[7], g[8], s) ||                       // d W1 [mlp, E], d b1
        dgrad(ws + w.dub, mlp, l.ff1_w, E, M, tmp, s)) return prego_fail_(PREGO_EINVAL, "backward: FFN-1 GEMM shape");   // d LN2 out = du . W1
    int nb = launch_ln_relu_bwd(tmp, (const float*)(ws + k.x_mid), (const float*)(ws + k.st2), l.ln2_w, l.ln2_b, M, E, 0.f, 0, 0, dx,
                                part, s, 0, 1);                                              // dx += LN2 backward
    launch_colsum_stage2(part, nb, 2 * E, vec, s);
    HIPCHK(hipMemcpyAsync(g[5], vec, (size_t)E * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(g[6], vec + E, (size_t)E * 4, hipMemcpyDeviceToDevice, s));
    // ---- attention: x += drop(proj(attn(LN1(x))))   (Attention.py:21-41, Transformer.py:24-32)
    dbr = dx;
    if (dthr || athr_of(h)) {
      launch_mask_convert(dx, (size_t)M * E, (float*)(ws + w.dxm), ws + w.dxb, dthr, dsc, site_seed(h, li, 1), s, athr_of(h), asc_of(h),
                          site_seed(h, li, 5));
      dbr = (const float*)(ws + w.dxm);
    } else launch_f32_to_bf16(dx, ws + w.dxb, (size_t)M * E, s);
    (void)dbr;
    if (wgrad(ws + w.dxb, E, ws + k.ao, E, M, Mp, g[3], g[4], s) ||                          // d Wproj [E, E], d proj bias
        dgrad(ws + w.dxb, E, l.proj_w, E, M, nullptr, s, ws + w.dO))                         // d o = dx . Wp (bf16, [B,N,h*dh])
      return prego_fail_(PREGO_EINVAL, "backward: projection GEMM shape");
    if (launch_attention_bwd(ws + k.q, ws + k.k, ws + k.vn, ws + k.ao, ws + w.dO, (const float*)(ws + k.lse), (float*)(ws + w.delta),
                             ws + w.dqkv, B, N, h->heads, dh, causal, 1.0f / sqrtf((float)dh), s, athr_of(h), asc_of(h), site_seed(h, li, 4)))
      return prego_fail_(PREGO_EINVAL, "attention backward launch failed");
    if (wgrad(ws + w.dqkv, 3 * E, ws + k.xn1, E, M, Mp, g[2], nullptr, s) ||                 // d Wqkv [3E, E] (no bias, Attention.py:16)
        dgrad(ws + w.dqkv, 3 * E, l.qkv_w, E, M, tmp, s)) return prego_fail_(PREGO_EINVAL, "backward: qkv GEMM shape");   // d LN1 out
    nb = launch_ln_relu_bwd(tmp, (const float*)(ws + k.x_in), (const float*)(ws + k.st1), l.ln1_w, l.ln1_b, M, E, 0.f, 0, 0, dx, part,
                            s, 0, 1);                                                        // dx += LN1 backward
    launch_colsum_stage2(part, nb, 2 * E, vec, s);
    HIPCHK(hipMemcpyAsync(g[0], vec, (size_t)E * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(g[1], vec + E, (size_t)E * 4, hipMemcpyDeviceToDevice, s));
  }
  // ---- tokens: positional table, cls token, encoding Linear (ViT.py:125-129)
  float* denc = (float*)(ws + w.denc);
  launch_vit_tokens_bwd(dx, B, T, E, denc, g_pe, g_cls, s, dthr, dsc, site_seed(h, 0, 0));
  launch_f32_to_bf16(denc, T1, (size_t)B * T * E, s);                                        // the wgrad's A operand in bf16 (rows = frames)
  if (wgrad(T1, E, ws + w.xb, din, B * T, w.MTp, g_enc_w, g_enc_b, s)) return prego_fail_(PREGO_EINVAL, "backward: encoding GEMM shape");   // d W_enc [E, din], d b_enc
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

#ifdef PREGO_DEBUG_ABI
// debug / unit test: the attention backward kernels alone (tests/test_gpu_vit_train.py)
extern "C" int prego_debug_attention_bwd(int batch, int len, int heads, int dh, int causal, const void* qs, const void* k, const void* v,
                                         const void* o, const void* dout, const float* lse, void* dqkv, prego_stream_t stream) {
  if (!qs || !k || !v || !o || !dout || !lse || !dqkv || batch <= 0 || len <= 0 || heads <= 0) return prego_fail_(PREGO_EINVAL, "debug attention bwd: bad arguments");
  float* delta = nullptr;
  HIPCHK(hipMalloc((void**)&delta, (size_t)batch * heads * len * 4));
  const int rc = launch_attention_bwd(qs, k, v, o, dout, lse, delta, dqkv, batch, len, heads, dh, causal, 1.0f / sqrtf((float)dh), (hipStream_t)stream);
  (void)hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(delta);
  if (rc) return prego_fail_(PREGO_EINVAL, "debug attention bwd: unsupported head dim %d", dh);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// debug / unit test: the attention forward kernel alone (tests/test_gpu_transformer.py)
extern "C" int prego_debug_attention_fwd(int batch, int n_query, int len, int heads, int dh, int causal, const void* qs, const void* k,
                                         const void* v, void* out, float* lse, prego_stream_t stream) {
  if (!qs || !k || !v || !out || batch <= 0 || len <= 0 || heads <= 0 || n_query <= 0 || n_query > len)
    return prego_fail_(PREGO_EINVAL, "debug attention fwd: bad arguments");
  const int rc = launch_flash_attention_v2(qs, k, v, out, batch, n_query, len, heads, dh, causal ? 1 : 0, (hipStream_t)stream, lse);
  (void)hipStreamSynchronize((hipStream_t)stream);
  if (rc) return prego_fail_(PREGO_EINVAL, "debug attention fwd: unsupported head dim %d", dh);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

#endif  // PREGO_DEBUG_ABI

// ---- AttentionLayer(FullAttention(mask_flag)) of attn.py:139-170,35-57 ---------------------------------------------
// the attention arithmetic shared by the stateless op and the handle: wqkv bf16 [3D][D] (rows q | k | v), bqkv fp32 [3D],
// wob bf16 [D][D]; act = 5 activation buffers of M*D bf16 (x, q, k, v, attention output)
static int attention_layer_run(int batch, int len, int d_model, int heads, int causal, const float* x, const void* wqkv,
                               const float* bqkv, const void* wob, const float* bo, float* out, char* act, hipStream_t s, bool f16 = false,
                               float* lse = nullptr, unsigned drop_thresh = 0, float drop_scale = 1.f, unsigned long long drop_seed = 0) {
  const size_t M = (size_t)batch * len, D = d_model, step = align_up(M * D * 2, 256);
  const int dh = d_model / heads;
  char* xb = act; char* q = act + step; char* k = act + 2 * step; char* vn = act + 3 * step; char* ao = act + 4 * step;
  launch_cat_convert(x, nullptr, (int)M, d_model, 0, xb, s, f16);
  GemmEpi e{}; e.f16 = f16 ? 1 : 0;
  e.mode = EPI_QKV; e.q = q; e.k = k; e.vn = vn; e.n_tok = len; e.heads = heads; e.dh = dh; e.emb = d_model;
  e.q_scale = 1.0f / sqrtf((float)dh);                                   // attn.py:44 scale = 1/sqrt(E)
  launch_gemm_bf16_nt_epi(xb, d_model, wqkv, d_model, bqkv, nullptr, 0, (int)M, 3 * d_model, d_model, e, s);
  if (launch_flash_attention_v2(q, k, vn, ao, batch, len, len, heads, dh, causal ? 1 : 0, s, lse, drop_thresh, drop_scale, drop_seed, f16)) return -1;
  launch_gemm_bf16_nt(ao, d_model, wob, d_model, bo, out, d_model, (int)M, d_model, d_model, s, f16);
  return 0;
}
static int attention_layer_check(int d_model, int heads) {
  if (d_model % 128 || heads <= 0 || d_model % heads) return prego_fail_(PREGO_EINVAL, "d_model %d / heads %d", d_model, heads);
  const int dh = d_model / heads;
  if (dh != 64 && dh != 128 && dh != 256) return prego_fail_(PREGO_EINVAL, "head dim %d: supported 64, 128, 256", dh);
  return 0;
}

// handle form: the four projection weights are converted to bf16 ONCE (set_weights), not on every call
struct prego_attn_layer {
  int d_model, heads;
  void* wqkv = nullptr; float* bqkv = nullptr; void* wo = nullptr; float* bo = nullptr;
  bool have_weights = false;
  bool f16 = false;                // IEEE fp16 operands instead of bf16 (prego_attention_layer_set_compute_dtype)
  bool f32 = false;                // fp32 operands (parity mode): wqkv32 [3D][D], wo32 [D][D]
  float* wqkv32 = nullptr; float* wo32 = nullptr;
  // FullAttention's nn.Dropout(attention_dropout) on A = softmax(scale * scores) (attn.py:39,54), training entry points only: a stateless hash
  // mask of (seed, element of A), the same in forward_train and backward (prego_attention_layer_set_dropout)
  float attn_drop_p = 0.f; unsigned long long drop_seed = 0;
};
extern "C" int prego_attention_layer_set_dropout(prego_attn_layer* h, float p, uint64_t seed) {
  if (!h) return prego_fail_(PREGO_EINVAL, "handle is NULL");
  if (!(p >= 0.f && p < 1.f)) return prego_fail_(PREGO_EINVAL, "attention_dropout p = %f", (double)p);
  h->attn_drop_p = p;
  h->drop_seed = seed;
  return PREGO_OK;
}
static inline unsigned al_thr(const prego_attn_layer* h) { return h->attn_drop_p > 0.f ? (unsigned)((double)h->attn_drop_p * 4294967296.0) : 0u; }
static inline float al_scale(const prego_attn_layer* h) { return h->attn_drop_p > 0.f ? 1.f / (1.f - h->attn_drop_p) : 1.f; }
extern "C" int prego_attention_layer_set_compute_dtype(prego_attn_layer* h, int compute_dtype) {
  if (!h) return prego_fail_(PREGO_EINVAL, "handle is NULL");
  if (compute_dtype != PREGO_BF16 && compute_dtype != PREGO_F16 && compute_dtype != PREGO_F32)
    return prego_fail_(PREGO_EINVAL, "AttentionLayer compute_dtype %d: PREGO_BF16, PREGO_F16 or PREGO_F32", compute_dtype);
  if ((compute_dtype == PREGO_F16) != h->f16 || (compute_dtype == PREGO_F32) != h->f32) h->have_weights = false;
  h->f16 = compute_dtype == PREGO_F16;
  h->f32 = compute_dtype == PREGO_F32;
  if (h->f32 && !h->wqkv32) {
    const size_t D = h->d_model;
    hipError_t e = hipMalloc((void**)&h->wqkv32, 3 * D * D * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&h->wo32, D * D * 4);
    if (e != hipSuccess) { h->f32 = false; return prego_fail_(PREGO_EHIP, "hipMalloc: %s", hipGetErrorString(e)); }
  }
  return PREGO_OK;
}
extern "C" int prego_attention_layer_create(prego_attn_layer** out, int d_model, int heads) {
  if (!out) return prego_fail_(PREGO_EINVAL, "out is NULL");
  *out = nullptr;
  if (attention_layer_check(d_model, heads)) return PREGO_EINVAL;
  prego_attn_layer* h = new prego_attn_layer();
  h->d_model = d_model; h->heads = heads;
  const size_t D = d_model;
  hipError_t e = hipMalloc(&h->wqkv, 3 * D * D * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&h->bqkv, 3 * D * 4);
  if (e == hipSuccess) e = hipMalloc(&h->wo, D * D * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&h->bo, D * 4);
  if (e != hipSuccess) { prego_attention_layer_destroy(h); return prego_fail_(PREGO_EHIP, "hipMalloc: %s", hipGetErrorString(e)); }
  *out = h;
  return PREGO_OK;
}
extern "C" void prego_attention_layer_destroy(prego_attn_layer* h) {
  if (!h) return;
  for (void* p : {h->wqkv, (void*)h->bqkv, h->wo, (void*)h->bo, (void*)h->wqkv32, (void*)h->wo32}) if (p) (void)hipFree(p);
  delete h;
}
extern "C" int prego_attention_layer_set_weights(prego_attn_layer* h, const float* wq, const float* bq, const float* wk,
                                                 const float* bk, const float* wv, const float* bv, const float* wo,
                                                 const float* bo, prego_stream_t stream) {
  if (!h || !wq || !bq || !wk || !bk || !wv || !bv || !wo || !bo) return prego_fail_(PREGO_EINVAL, "NULL argument");
  hipStream_t s = (hipStream_t)stream;
  const int d = h->d_model;
  const size_t D = d;
  if (h->f32) {
    HIPCHK(hipMemcpyAsync(h->wqkv32, wq, D * D * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(h->wqkv32 + D * D, wk, D * D * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(h->wqkv32 + 2 * D * D, wv, D * D * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(h->wo32, wo, D * D * 4, hipMemcpyDeviceToDevice, s));
  } else {
  launch_pad_convert(true, wq, d, d, d, h->wqkv, d, d, s, h->f16);
  launch_pad_convert(true, wk, d, d, d, (char*)h->wqkv + D * D * 2, d, d, s, h->f16);
  launch_pad_convert(true, wv, d, d, d, (char*)h->wqkv + 2 * D * D * 2, d, d, s, h->f16);
  launch_pad_convert(true, wo, d, d, d, h->wo, d, d, s, h->f16);
  }
  HIPCHK(hipMemcpyAsync(h->bqkv, bq, D * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(h->bqkv + D, bk, D * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(h->bqkv + 2 * D, bv, D * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(h->bo, bo, D * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipGetLastError());
  h->have_weights = true;
  return PREGO_OK;
}
extern "C" size_t prego_attention_layer_handle_workspace_bytes(const prego_attn_layer* h, int batch, int len) {
  if (!h) return 0;
  if (h->f32) return 4 * align_up((size_t)batch * len * h->d_model * 4, 256);       // q | k | v rows [M, 3D] and the attention output
  return 5 * align_up((size_t)batch * len * h->d_model * 2, 256);
}
extern "C" int prego_attention_layer_handle_forward(prego_attn_layer* h, int batch, int len, int causal, const float* x, float* out,
                                                    void* workspace, size_t workspace_bytes, prego_stream_t stream) {
  if (!h || !x || !out || !workspace) return prego_fail_(PREGO_EINVAL, "NULL argument");
  if (!h->have_weights) return prego_fail_(PREGO_EINVAL, "forward before set_weights");
  if (batch <= 0 || len <= 0) return prego_fail_(PREGO_EINVAL, "batch %d, len %d", batch, len);
  if (workspace_bytes < prego_attention_layer_handle_workspace_bytes(h, batch, len)) return prego_fail_(PREGO_EWORKSPACE, "workspace too small");
  if (h->f32) {                 // attn.py:151-170 with fp32 operands: biased projections on the exact-fp32 GEMM, fp32 attention
    hipStream_t s = (hipStream_t)stream;
    const int D = h->d_model, M = batch * len, dh = D / h->heads;
    float* qkv = (float*)workspace;
    float* ao = qkv + (size_t)3 * align_up((size_t)M * D * 4, 256) / 4;
    launch_gemm_f32_nt(x, D, h->wqkv32, D, h->bqkv, qkv, 3 * D, M, 3 * D, D, s);
    if (launch_attention_f32(qkv, 3 * D, 0, D, 2 * D, ao, batch, len, len, h->heads, dh, causal ? 1 : 0, 1.0f / sqrtf((float)dh), s))
      return prego_fail_(PREGO_EINVAL, "attention launch failed");
    launch_gemm_f32_nt(ao, D, h->wo32, D, h->bo, out, D, M, D, D, s);
    HIPCHK(hipGetLastError());
    return PREGO_OK;
  }
  if (attention_layer_run(batch, len, h->d_model, h->heads, causal, x, h->wqkv, h->bqkv, h->wo, h->bo, out, (char*)workspace,
                          (hipStream_t)stream, h->f16))
    return prego_fail_(PREGO_EINVAL, "attention launch failed");
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// ---- training: forward that keeps q, k, v, the attention output and the row log-sum-exp, and the backward over them --------
// (attn.py:139-170 under autograd: the projections are nn.Linear, the attention FullAttention.forward attn.py:35-57; attention_dropout
// (attn.py:39,54: nn.Dropout on A, active under module.train()) is the handle's prego_attention_layer_set_dropout state, p = 0 by default)
struct AttnTrainWs { size_t act, lse, dyb, dO, dqkv, delta, T1, T2, WT, part, dW, vec, total; int Mp; };
static AttnTrainWs attn_train_ws(int d_model, int heads, int batch, int len) {
  const size_t M = (size_t)batch * len, D = d_model;
  AttnTrainWs w{};
  w.Mp = (int)align_up(M, 64);
  size_t off = 0;
  auto put = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  w.act = put(5 * align_up(M * D * 2, 256)); w.lse = put((size_t)batch * heads * len * 4);
  w.dyb = put(M * D * 2); w.dO = put(M * D * 2); w.dqkv = put(M * 3 * D * 2); w.delta = put((size_t)batch * heads * len * 4);
  w.T1 = put(3 * D * (size_t)w.Mp * 2); w.T2 = put(3 * D * (size_t)w.Mp * 2); w.WT = put(3 * D * D * 2);
  w.part = put((M / 64 + 2) * 3 * D * 4); w.dW = put(3 * D * D * 4); w.vec = put(3 * D * 4);
  w.total = off;
  return w;
}
extern "C" size_t prego_attention_layer_train_workspace_bytes(const prego_attn_layer* h, int batch, int len) {
  return (h && batch > 0 && len > 0) ? attn_train_ws(h->d_model, h->heads, batch, len).total : 0;
}
extern "C" int prego_attention_layer_forward_train(prego_attn_layer* h, int batch, int len, int causal, const float* x, float* out,
                                                   void* workspace, size_t workspace_bytes, prego_stream_t stream) {
  if (!h || !x || !out || !workspace) return prego_fail_(PREGO_EINVAL, "NULL argument");
  if (h->f16 || h->f32) return prego_fail_(PREGO_EINVAL, "prego_attention_layer_forward_train on an fp16- / fp32-operand handle: training runs on bf16 handles");
  if (!h->have_weights) return prego_fail_(PREGO_EINVAL, "forward before set_weights");
  if (batch <= 0 || len <= 0) return prego_fail_(PREGO_EINVAL, "batch %d, len %d", batch, len);
  const AttnTrainWs w = attn_train_ws(h->d_model, h->heads, batch, len);
  if (workspace_bytes < w.total) return prego_fail_(PREGO_EWORKSPACE, "training workspace %zu < %zu", workspace_bytes, w.total);
  char* ws = (char*)workspace;
  if (attention_layer_run(batch, len, h->d_model, h->heads, causal, x, h->wqkv, h->bqkv, h->wo, h->bo, out, ws + w.act,
                          (hipStream_t)stream, false, (float*)(ws + w.lse), al_thr(h), al_scale(h), h->drop_seed))
    return prego_fail_(PREGO_EINVAL, "attention launch failed");
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}
// grads: 8 fp32 tensors in set_weights order (wq, bq, wk, bk, wv, bv, wo, bo), overwritten; dx [batch*len, d_model] or NULL
extern "C" int prego_attention_layer_backward(prego_attn_layer* h, int batch, int len, int causal, const float* dout, float* dx,
                                              float* const* grads, int n_tensors, void* workspace, size_t workspace_bytes,
                                              prego_stream_t stream) {
  if (!h || !dout || !grads || !workspace) return prego_fail_(PREGO_EINVAL, "NULL argument");
  if (h->f16 || h->f32) return prego_fail_(PREGO_EINVAL, "prego_attention_layer_backward on an fp16- / fp32-operand handle: training runs on bf16 handles");
  if (n_tensors != 8) return prego_fail_(PREGO_EINVAL, "expected 8 gradient tensors, got %d", n_tensors);
  for (int i = 0; i < 8; ++i) if (!grads[i]) return prego_fail_(PREGO_EINVAL, "gradient tensor %d is NULL", i);
  if (batch <= 0 || len <= 0) return prego_fail_(PREGO_EINVAL, "batch %d, len %d", batch, len);
  const AttnTrainWs w = attn_train_ws(h->d_model, h->heads, batch, len);
  if (workspace_bytes < w.total) return prego_fail_(PREGO_EWORKSPACE, "training workspace %zu < %zu", workspace_bytes, w.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int D = h->d_model, M = batch * len, dh = D / h->heads, Mp = w.Mp;
  const size_t step = align_up((size_t)M * D * 2, 256), DD = (size_t)D * D;
  char* xb = ws + w.act; char* q = xb + step; char* k = xb + 2 * step; char* vn = xb + 3 * step; char* ao = xb + 4 * step;
  float* part = (float*)(ws + w.part); float* vec = (float*)(ws + w.vec); float* dW = (float*)(ws + w.dW);
  char* T1 = ws + w.T1; char* T2 = ws + w.T2; char* WT = ws + w.WT;
  // ---- out_projection (attn.py:170): d bo, d Wo, d (attention output)
  launch_f32_to_bf16(dout, ws + w.dyb, (size_t)M * D, s);
  if (wgrad(ws + w.dyb, D, ao, D, M, Mp, grads[6], grads[7], s) ||                           // d Wo, d bo
      dgrad(ws + w.dyb, D, h->wo, D, M, nullptr, s, ws + w.dO)) return prego_fail_(PREGO_EINVAL, "backward: out_projection GEMM shape");
  // ---- softmax(scale * Q K^T + mask) V (attn.py:41-52)
  if (launch_attention_bwd(q, k, vn, ao, ws + w.dO, (const float*)(ws + w.lse), (float*)(ws + w.delta), ws + w.dqkv, batch, len,
                           h->heads, dh, causal ? 1 : 0, 1.0f / sqrtf((float)dh), s, al_thr(h), al_scale(h), h->drop_seed))
    return prego_fail_(PREGO_EINVAL, "attention backward launch failed");
  // ---- query / key / value projections (attn.py:160-162): rows of dqkv are [dq | dk | dv]
  if (wgrad(ws + w.dqkv, 3 * D, xb, D, M, Mp, dW, vec, s)) return prego_fail_(PREGO_EINVAL, "backward: qkv GEMM shape");
  for (int j = 0; j < 3; ++j) {
    HIPCHK(hipMemcpyAsync(grads[2 * j], dW + j * DD, DD * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(grads[2 * j + 1], vec + (size_t)j * D, (size_t)D * 4, hipMemcpyDeviceToDevice, s));
  }
  if (dx) {     // self-attention: queries = keys = values = x, the three input gradients add up = dqkv . Wqkv
    if (dgrad(ws + w.dqkv, 3 * D, h->wqkv, D, M, dx, s)) return prego_fail_(PREGO_EINVAL, "backward: qkv dgrad shape");
  }
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// stateless op (weights passed per call and converted into the workspace first)
extern "C" size_t prego_attention_layer_workspace_bytes(int batch, int len, int d_model) {
  const size_t M = (size_t)batch * len, D = d_model;
  return align_up(3 * D * D * 2, 256) + align_up(3 * D * 4, 256) + align_up(D * D * 2, 256) + 5 * align_up(M * D * 2, 256);
}

extern "C" int prego_attention_layer_forward(int batch, int len, int d_model, int heads, int causal, const float* x,
                                             const float* wq, const float* bq, const float* wk, const float* bk,
                                             const float* wv, const float* bv, const float* wo, const float* bo, float* out,
                                             void* workspace, size_t workspace_bytes, prego_stream_t stream) {
  if (!x || !wq || !bq || !wk || !bk || !wv || !bv || !wo || !bo || !out || !workspace) return prego_fail_(PREGO_EINVAL, "NULL argument");
  if (attention_layer_check(d_model, heads)) return PREGO_EINVAL;
  if (workspace_bytes < prego_attention_layer_workspace_bytes(batch, len, d_model)) return prego_fail_(PREGO_EWORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t D = d_model;
  char* p = (char*)workspace;
  auto carve = [&](size_t bytes) { char* q = p; p += align_up(bytes, 256); return q; };
  char* wqkv = carve(3 * D * D * 2); float* bqkv = (float*)carve(3 * D * 4); char* wob = carve(D * D * 2);
  launch_pad_convert(true, wq, d_model, d_model, d_model, wqkv, d_model, d_model, s);
  launch_pad_convert(true, wk, d_model, d_model, d_model, wqkv + D * D * 2, d_model, d_model, s);
  launch_pad_convert(true, wv, d_model, d_model, d_model, wqkv + 2 * D * D * 2, d_model, d_model, s);
  launch_pad_convert(true, wo, d_model, d_model, d_model, wob, d_model, d_model, s);
  HIPCHK(hipMemcpyAsync(bqkv, bq, D * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(bqkv + D, bk, D * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(bqkv + 2 * D, bv, D * 4, hipMemcpyDeviceToDevice, s));
  if (attention_layer_run(batch, len, d_model, heads, causal, x, wqkv, bqkv, wob, bo, out, p, s)) return prego_fail_(PREGO_EINVAL, "attention launch failed");
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}
