// C ABI of the "Transformer" (ViTEnc) path and of the causal AttentionLayer op (include/prego_amd.h).
// bf16 MFMA operands, fp32 accumulation / residual stream / LayerNorm / softmax.
#include "../../include/prego_amd.h"
#include "kernels.h"

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
};
struct prego_vit {
  int d_rgb, d_flow, emb, mlp, heads, layers, window, ncls;
  void* enc_w = nullptr; float* enc_b = nullptr; float* cls = nullptr; float* pe = nullptr;
  std::vector<VitLayer> L;
  float *lnf_w = nullptr, *lnf_b = nullptr, *head_w = nullptr, *head_b = nullptr;
  std::vector<void*> allocs;
  bool have_weights = false;
};

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

extern "C" int prego_vit_num_tensors(const prego_vit* h) { return h ? 4 + 11 * h->layers + 4 : 0; }

extern "C" int prego_vit_set_weights(prego_vit* h, const float* const* t, int n_tensors, prego_stream_t stream) {
  if (!h || !t) return prego_fail_(PREGO_EINVAL, "NULL");
  if (n_tensors != prego_vit_num_tensors(h)) return prego_fail_(PREGO_EINVAL, "expected %d tensors, got %d", prego_vit_num_tensors(h), n_tensors);
  for (int i = 0; i < n_tensors; ++i) if (!t[i]) return prego_fail_(PREGO_EINVAL, "tensor %d is NULL", i);
  hipStream_t s = (hipStream_t)stream;
  const int E = h->emb, din = h->d_rgb + h->d_flow, mlp = h->mlp;
  int k = 0;
  auto f32 = [&](float* dst, size_t n) { return hipMemcpyAsync(dst, t[k++], n * 4, hipMemcpyDeviceToDevice, s); };
  launch_pad_convert(true, t[k++], E, din, din, h->enc_w, E, din, s);
  HIPCHK(f32(h->enc_b, E)); HIPCHK(f32(h->cls, E)); HIPCHK(f32(h->pe, (size_t)(h->window + 1) * E));
  for (auto& l : h->L) {
    HIPCHK(f32(l.ln1_w, E)); HIPCHK(f32(l.ln1_b, E));
    launch_pad_convert(true, t[k++], 3 * E, E, E, l.qkv_w, 3 * E, E, s);
    launch_pad_convert(true, t[k++], E, E, E, l.proj_w, E, E, s);
    HIPCHK(f32(l.proj_b, E)); HIPCHK(f32(l.ln2_w, E)); HIPCHK(f32(l.ln2_b, E));
    launch_pad_convert(true, t[k++], mlp, E, E, l.ff1_w, mlp, E, s);
    HIPCHK(f32(l.ff1_b, mlp));
    launch_pad_convert(true, t[k++], E, mlp, mlp, l.ff2_w, E, mlp, s);
    HIPCHK(f32(l.ff2_b, E));
  }
  HIPCHK(f32(h->lnf_w, E)); HIPCHK(f32(h->lnf_b, E)); HIPCHK(f32(h->head_w, (size_t)h->ncls * E)); HIPCHK(f32(h->head_b, h->ncls));
  HIPCHK(hipGetLastError());
  h->have_weights = true;
  return PREGO_OK;
}

struct VitWs { size_t xb, enc, x, xn, q, k, vt, ao, f, total; int npad; };
static VitWs vit_ws(const prego_vit* h, int B) {
  const size_t E = h->emb, T = h->window, N = T + 1, din = h->d_rgb + h->d_flow;
  VitWs w{};
  w.npad = (int)align_up(N, 64);
  size_t off = 0;
  auto put = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  w.xb = put((size_t)B * T * din * 2); w.enc = put((size_t)B * T * E * 4); w.x = put((size_t)B * N * E * 4);
  w.xn = put((size_t)B * N * E * 2); w.q = put((size_t)B * N * E * 2); w.k = put((size_t)B * N * E * 2);
  w.vt = put((size_t)B * E * w.npad * 2); w.ao = put((size_t)B * N * E * 2); w.f = put((size_t)B * N * h->mlp * 2);
  w.total = off;
  return w;
}
extern "C" size_t prego_vit_workspace_bytes(const prego_vit* h, int batch) { return (h && batch > 0) ? vit_ws(h, batch).total : 0; }

// one pre-norm encoder block on the fp32 residual stream x [M = B*N, E] (Transformer.py:60-77)
static int encoder_block(const prego_vit* h, const VitLayer& l, float* x, char* ws, const VitWs& w, int B, int N, int causal,
                         hipStream_t s) {
  const int E = h->emb, M = B * N, dh = E / h->heads;
  launch_ln_relu(true, x, l.ln1_w, l.ln1_b, M, E, 1e-5f, ws + w.xn, nullptr, 0.f, 0, 0, s, 0);
  GemmEpi e{};
  e.mode = EPI_QKV; e.q = ws + w.q; e.k = ws + w.k; e.vt = ws + w.vt; e.n_tok = N; e.n_pad = w.npad; e.heads = h->heads;
  e.dh = dh; e.emb = E; e.q_scale = 1.0f / sqrtf((float)dh);                      // Attention.py:14 (dh^-0.5)
  if (hipMemsetAsync(ws + w.vt, 0, (size_t)B * E * w.npad * 2, s) != hipSuccess) return -1;   // V^T pad must be finite
  launch_gemm_bf16_nt_epi(ws + w.xn, E, l.qkv_w, E, nullptr, nullptr, 0, M, 3 * E, E, e, s);
  if (launch_flash_attention(ws + w.q, ws + w.k, ws + w.vt, ws + w.ao, B, N, w.npad, h->heads, dh, causal, s)) return -1;
  GemmEpi r{}; r.mode = EPI_RESIDUAL;
  launch_gemm_bf16_nt_epi(ws + w.ao, E, l.proj_w, E, l.proj_b, x, E, M, E, E, r, s);          // x += proj(attn)
  launch_ln_relu(true, x, l.ln2_w, l.ln2_b, M, E, 1e-5f, ws + w.xn, nullptr, 0.f, 0, 0, s, 0);
  GemmEpi g{}; g.mode = EPI_GELU_BF16; g.out_b = ws + w.f;
  launch_gemm_bf16_nt_epi(ws + w.xn, E, l.ff1_w, E, l.ff1_b, nullptr, h->mlp, M, h->mlp, E, g, s);   // gelu(W1 x + b1)
  launch_gemm_bf16_nt_epi(ws + w.f, h->mlp, l.ff2_w, h->mlp, l.ff2_b, x, E, M, E, h->mlp, r, s);     // x += W2 . + b2
  return 0;
}

extern "C" int prego_vit_forward(prego_vit* h, int batch, const float* rgb, const float* flow, float* out_logits, int flags,
                                 void* workspace, size_t workspace_bytes, prego_stream_t stream) {
  if (!h || !out_logits || !workspace) return prego_fail_(PREGO_EINVAL, "NULL argument");
  if (!h->have_weights) return prego_fail_(PREGO_EINVAL, "forward before set_weights");
  if (batch <= 0) return prego_fail_(PREGO_EINVAL, "batch %d", batch);
  if ((h->d_rgb > 0 && !rgb) || (h->d_flow > 0 && !flow && h->d_rgb == 0)) return prego_fail_(PREGO_EINVAL, "missing input");
  const VitWs w = vit_ws(h, batch);
  if (workspace_bytes < w.total) return prego_fail_(PREGO_EWORKSPACE, "workspace %zu < %zu", workspace_bytes, w.total);
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int B = batch, T = h->window, N = T + 1, E = h->emb, din = h->d_rgb + h->d_flow;
  launch_cat_convert(rgb, flow, B * T, h->d_rgb, h->d_flow, ws + w.xb, s);
  launch_gemm_bf16_nt(ws + w.xb, din, h->enc_w, din, h->enc_b, (float*)(ws + w.enc), E, B * T, E, din, s);   // ViT.py:125
  launch_vit_tokens((const float*)(ws + w.enc), h->cls, h->pe, B, T, E, (float*)(ws + w.x), s);              // ViT.py:126-129
  for (const auto& l : h->L)
    if (encoder_block(h, l, (float*)(ws + w.x), ws, w, B, N, (flags & 1) ? 1 : 0, s)) return prego_fail_(PREGO_EINVAL, "encoder block launch failed");
  launch_vit_head((const float*)(ws + w.x), B, N, E, h->lnf_w, h->lnf_b, h->head_w, h->head_b, h->ncls, out_logits, s);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}

// ---- AttentionLayer(FullAttention(mask_flag)) of attn.py:139-170,35-57: stateless op -------------------------------
extern "C" size_t prego_attention_layer_workspace_bytes(int batch, int len, int d_model) {
  const size_t M = (size_t)batch * len, D = d_model, npad = align_up((size_t)len, 64);
  return align_up(3 * D * D * 2, 256) + align_up(3 * D * 4, 256) + align_up(D * D * 2, 256) + 4 * align_up(M * D * 2, 256) +
         align_up((size_t)batch * D * npad * 2, 256);
}

extern "C" int prego_attention_layer_forward(int batch, int len, int d_model, int heads, int causal, const float* x,
                                             const float* wq, const float* bq, const float* wk, const float* bk,
                                             const float* wv, const float* bv, const float* wo, const float* bo, float* out,
                                             void* workspace, size_t workspace_bytes, prego_stream_t stream) {
  if (!x || !wq || !bq || !wk || !bk || !wv || !bv || !wo || !bo || !out || !workspace) return prego_fail_(PREGO_EINVAL, "NULL argument");
  if (d_model % 128 || heads <= 0 || d_model % heads) return prego_fail_(PREGO_EINVAL, "d_model %d / heads %d", d_model, heads);
  const int dh = d_model / heads;
  if (dh != 64 && dh != 128 && dh != 256) return prego_fail_(PREGO_EINVAL, "head dim %d: supported 64, 128, 256", dh);
  if (workspace_bytes < prego_attention_layer_workspace_bytes(batch, len, d_model)) return prego_fail_(PREGO_EWORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t M = (size_t)batch * len, D = d_model, npad = align_up((size_t)len, 64);
  char* p = (char*)workspace;
  auto carve = [&](size_t bytes) { char* q = p; p += align_up(bytes, 256); return q; };
  char* wqkv = carve(3 * D * D * 2); float* bqkv = (float*)carve(3 * D * 4); char* wob = carve(D * D * 2);
  char* xb = carve(M * D * 2); char* q = carve(M * D * 2); char* k = carve(M * D * 2); char* ao = carve(M * D * 2);
  char* vt = carve((size_t)batch * D * npad * 2);
  launch_pad_convert(true, wq, d_model, d_model, d_model, wqkv, d_model, d_model, s);
  launch_pad_convert(true, wk, d_model, d_model, d_model, wqkv + D * D * 2, d_model, d_model, s);
  launch_pad_convert(true, wv, d_model, d_model, d_model, wqkv + 2 * D * D * 2, d_model, d_model, s);
  launch_pad_convert(true, wo, d_model, d_model, d_model, wob, d_model, d_model, s);
  HIPCHK(hipMemcpyAsync(bqkv, bq, D * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(bqkv + D, bk, D * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(bqkv + 2 * D, bv, D * 4, hipMemcpyDeviceToDevice, s));
  launch_cat_convert(x, nullptr, (int)M, d_model, 0, xb, s);
  HIPCHK(hipMemsetAsync(vt, 0, (size_t)batch * D * npad * 2, s));
  GemmEpi e{};
  e.mode = EPI_QKV; e.q = q; e.k = k; e.vt = vt; e.n_tok = len; e.n_pad = (int)npad; e.heads = heads; e.dh = dh; e.emb = d_model;
  e.q_scale = 1.0f / sqrtf((float)dh);                                   // attn.py:44 scale = 1/sqrt(E)
  launch_gemm_bf16_nt_epi(xb, d_model, wqkv, d_model, bqkv, nullptr, 0, (int)M, 3 * d_model, d_model, e, s);
  if (launch_flash_attention(q, k, vt, ao, batch, len, (int)npad, heads, dh, causal ? 1 : 0, s)) return prego_fail_(PREGO_EINVAL, "attention launch failed");
  launch_gemm_bf16_nt(ao, d_model, wob, d_model, bo, out, d_model, (int)M, d_model, d_model, s);
  HIPCHK(hipGetLastError());
  return PREGO_OK;
}
