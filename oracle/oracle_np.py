"""CPU oracle: numpy restatement of PREGO's step_recognition hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under prego_amd/ may import this file; it is
used by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg as
the *checker*, never as the product path (the product path fails loudly when the
HIP library is missing).

The arithmetic of the reference lives in PyTorch (third-party, unpinned:
/root/reference/requirements.txt:9 `torch>=2.0.0`; the build container has torch
2.10.0+rocm7.0 CPU kernels).  Each function below restates the published
algorithm of the torch module the reference calls and cites the reference call
site.  Parity is pinned by tests/golden/*.npz, which oracle/gen_golden.py
produced by importing the real reference modules from /root/reference in the
build container (the reference has no tests or golden vectors of its own:
SURVEY.md section 4).

All functions take/return numpy arrays; `dt` selects float64 (default, the
"truth") or float32 (to see fp32 rounding).
"""
from __future__ import annotations

import math
import numpy as np

LN_EPS = 1e-5  # nn.LayerNorm default, rnn.py:41 / Transformer.py:17,27 / ViT.py:79


# --------------------------------------------------------------------------- #
# elementary ops
# --------------------------------------------------------------------------- #
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def log_softmax(x, axis=-1):
    m = np.max(x, axis=axis, keepdims=True)
    s = x - m
    return s - np.log(np.sum(np.exp(s), axis=axis, keepdims=True))


def layernorm(x, gamma, beta, eps=LN_EPS):
    """nn.LayerNorm over the last dim: biased variance, eps inside the sqrt."""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def _erf(x):
    from scipy.special import erf          # vectorised; np.vectorize(math.erf) is 100x slower on [B,N,mlp] tensors
    return erf(x)


def gelu_erf(x):
    """nn.GELU() default = exact erf form (Transformer.py:40)."""
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def linear(x, w, b=None):
    y = x @ w.T
    return y if b is None else y + b


# --------------------------------------------------------------------------- #
# path A: MROAD ("MiniROAD"), model/rnn/rnn.py:18-71
# --------------------------------------------------------------------------- #
def gru_step(gi, h, w_hh, b_hh):
    """One nn.GRU cell step (rnn.py:38,61).  Row blocks of W/b are [r; z; n];
    n = tanh(gi_n + r * (W_hn h + b_hn)); h' = (1-z) n + z h."""
    H = h.shape[-1]
    gh = h @ w_hh.T + b_hh
    r = sigmoid(gi[..., :H] + gh[..., :H])
    z = sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = np.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1.0 - z) * n + z * h, (r, z, n, gh)


def miniroad_forward(sd, rgb, flow=None, training=False, h0=None, dt=np.float64, keep=False):
    """MROAD.forward (rnn.py:51-71) with dropout off.

    rgb [B,T,Dr], flow [B,T,Df] or None (= the --no_flow model, or an all-zero
    flow half whose weight columns are simply not multiplied: exact either way
    because 0*w == 0).  Returns dict with 'logits' (probabilities in eval mode,
    raw logits in training mode: rnn.py:66-70) and, if keep, the intermediates.
    """
    p = {k: np.asarray(v, dtype=dt) for k, v in sd.items()}
    x = np.asarray(rgb, dtype=dt)
    w1 = p["layer1.0.weight"]
    if flow is not None:
        x = np.concatenate([x, np.asarray(flow, dtype=dt)], axis=2)      # rnn.py:53
    else:
        w1 = w1[:, : x.shape[2]]
    B, T, _ = x.shape
    H = p["gru.weight_hh_l0"].shape[1]
    y = linear(x, w1, p["layer1.0.bias"])                                # rnn.py:40
    e = np.maximum(layernorm(y, p["layer1.1.weight"], p["layer1.1.bias"]), 0.0)  # :41-42
    gi = linear(e, p["gru.weight_ih_l0"], p["gru.bias_ih_l0"])            # GRU input proj
    n_layers = sum(1 for k in p if k.startswith("gru.weight_hh_l"))       # nn.GRU(.., num_layers) (rnn.py:32,38): gru.*_l0, _l1, ...
    h0a = None if h0 is None else np.asarray(h0, dtype=dt)
    h_last = []
    gi0 = gi
    inp = None
    for l in range(n_layers):
        if l > 0:                                                         # layer l's input is layer l - 1's h_t (no dropout between layers: nn.GRU default)
            gi = linear(inp, p[f"gru.weight_ih_l{l}"], p[f"gru.bias_ih_l{l}"])
        if h0a is None:
            h = np.zeros((B, H), dtype=dt)                                # rnn.py:49,60
        else:
            h = h0a if (n_layers == 1 and h0a.ndim == 2) else h0a[l]
        hs = np.empty((B, T, H), dtype=dt)
        for t in range(T):                                                # rnn.py:61
            h, _ = gru_step(gi[:, t], h, p[f"gru.weight_hh_l{l}"], p[f"gru.bias_hh_l{l}"])
            hs[:, t] = h
        h_last.append(h)
        inp = hs
    logits = linear(np.maximum(hs, 0.0), p["f_classification.0.weight"],
                    p["f_classification.0.bias"])                         # rnn.py:62-64
    out = {"logits": logits if training else softmax(logits)}            # rnn.py:66-70
    if keep:
        out.update(y=y, e=e, gi=gi0, h=hs, raw_logits=logits, h_last=h_last[0] if n_layers == 1 else np.stack(h_last))
    return out


# --------------------------------------------------------------------------- #
# loss: criterions/loss.py:15-34 (OadLoss, "NONUNIFORM")
# --------------------------------------------------------------------------- #
def oad_loss(logits, target, reduction="mean"):
    """end_loss + mlce_loss: last frame only; target L2-normalised per row with
    F.normalize's eps=1e-12 clamp (all-zero padding rows contribute 0)."""
    lg = logits[:, -1, :]
    tg = target[:, -1, :]
    nrm = np.maximum(np.sqrt((tg ** 2).sum(axis=1, keepdims=True)), 1e-12)
    per = np.sum(-(tg / nrm) * log_softmax(lg), axis=1)
    return per.mean() if reduction == "mean" else per.sum()


def oad_loss_grad(logits, target, reduction="mean"):
    """d loss / d logits ([B,T,C], non-zero only at t = T-1); reduction 'mean' or 'sum' (loss.py:30-33)."""
    B = logits.shape[0] if reduction == "mean" else 1
    lg = logits[:, -1, :]
    tg = target[:, -1, :]
    nrm = np.maximum(np.sqrt((tg ** 2).sum(axis=1, keepdims=True)), 1e-12)
    y = tg / nrm
    g = np.zeros_like(logits)
    g[:, -1, :] = (softmax(lg) * y.sum(axis=1, keepdims=True) - y) / B
    return g


def miniroad_loss_and_grads(sd, rgb, flow, target, dt=np.float64):
    """Forward (training mode, dropout=0) + OadLoss + full BPTT, by hand.
    Returns (loss, grads dict keyed like the state_dict).  Follows train.py:20-23
    (fwd, loss, backward) for the MROAD graph of rnn.py:51-71."""
    p = {k: np.asarray(v, dtype=dt) for k, v in sd.items()}
    out = miniroad_forward(sd, rgb, flow, training=True, dt=dt, keep=True)
    x = np.asarray(rgb, dtype=dt)
    if flow is not None:
        x = np.concatenate([x, np.asarray(flow, dtype=dt)], axis=2)
    B, T, _ = x.shape
    H = p["gru.weight_hh_l0"].shape[1]
    tgt = np.asarray(target, dtype=dt)
    loss = oad_loss(out["raw_logits"], tgt)
    dlog = oad_loss_grad(out["raw_logits"], tgt)
    g = {}
    hs, gi, e, y = out["h"], out["gi"], out["e"], out["y"]
    hr = np.maximum(hs, 0.0)
    g["f_classification.0.weight"] = np.einsum("btc,bth->ch", dlog, hr)
    g["f_classification.0.bias"] = dlog.sum(axis=(0, 1))
    dhs = (dlog @ p["f_classification.0.weight"]) * (hs > 0)
    w_hh, b_hh = p["gru.weight_hh_l0"], p["gru.bias_hh_l0"]
    dgi = np.zeros_like(gi)
    dw_hh = np.zeros_like(w_hh)
    db_hh = np.zeros_like(b_hh)
    dh = np.zeros((B, H), dtype=dt)
    for t in range(T - 1, -1, -1):
        hprev = hs[:, t - 1] if t > 0 else np.zeros((B, H), dtype=dt)
        _, (r, z, n, gh) = gru_step(gi[:, t], hprev, w_hh, b_hh)
        dh = dh + dhs[:, t]
        dn = dh * (1.0 - z)
        dz = dh * (hprev - n)
        dpre_n = dn * (1.0 - n * n)
        dr = dpre_n * gh[:, 2 * H:]
        dpre_r = dr * r * (1.0 - r)
        dpre_z = dz * z * (1.0 - z)
        dgi[:, t] = np.concatenate([dpre_r, dpre_z, dpre_n], axis=1)
        dgh = np.concatenate([dpre_r, dpre_z, dpre_n * r], axis=1)
        dw_hh += dgh.T @ hprev
        db_hh += dgh.sum(axis=0)
        dh = dh * z + dgh @ w_hh
    g["gru.weight_hh_l0"] = dw_hh
    g["gru.bias_hh_l0"] = db_hh
    g["gru.weight_ih_l0"] = np.einsum("btg,bte->ge", dgi, e)
    g["gru.bias_ih_l0"] = dgi.sum(axis=(0, 1))
    de = (dgi @ p["gru.weight_ih_l0"]) * (e > 0)
    gam = p["layer1.1.weight"]
    mu = y.mean(axis=-1, keepdims=True)
    var = ((y - mu) ** 2).mean(axis=-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + LN_EPS)
    xh = (y - mu) * rstd
    g["layer1.1.weight"] = (de * xh).sum(axis=(0, 1))
    g["layer1.1.bias"] = de.sum(axis=(0, 1))
    dxh = de * gam
    dy = rstd * (dxh - dxh.mean(axis=-1, keepdims=True) - xh * (dxh * xh).mean(axis=-1, keepdims=True))
    w1 = p["layer1.0.weight"]
    g["layer1.0.weight"] = np.einsum("bte,btd->ed", dy, x)
    if g["layer1.0.weight"].shape != w1.shape:      # flow=None: zero-flow columns get zero grad
        full = np.zeros_like(w1)
        full[:, : x.shape[2]] = g["layer1.0.weight"]
        g["layer1.0.weight"] = full
    g["layer1.0.bias"] = dy.sum(axis=(0, 1))
    return loss, g


def adamw_step(param, grad, m, v, step, lr=1e-4, wd=0.05, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW (main.py:62-67: lr 1e-4, weight_decay 0.05, default
    betas/eps, amsgrad off).  `step` is 1-based.  Returns (param, m, v)."""
    param = param * (1.0 - lr * wd)
    m = b1 * m + (1.0 - b1) * grad
    v = b2 * v + (1.0 - b2) * grad * grad
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    return param - (lr / bc1) * (m / denom), m, v


# --------------------------------------------------------------------------- #
# path B: ViTEnc ("Transformer") and the causal attention of attn.py
# --------------------------------------------------------------------------- #
def self_attention(x, qkv_w, proj_w, proj_b, heads, causal=False):
    """SelfAttention.forward (Attention.py:21-41): fused bias-free qkv, split as
    reshape(B,N,3,h,dh); scale dh^-0.5; softmax; AV; proj.  causal=True is NOT in
    the reference module (it has no mask) - it is our extension for config 4,
    checked against attn.py's FullAttention below."""
    B, N, C = x.shape
    dh = C // heads
    qkv = linear(x, qkv_w).reshape(B, N, 3, heads, dh).transpose(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = np.einsum("bhid,bhjd->bhij", q, k) * (dh ** -0.5)
    if causal:
        s = np.where(np.triu(np.ones((N, N), dtype=bool), 1), -np.inf, s)
    a = softmax(s)
    o = np.einsum("bhij,bhjd->bhid", a, v).transpose(0, 2, 1, 3).reshape(B, N, C)
    return linear(o, proj_w, proj_b)


def causal_attention_layer(x, wq, bq, wk, bk, wv, bv, wo, bo, heads, mask_flag=True):
    """AttentionLayer(FullAttention(mask_flag)) (attn.py:139-170, 35-57, 10-18):
    biased Q/K/V/O projections, scores einsum(blhe,bshe->bhls), masked_fill(triu(1), -inf),
    softmax(scale * scores) with scale = 1/sqrt(E), einsum(bhls,bshd->blhd)."""
    B, L, D = x.shape
    E = D // heads
    q = linear(x, wq, bq).reshape(B, L, heads, E)
    k = linear(x, wk, bk).reshape(B, L, heads, E)
    v = linear(x, wv, bv).reshape(B, L, heads, E)
    s = np.einsum("blhe,bshe->bhls", q, k)
    if mask_flag:
        s = np.where(np.triu(np.ones((L, L), dtype=bool), 1), -np.inf, s)
    a = softmax(s / math.sqrt(E))
    o = np.einsum("bhls,bshd->blhd", a, v).reshape(B, L, D)
    return linear(o, wo, bo)


def causal_attention_layer_grads(x, wq, bq, wk, bk, wv, bv, wo, bo, heads, dout, mask_flag=True):
    """d/d(x, parameters) of sum(causal_attention_layer(...) * dout): what autograd computes through attn.py:151-170
    (nn.Linear projections, attn.py:41-52 scores / mask / softmax / A.V).  Returns (dx, [dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo])."""
    B, L, D = x.shape
    E = D // heads
    x2 = x.reshape(B * L, D)
    q = linear(x, wq, bq).reshape(B, L, heads, E)
    k = linear(x, wk, bk).reshape(B, L, heads, E)
    v = linear(x, wv, bv).reshape(B, L, heads, E)
    s = np.einsum("blhe,bshe->bhls", q, k)
    if mask_flag:
        s = np.where(np.triu(np.ones((L, L), dtype=bool), 1), -np.inf, s)
    sc = 1.0 / math.sqrt(E)
    a = softmax(s * sc)
    o = np.einsum("bhls,bshd->blhd", a, v).reshape(B * L, D)
    dy = dout.reshape(B * L, D)
    dwo, dbo = dy.T @ o, dy.sum(0)
    do = (dy @ wo).reshape(B, L, heads, E)
    dv = np.einsum("bhls,blhd->bshd", a, do)
    da = np.einsum("blhd,bshd->bhls", do, v)
    ds = a * (da - (da * a).sum(-1, keepdims=True)) * sc          # softmax backward; masked entries have a = 0
    dq = np.einsum("bhls,bshe->blhe", ds, k).reshape(B * L, D)
    dk = np.einsum("bhls,blhe->bshe", ds, q).reshape(B * L, D)
    dv = dv.reshape(B * L, D)
    dx = (dq @ wq + dk @ wk + dv @ wv).reshape(B, L, D)
    return dx, [dq.T @ x2, dq.sum(0), dk.T @ x2, dk.sum(0), dv.T @ x2, dv.sum(0), dwo, dbo]


def feedforward(x, w1, b1, w2, b2):
    """FeedForward (Transformer.py:35-47), dropout off."""
    return linear(gelu_erf(linear(x, w1, b1)), w2, b2)


def vit_forward(sd, rgb, flow, heads, num_layers=1, dt=np.float64, causal=False, keep=False):
    """ViTEnc.forward (ViT.py:117-143) with TransformerModel (Transformer.py:50-82):
    encode, cat cls token at the END (ViT.py:128), + learned pos-emb, pre-norm
    blocks, final LN, select token 0 (ViT.py:136), linear head.  Output [B,1,C]
    raw logits in both modes."""
    p = {k: np.asarray(v, dtype=dt) for k, v in sd.items() if k != "position_encoding.position_ids"}
    x = np.asarray(rgb, dtype=dt)
    if flow is not None:
        x = np.concatenate([x, np.asarray(flow, dtype=dt)], axis=2)
    x = linear(x, p["linear_encoding.weight"], p["linear_encoding.bias"])
    B = x.shape[0]
    x = np.concatenate([x, np.broadcast_to(p["cls_token"], (B, 1, x.shape[2]))], axis=1)
    x = x + p["position_encoding.pe.weight"][None, : x.shape[1]]
    inter = {}
    for l in range(num_layers):
        a, f = 2 * l, 2 * l + 1
        xn = layernorm(x, p[f"encoder.net.{a}.fn.norm.weight"], p[f"encoder.net.{a}.fn.norm.bias"])
        att = self_attention(xn, p[f"encoder.net.{a}.fn.fn.qkv.weight"],
                             p[f"encoder.net.{a}.fn.fn.proj.weight"],
                             p[f"encoder.net.{a}.fn.fn.proj.bias"], heads, causal=causal)
        x = x + att
        xn2 = layernorm(x, p[f"encoder.net.{f}.fn.norm.weight"], p[f"encoder.net.{f}.fn.norm.bias"])
        ff = feedforward(xn2, p[f"encoder.net.{f}.fn.fn.net.0.weight"], p[f"encoder.net.{f}.fn.fn.net.0.bias"],
                         p[f"encoder.net.{f}.fn.fn.net.3.weight"], p[f"encoder.net.{f}.fn.fn.net.3.bias"])
        if keep and l == 0:
            inter.update(ln1=xn, attn=att, ln2=xn2, ffn=ff)
        x = x + ff
    x = layernorm(x, p["pre_head_ln.weight"], p["pre_head_ln.bias"])
    logits = linear(x[:, 0], p["mlp_head.weight"], p["mlp_head.bias"])[:, None, :]
    out = {"logits": logits}
    out.update(inter)
    return out


def _ln_bwd(dy, x, gamma):
    """backward of layernorm(x, gamma, beta): returns (dx, dgamma, dbeta)"""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + LN_EPS)
    xh = (x - mu) * rstd
    red = tuple(range(x.ndim - 1))
    dg = (dy * xh).sum(axis=red)
    db = dy.sum(axis=red)
    dxh = dy * gamma
    dx = rstd * (dxh - dxh.mean(axis=-1, keepdims=True) - xh * (dxh * xh).mean(axis=-1, keepdims=True))
    return dx, dg, db


def hash_dropout_mask(seed, n, p):
    """The build's stateless dropout mask (csrc/common.h: dropout_keep_): keep element i iff
    mix32(seed * 0x9E3779B97F4A7C15 + i) >= p * 2^32, kept values scaled by 1 / (1 - p).  Returned as a float64 multiplier [n].
    (The reference's nn.Dropout draws from torch's RNG stream, which cannot be matched; what can be checked is that forward and
    backward of the HIP path use ONE mask and that the arithmetic around it is the reference's.)"""
    if p <= 0:
        return np.ones(n)
    with np.errstate(over="ignore"):
        x = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.arange(n, dtype=np.uint64)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xFF51AFD7ED558CCD)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xC4CEB9FE1A85EC53)
        x ^= x >> np.uint64(33)
    thresh = np.uint64(int(p * 4294967296.0))
    keep = (x & np.uint64(0xFFFFFFFF)) >= thresh
    return keep.astype(np.float64) / (1.0 - p)


def vit_loss_and_grads(sd, rgb, flow, target, heads, num_layers=1, dt=np.float64, causal=False, masks=None):
    """Training step of the `Transformer` registry entry by hand: ViTEnc.forward (ViT.py:117-143, dropouts 0), OadLoss on the
    [B,1,C] logits (loss.py:15-34: logits[:, -1] is the only row, target[:, -1] the last frame's label), and the full backward
    through the head, the final LayerNorm (token 0 only), every pre-norm block (Transformer.py:60-77: FFN with exact-erf GELU,
    SelfAttention of Attention.py:21-41), the learned positional table, the cls token (appended at the END, ViT.py:128) and the
    encoding Linear.  `masks` (optional): multiplicative dropout masks at the reference's nn.Dropout sites outside the attention
    module: masks["pe"] [B,N,E] (ViT.py:130), masks[(l, "attn")] [B,N,E] (PreNormDrop, Transformer.py:31), masks[(l, "gelu")]
    [B,N,mlp] and masks[(l, "ffn")] [B,N,E] (FeedForward, Transformer.py:41,46); inside it: masks[(l, "prob")] [B,h,N,N]
    (attn_drop, Attention.py:36) and masks[(l, "proj")] [B,N,E] (proj_drop, Attention.py:40).
    Returns (loss, logits [B,1,C], grads keyed like the state_dict)."""
    masks = masks or {}
    one = 1.0
    p = {k: np.asarray(v, dtype=dt) for k, v in sd.items() if k != "position_encoding.position_ids"}
    X = np.asarray(rgb, dtype=dt)
    if flow is not None:
        X = np.concatenate([X, np.asarray(flow, dtype=dt)], axis=2)
    B, T, _ = X.shape
    E = p["linear_encoding.weight"].shape[0]
    N, dh = T + 1, E // heads
    scale = dh ** -0.5
    x = linear(X, p["linear_encoding.weight"], p["linear_encoding.bias"])
    x = np.concatenate([x, np.broadcast_to(p["cls_token"].reshape(1, 1, E), (B, 1, E))], axis=1)
    x = (x + p["position_encoding.pe.weight"][None, :N]) * masks.get("pe", one)
    cache = []
    for l in range(num_layers):
        a_, f_ = 2 * l, 2 * l + 1
        c = {"x_in": x}
        xn = layernorm(x, p[f"encoder.net.{a_}.fn.norm.weight"], p[f"encoder.net.{a_}.fn.norm.bias"])
        qkv = linear(xn, p[f"encoder.net.{a_}.fn.fn.qkv.weight"]).reshape(B, N, 3, heads, dh).transpose(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        sc = np.einsum("bhid,bhjd->bhij", q, k) * scale
        if causal:
            sc = np.where(np.triu(np.ones((N, N), dtype=bool), 1), -np.inf, sc)
        att = softmax(sc)
        att_d = att * masks.get((l, "prob"), one)                 # attn_drop on the probabilities (Attention.py:36)
        o = np.einsum("bhij,bhjd->bhid", att_d, v).transpose(0, 2, 1, 3).reshape(B, N, E)
        x = x + (linear(o, p[f"encoder.net.{a_}.fn.fn.proj.weight"], p[f"encoder.net.{a_}.fn.fn.proj.bias"])
                 * masks.get((l, "proj"), one) * masks.get((l, "attn"), one))       # proj_drop, then PreNormDrop
        c.update(xn=xn, q=q, k=k, v=v, att=att, att_d=att_d, o=o, x_mid=x)
        xn2 = layernorm(x, p[f"encoder.net.{f_}.fn.norm.weight"], p[f"encoder.net.{f_}.fn.norm.bias"])
        u = linear(xn2, p[f"encoder.net.{f_}.fn.fn.net.0.weight"], p[f"encoder.net.{f_}.fn.fn.net.0.bias"])
        f = gelu_erf(u) * masks.get((l, "gelu"), one)
        x = x + linear(f, p[f"encoder.net.{f_}.fn.fn.net.3.weight"], p[f"encoder.net.{f_}.fn.fn.net.3.bias"]) * masks.get((l, "ffn"), one)
        c.update(xn2=xn2, u=u, f=f)
        cache.append(c)
    xf0 = layernorm(x[:, 0], p["pre_head_ln.weight"], p["pre_head_ln.bias"])
    logits = linear(xf0, p["mlp_head.weight"], p["mlp_head.bias"])[:, None, :]
    tgt = np.asarray(target, dtype=dt)
    loss = oad_loss(logits, tgt)
    dlog = oad_loss_grad(logits, tgt)[:, 0, :]
    g = {}
    g["mlp_head.weight"] = dlog.T @ xf0
    g["mlp_head.bias"] = dlog.sum(axis=0)
    d0, g["pre_head_ln.weight"], g["pre_head_ln.bias"] = _ln_bwd(dlog @ p["mlp_head.weight"], x[:, 0], p["pre_head_ln.weight"])
    dx = np.zeros_like(x)
    dx[:, 0] = d0
    for l in range(num_layers - 1, -1, -1):
        a_, f_ = 2 * l, 2 * l + 1
        c = cache[l]
        w2, w1 = p[f"encoder.net.{f_}.fn.fn.net.3.weight"], p[f"encoder.net.{f_}.fn.fn.net.0.weight"]
        dbr = dx * masks.get((l, "ffn"), one)                      # gradient entering the FFN branch through its output dropout
        g[f"encoder.net.{f_}.fn.fn.net.3.weight"] = np.einsum("bne,bnm->em", dbr, c["f"])
        g[f"encoder.net.{f_}.fn.fn.net.3.bias"] = dbr.sum(axis=(0, 1))
        u = c["u"]
        dgelu = 0.5 * (1.0 + _erf(u / math.sqrt(2.0))) + u * np.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi)
        du = (dbr @ w2) * masks.get((l, "gelu"), one) * dgelu
        g[f"encoder.net.{f_}.fn.fn.net.0.weight"] = np.einsum("bnm,bne->me", du, c["xn2"])
        g[f"encoder.net.{f_}.fn.fn.net.0.bias"] = du.sum(axis=(0, 1))
        dxa, g[f"encoder.net.{f_}.fn.norm.weight"], g[f"encoder.net.{f_}.fn.norm.bias"] = _ln_bwd(
            du @ w1, c["x_mid"], p[f"encoder.net.{f_}.fn.norm.weight"])
        dx = dx + dxa
        wp, wq = p[f"encoder.net.{a_}.fn.fn.proj.weight"], p[f"encoder.net.{a_}.fn.fn.qkv.weight"]
        dbr = dx * masks.get((l, "attn"), one) * masks.get((l, "proj"), one)
        g[f"encoder.net.{a_}.fn.fn.proj.weight"] = np.einsum("bne,bnd->ed", dbr, c["o"])
        g[f"encoder.net.{a_}.fn.fn.proj.bias"] = dbr.sum(axis=(0, 1))
        do = (dbr @ wp).reshape(B, N, heads, dh).transpose(0, 2, 1, 3)
        dv = np.einsum("bhij,bhid->bhjd", c["att_d"], do)
        da = np.einsum("bhid,bhjd->bhij", do, c["v"]) * masks.get((l, "prob"), one)
        ds = c["att"] * (da - (da * c["att"]).sum(axis=-1, keepdims=True))
        dq = np.einsum("bhij,bhjd->bhid", ds, c["k"]) * scale
        dk = np.einsum("bhij,bhid->bhjd", ds, c["q"]) * scale
        dqkv = np.stack([dq, dk, dv], axis=0).transpose(1, 3, 0, 2, 4).reshape(B, N, 3 * E)
        g[f"encoder.net.{a_}.fn.fn.qkv.weight"] = np.einsum("bnj,bne->je", dqkv, c["xn"])
        dxa, g[f"encoder.net.{a_}.fn.norm.weight"], g[f"encoder.net.{a_}.fn.norm.bias"] = _ln_bwd(
            dqkv @ wq, c["x_in"], p[f"encoder.net.{a_}.fn.norm.weight"])
        dx = dx + dxa
    dx = dx * masks.get("pe", one)
    g["position_encoding.pe.weight"] = dx.sum(axis=0)
    g["cls_token"] = dx[:, N - 1].sum(axis=0).reshape(1, 1, E)
    g["linear_encoding.weight"] = np.einsum("bte,btd->ed", dx[:, :T], X)
    g["linear_encoding.bias"] = dx[:, :T].sum(axis=(0, 1))
    return loss, logits, g


# --------------------------------------------------------------------------- #
# eval-loop semantics (trainer/eval.py:36-65) and aggregation (utils/aggregate.py:46-90)
# --------------------------------------------------------------------------- #
def eval_argmax(prob, target):
    """Per video: pred = argmax(prob, axis=1); gt = argmax(one-hot target, axis=1)."""
    return np.argmax(prob, axis=1), np.argmax(target, axis=1)


def aggregate(data, window_size=200):
    """utils/aggregate.py:46-90: 200-frame majority vote (np.bincount argmax =
    lowest id on ties), change indices, consecutive de-duplication."""
    def changes(a):
        r = [i for i in range(1, len(a)) if a[i] != a[i - 1]]
        r.append(len(a))
        return r

    def dedup(a):
        r = [a[0]]
        for i in range(1, len(a)):
            if a[i] != a[i - 1]:
                r.append(a[i])
        return r

    out = {}
    for key, value in data.items():
        pred = np.asarray(value["pred"])
        gt = list(value["gt"])
        new = np.zeros_like(pred)
        for s in range(0, len(pred), window_size):
            e = min(s + window_size, len(pred))
            new[s:e] = np.argmax(np.bincount(pred[s:e]))
        out[key] = {
            "pred": [int(v) for v in dedup(list(new))],
            "gt": [int(v) for v in dedup(gt)],
            "changes_pred": changes(list(new)),
            "changes_gt": changes(gt),
        }
    return out
