"""Build-owned deterministic weight / input generator.

The reference ships no checkpoint and no features, so every
parity fixture, GPU test and bench run regenerates identical tensors from a seed
instead of committing 72 MB of weights.  The generator is counter based
(splitmix64 -> uniform), so any tensor can be produced independently, on any
box, with numpy only.

Initialisation bounds follow what the reference relies on implicitly, i.e. the
PyTorch defaults of the modules it instantiates:
  nn.Linear  (model/rnn/rnn.py:40,46)  weight,bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in))
  nn.GRU     (model/rnn/rnn.py:38)     all four tensors ~ U(-1/sqrt(H), 1/sqrt(H))
  nn.LayerNorm (model/rnn/rnn.py:41)   weight=1, bias=0  (we jitter them so that a
                                       kernel that ignores gamma/beta cannot pass)
"""
from __future__ import annotations

import zlib
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _splitmix64(idx: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 output for counters `idx` (uint64 array) on stream `seed`."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (idx + np.uint64(1)) * _GOLD) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def stream_seed(seed: int, name: str) -> int:
    """Independent 64-bit stream id for a named tensor."""
    h = zlib.crc32(name.encode()) & 0xFFFFFFFF
    return int(_splitmix64(np.array([h], dtype=np.uint64), seed)[0])


def uniform01(shape, seed: int, name: str) -> np.ndarray:
    """float64 uniforms in [0,1), 53-bit, deterministic in (seed, name, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    bits = _splitmix64(idx, stream_seed(seed, name))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def uniform(shape, lo: float, hi: float, seed: int, name: str, dtype=np.float32) -> np.ndarray:
    return (lo + (hi - lo) * uniform01(shape, seed, name)).astype(dtype)


def normal(shape, seed: int, name: str, dtype=np.float32) -> np.ndarray:
    """Box-Muller on two uniform streams."""
    u1 = uniform01(shape, seed, name + "/u1")
    u2 = uniform01(shape, seed, name + "/u2")
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * np.pi * u2)).astype(dtype)


def tsn_features(shape, seed: int, name: str = "rgb") -> np.ndarray:
    """Synthetic TSN-like features: ResNet-50 pooled activations are post-ReLU,
    i.e. non-negative and sparse-ish."""
    return np.maximum(normal(shape, seed, name), 0.0).astype(np.float32)


def miniroad_state_dict(cfg: dict, seed: int = 20, head_gain: float = 1.0) -> dict:
    """state_dict (numpy, fp32) with the reference MROAD key names and shapes
    (model/rnn/rnn.py:38-47; keys listed in SURVEY.md section 5).

    head_gain > 1 scales f_classification.0.weight so that the softmax is as
    peaked as a trained model's (random-init logits are nearly flat, which makes
    'identical argmax' a coin toss at 1e-6 margins).
    """
    from .config import input_dim  # local import: config has no heavy deps

    din = input_dim(cfg)
    e, h, c = cfg["embedding_dim"], cfg["hidden_dim"], cfg["num_classes"]
    b1 = 1.0 / np.sqrt(din)
    bh = 1.0 / np.sqrt(h)
    sd = {
        "gru.weight_ih_l0": uniform((3 * h, e), -bh, bh, seed, "gru.weight_ih_l0"),
        "gru.weight_hh_l0": uniform((3 * h, h), -bh, bh, seed, "gru.weight_hh_l0"),
        "gru.bias_ih_l0": uniform((3 * h,), -bh, bh, seed, "gru.bias_ih_l0"),
        "gru.bias_hh_l0": uniform((3 * h,), -bh, bh, seed, "gru.bias_hh_l0"),
        "layer1.0.weight": uniform((e, din), -b1, b1, seed, "layer1.0.weight"),
        "layer1.0.bias": uniform((e,), -b1, b1, seed, "layer1.0.bias"),
        "layer1.1.weight": uniform((e,), 0.75, 1.25, seed, "layer1.1.weight"),
        "layer1.1.bias": uniform((e,), -0.1, 0.1, seed, "layer1.1.bias"),
        "f_classification.0.weight": (head_gain * uniform((c, h), -bh, bh, seed, "f_classification.0.weight")).astype(np.float32),
        "f_classification.0.bias": uniform((c,), -bh, bh, seed, "f_classification.0.bias"),
    }
    for l in range(1, int(cfg.get("num_layers", 1))):        # stacked GRU (rnn.py:32,38): layer l reads layer l - 1's h_t
        sd[f"gru.weight_ih_l{l}"] = uniform((3 * h, h), -bh, bh, seed, f"gru.weight_ih_l{l}")
        sd[f"gru.weight_hh_l{l}"] = uniform((3 * h, h), -bh, bh, seed, f"gru.weight_hh_l{l}")
        sd[f"gru.bias_ih_l{l}"] = uniform((3 * h,), -bh, bh, seed, f"gru.bias_ih_l{l}")
        sd[f"gru.bias_hh_l{l}"] = uniform((3 * h,), -bh, bh, seed, f"gru.bias_hh_l{l}")
    return sd


def _lin(shape_out_in, seed, name):
    fan_in = shape_out_in[1]
    b = 1.0 / np.sqrt(fan_in)
    return uniform(shape_out_in, -b, b, seed, name + ".weight"), uniform((shape_out_in[0],), -b, b, seed, name + ".bias")


def vit_state_dict(cfg: dict, seed: int = 20) -> dict:
    """state_dict with the reference ViTEnc key names/shapes (ViT.py:25-90,
    Transformer.py:50-82, Attention.py:16-18, PositionalEncoding.py:25-34; keys in
    SURVEY.md section 5).  cls_token is zero-init in the reference; we give it a
    small random value so a kernel that drops it cannot pass."""
    from .config import input_dim

    din = input_dim(cfg) * cfg["patch_dim"] * cfg["patch_dim"]
    e, mlp, c = cfg["embedding_dim"], cfg["hidden_dim"], cfg["num_classes"]
    n = cfg["window_size"] // cfg["patch_dim"] + 1
    sd = {}
    sd["cls_token"] = uniform((1, 1, e), -0.5, 0.5, seed, "cls_token")
    sd["linear_encoding.weight"], sd["linear_encoding.bias"] = _lin((e, din), seed, "linear_encoding")
    sd["position_encoding.pe.weight"] = (0.5 * normal((n, e), seed, "position_encoding.pe.weight")).astype(np.float32)
    sd["position_encoding.position_ids"] = np.arange(n, dtype=np.int64)[None, :]
    for l in range(cfg["num_layers"]):
        a, f = 2 * l, 2 * l + 1
        sd[f"encoder.net.{a}.fn.norm.weight"] = uniform((e,), 0.75, 1.25, seed, f"enc{a}.norm.w")
        sd[f"encoder.net.{a}.fn.norm.bias"] = uniform((e,), -0.1, 0.1, seed, f"enc{a}.norm.b")
        b = 1.0 / np.sqrt(e)
        sd[f"encoder.net.{a}.fn.fn.qkv.weight"] = uniform((3 * e, e), -b, b, seed, f"enc{a}.qkv.w")
        sd[f"encoder.net.{a}.fn.fn.proj.weight"], sd[f"encoder.net.{a}.fn.fn.proj.bias"] = _lin((e, e), seed, f"enc{a}.proj")
        sd[f"encoder.net.{f}.fn.norm.weight"] = uniform((e,), 0.75, 1.25, seed, f"enc{f}.norm.w")
        sd[f"encoder.net.{f}.fn.norm.bias"] = uniform((e,), -0.1, 0.1, seed, f"enc{f}.norm.b")
        sd[f"encoder.net.{f}.fn.fn.net.0.weight"], sd[f"encoder.net.{f}.fn.fn.net.0.bias"] = _lin((mlp, e), seed, f"enc{f}.ff0")
        sd[f"encoder.net.{f}.fn.fn.net.3.weight"], sd[f"encoder.net.{f}.fn.fn.net.3.bias"] = _lin((e, mlp), seed, f"enc{f}.ff3")
    sd["pre_head_ln.weight"] = uniform((e,), 0.75, 1.25, seed, "pre_head_ln.w")
    sd["pre_head_ln.bias"] = uniform((e,), -0.1, 0.1, seed, "pre_head_ln.b")
    sd["mlp_head.weight"], sd["mlp_head.bias"] = _lin((c, e), seed, "mlp_head")
    return sd


def attention_layer_state_dict(d_model: int, seed: int = 20) -> dict:
    """AttentionLayer (attn.py:139-151) parameter names/shapes."""
    sd = {}
    for nm in ("query_projection", "key_projection", "value_projection", "out_projection"):
        sd[nm + ".weight"], sd[nm + ".bias"] = _lin((d_model, d_model), seed, "attnlayer." + nm)
    return sd


# ---- fixture G11: TRAINED weights -----------------------------------------------------------------------------------------------
G11_LEVELS = 7      # oracle/train_g11.py: the trained delta of every tensor on a 15-level grid, two levels per byte


def g11_state_dict(tag: str, golden_dir: str | None = None) -> dict:
    """The fixture model of G11: `miniroad_state_dict(cfg, seed 20)` + the de-quantised delta the imported reference's own training loop
    produced (oracle/train_g11.py: `train_one_epoch`, AdamW lr 1e-4 wd 0.05, dropout 0.2, on `workloads.action_video` windows), read
    from tests/golden/g11_weights_<tag>.npz.  tag: 'a101' (86 classes) or 'epic' (12).  Pure numpy, fp64 sums rounded once to fp32:
    the same bits on every box - the reference's Evaluate produced tests/golden/g11_eval_<tag>.npz from exactly these tensors."""
    import os
    from .config import assembly101_cfg, epic_tent_cfg
    cfg = {"a101": assembly101_cfg, "epic": epic_tent_cfg}[tag]()
    if golden_dir is None:
        golden_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    z = np.load(os.path.join(golden_dir, f"g11_weights_{tag}.npz"))
    sd = miniroad_state_dict(cfg, seed=20)
    out = {}
    for k, w0 in sd.items():
        packed = z["q." + k]
        u = np.empty(packed.size * 2, np.uint8)
        u[0::2] = packed & 15
        u[1::2] = packed >> 4
        q = u[: w0.size].astype(np.float64) - G11_LEVELS
        out[k] = (w0.astype(np.float64).reshape(-1) + q * float(z["scale." + k])).astype(np.float32).reshape(w0.shape)
    return out
