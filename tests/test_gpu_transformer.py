"""GPU parity of the transformer path (a11-a14): ViTEnc forward against the reference's output (G5) and the causal
AttentionLayer(FullAttention) against the reference's dead-code-but-only causal definition (G6) at L=128 and L=1024
(BASELINE config 4).  bf16 MFMA operands: tolerance 1e-2 relative to the output scale (logits are O(1))."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle_np as O            # noqa: E402
from prego_amd import weights as W           # noqa: E402
from prego_amd.config import assembly101_cfg  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")


def _vit_cfg(**kw):
    return assembly101_cfg(model="Transformer", window_size=128, patch_dim=1, num_heads=8, attn_dropout_rate=0.0, dropout=0.0, **kw)


# north star: 1e-3 (fp32) / 1e-2 (bf16); fp16 operands meet the fp32 figure on this path; the fp32-operand parity mode is held to 2e-5
VTOL = {"fp16": 1e-3, "bf16": 1e-2, "fp32": 2e-5}


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
def test_g5_vit_forward_matches_reference(dtype):
    from prego_amd.registry import build_model
    import prego_amd.transformer  # noqa: F401
    g = np.load(os.path.join(G, "g5_vit_forward.npz"))
    cfg = _vit_cfg(compute_dtype=dtype)
    sd = W.vit_state_dict(cfg, 20)
    m = build_model(cfg, "cuda:0")
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})   # strict: same keys as the reference
    m.eval()
    rgb = torch.from_numpy(W.tsn_features((2, 128, 2048), 20, "g5.rgb")).cuda()
    flow = torch.from_numpy(W.tsn_features((2, 128, 2048), 20, "g5.flow")).cuda()
    with torch.no_grad():
        out = m(rgb, flow)["logits"]
    torch.cuda.synchronize()
    assert out.shape == (2, 1, 86)
    got = out.cpu().numpy()
    err = np.abs(got - g["logits"]).max()
    print(f"vit logits {dtype} max abs err", err, "scale", np.abs(g["logits"]).max())
    assert err < VTOL[dtype] * max(1.0, np.abs(g["logits"]).max())
    assert np.array_equal(got.argmax(-1), g["logits"].argmax(-1))


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
def test_g5c_vit_long_window_1024_matches_reference(dtype):
    """BASELINE configs[3] sizes the transformer path at the LONG window: ViTEnc(window_size = 1024) = 1 025 tokens per window
    (ViT.py:117-143, learned positional table of 1 025 rows: PositionalEncoding.py:25-41), 8 heads of 256, 12 classes (Epic-tent-O).
    Against the reference's own logits (fixture G5c), and - the causal_attention extension on - against the numpy oracle; four
    windows so that the 4 100 token rows take the ping-pong GEMM epilogues and the 8-wave attention kernel's long-sequence form."""
    from prego_amd.registry import build_model
    import prego_amd.transformer  # noqa: F401
    g = np.load(os.path.join(G, "g5c_vit_forward_w1024.npz"))
    cfg = assembly101_cfg(model="Transformer", window_size=1024, patch_dim=1, num_heads=8, attn_dropout_rate=0.0, dropout=0.0, num_classes=12,
                          compute_dtype=dtype)
    sd = W.vit_state_dict(cfg, 20)
    assert sd["position_encoding.pe.weight"].shape == (1025, 2048)
    m = build_model(cfg, "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    rgb = W.tsn_features((4, 1024, 2048), 21, "g5c.more.rgb")
    flow = W.tsn_features((4, 1024, 2048), 21, "g5c.more.flow")
    rgb[:2] = W.tsn_features((2, 1024, 2048), 20, "g5c.rgb")
    flow[:2] = W.tsn_features((2, 1024, 2048), 20, "g5c.flow")
    trgb, tflow = torch.from_numpy(rgb).cuda(), torch.from_numpy(flow).cuda()
    with torch.no_grad():
        out = m(trgb, tflow)["logits"].cpu().numpy()
    assert out.shape == (4, 1, 12)
    scale = max(1.0, float(np.abs(g["logits"]).max()))
    err = float(np.abs(out[:2] - g["logits"]).max())
    print(f"vit window 1024 {dtype}: max abs err {err:.2e} (scale {scale:.2f})")
    assert err < VTOL[dtype] * scale
    assert np.array_equal(out[:2].argmax(-1), g["logits"].argmax(-1))
    if dtype == "fp32":
        return                                      # the fp32-operand mode has no causal attention kernel
    # causal_attention (extension): token i attends to tokens <= i; the cls token sits at the END, token 0 is what the head reads
    mc = build_model(dict(cfg, causal_attention=True), "cuda:0")
    mc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    mc.eval()
    with torch.no_grad():
        outc = mc(trgb[2:], tflow[2:])["logits"].cpu().numpy()
    ref = O.vit_forward(sd, rgb[2:], flow[2:], heads=8, causal=True)["logits"]
    errc = float(np.abs(outc - ref).max())
    print(f"vit window 1024 causal {dtype}: max abs err {errc:.2e}")
    assert errc < VTOL[dtype] * max(1.0, float(np.abs(ref).max()))


def test_fp32_operand_mode_two_layers_noncausal_and_its_limits():
    """compute_dtype='fp32' (parity mode): ViTEnc with two layers against G5b's reference logits, the unmasked AttentionLayer at a
    ragged length against the numpy oracle, and the entry points the mode does not cover fail loudly"""
    from prego_amd.registry import build_model
    from prego_amd.transformer import AttentionLayer
    from prego_amd._lib import PregoError
    import prego_amd.transformer  # noqa: F401
    g = np.load(os.path.join(G, "g5b_vit_train_L2.npz"))
    cfg = dict(_vit_cfg(compute_dtype="fp32"), num_layers=2)
    m = build_model(cfg, "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.vit_state_dict(cfg, 20).items()})
    m.eval()
    rgb = torch.from_numpy(W.tsn_features((2, 128, 2048), 20, "g5.rgb")).cuda()
    flow = torch.from_numpy(W.tsn_features((2, 128, 2048), 20, "g5.flow")).cuda()
    with torch.no_grad():
        got = m(rgb, flow)["logits"][:, 0].cpu().numpy()
    err = np.abs(got - g["logits"].reshape(got.shape)).max()
    print("vit 2 layers fp32 operands: max abs err", err)
    assert err < 2e-5 * max(1.0, np.abs(g["logits"]).max())
    with pytest.raises(PregoError, match="fp32-operand"):
        m.forward_frames(rgb[0], flow[0])
    sd = W.attention_layer_state_dict(1024, 21)
    names = ("query_projection", "key_projection", "value_projection", "out_projection")
    x = W.normal((2, 77, 1024), 21, "f32.x")
    ref = O.causal_attention_layer(x.astype(np.float64), *[sd[n + s].astype(np.float64) for n in names for s in (".weight", ".bias")],
                                   heads=8, mask_flag=False)
    layer = AttentionLayer(*[torch.from_numpy(sd[n + s]).cuda() for n in names for s in (".weight", ".bias")], n_heads=8,
                           mask_flag=False, compute_dtype="fp32")
    out = layer(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.abs(out - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
    # the training entry points refuse an fp32-operand handle (raw C ABI)
    import ctypes as C
    from prego_amd import _lib
    lib = _lib.load()
    xt = torch.from_numpy(x).cuda()
    o2 = torch.empty_like(xt)
    ws = torch.empty(1 << 26, dtype=torch.uint8, device="cuda")
    rc = lib.prego_attention_layer_forward_train(layer.h, 2, 77, 0, C.c_void_p(xt.data_ptr()), C.c_void_p(o2.data_ptr()), C.c_void_p(ws.data_ptr()),
                                                 ws.numel(), None)
    assert rc != 0 and b"bf16 handles" in lib.prego_last_error()


def test_vit_state_dict_keys_match_reference():
    from prego_amd.registry import build_model
    import prego_amd.transformer  # noqa: F401
    cfg = _vit_cfg()
    m = build_model(cfg, "cpu")
    ref = W.vit_state_dict(cfg, 20)
    sd = m.state_dict()
    assert set(sd.keys()) == set(ref.keys())
    for k in ref:
        assert tuple(sd[k].shape) == ref[k].shape, k
    assert sum(p.numel() for p in m.parameters()) == 29822038


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("L", [128, 1024])
def test_g6_causal_attention_layer(L, dtype):
    from prego_amd.transformer import attention_layer
    g = np.load(os.path.join(G, f"g6_causal_attention_L{L}.npz"))
    sd = W.attention_layer_state_dict(2048, 20)
    x = W.normal((1, L, 2048), 20, f"g6.x.{L}")
    names = ("query_projection", "key_projection", "value_projection", "out_projection")
    args = [torch.from_numpy(sd[n + s]).cuda() for n in names for s in (".weight", ".bias")]
    out = attention_layer(torch.from_numpy(x).cuda(), *args, n_heads=8, mask_flag=True, compute_dtype=dtype)[0].cpu().numpy()
    ref = g["out"]
    err = np.abs(out[g["rows"]] - ref).max()
    print(f"causal attention L={L} {dtype}: max abs err {err:.3e}, output scale {np.abs(ref).max():.3f}")
    assert err < VTOL[dtype] * max(1.0, np.abs(ref).max())
    # causality property at full size: perturbing the last frame leaves rows 0..L-2 bit-identical
    x2 = x.copy()
    x2[0, -1] += 1.0
    out2 = attention_layer(torch.from_numpy(x2).cuda(), *args, n_heads=8, mask_flag=True, compute_dtype=dtype)[0].cpu().numpy()
    assert np.array_equal(out[:-1], out2[:-1])
    assert not np.array_equal(out[-1], out2[-1])


@pytest.mark.parametrize("d,B,L,mask", [(512, 2, 192, True), (2048, 1, 128, True), (1024, 2, 100, False)])
def test_g6b_attention_layer_backward(d, B, L, mask):
    """AttentionLayer under autograd (attn.py:151-170 inside loss.backward()): gradients of x and of the eight projection parameters
    against the reference's own autograd (fixture G6b) - cosine over the sampled entries and norm, the bar of the ViT training test -
    and every tensor in full against the numpy backward of the oracle."""
    from prego_amd.transformer import AttentionLayer
    g = np.load(os.path.join(G, f"g6b_attention_grads_d{d}_L{L}_{'causal' if mask else 'full'}.npz"))
    sd = W.attention_layer_state_dict(d, 20)
    names = ("query_projection", "key_projection", "value_projection", "out_projection")
    keys = [n + s for n in names for s in (".weight", ".bias")]
    params = [torch.from_numpy(sd[k]).cuda().requires_grad_(True) for k in keys]
    x_np, G_np = W.normal((B, L, d), 20, f"g6b.x.{d}.{L}"), W.normal((B, L, d), 20, f"g6b.g.{d}.{L}")
    x = torch.from_numpy(x_np).cuda().requires_grad_(True)
    layer = AttentionLayer(*params, n_heads=8, mask_flag=mask)
    out = layer(x)
    assert out.requires_grad
    (out * torch.from_numpy(G_np).cuda()).sum().backward()
    dx_o, grads_o = O.causal_attention_layer_grads(x_np.astype(np.float64), *[sd[k].astype(np.float64) for k in keys], heads=8,
                                                   dout=G_np.astype(np.float64), mask_flag=mask)
    assert abs(float(out.detach().double().norm()) - float(g["out_norm"])) < 1e-2 * float(g["out_norm"])
    worst = (2.0, "")
    qb = float(g["norm.query_projection.bias"])
    for k, t, full in [("x", x, dx_o)] + list(zip(keys, params, grads_o)):
        got = t.grad.detach().cpu().numpy().astype(np.float64)
        if k == "key_projection.bias":          # zero in real arithmetic: what is left is rounding, small against its sibling
            assert np.linalg.norm(got) < 0.05 * qb, (np.linalg.norm(got), qb)
            continue
        ref_n = float(g["norm." + k])
        assert abs(np.linalg.norm(got) - ref_n) < 0.05 * ref_n, (k, np.linalg.norm(got), ref_n)
        a, r = got.reshape(-1)[g["idx." + k]], g["val." + k].astype(np.float64)
        cos_s = float(a @ r / (np.linalg.norm(a) * np.linalg.norm(r) + 1e-30))
        cos_f = float(got.reshape(-1) @ full.reshape(-1) / (np.linalg.norm(got) * np.linalg.norm(full) + 1e-30))
        worst = min(worst, (min(cos_s, cos_f), k))
        assert cos_s > 0.995 and cos_f > 0.999, (k, cos_s, cos_f)
    print(f"attention layer backward d={d} L={L} mask={mask}: worst cosine {worst[0]:.5f} ({worst[1]})")
    # the parameters move (an optimizer step): the training handle re-ingests them, eval and train forwards agree again
    with torch.no_grad():
        params[6].mul_(0.5)
    out2 = layer(x)
    assert torch.allclose(out2.detach() - params[7].detach(), (out.detach() - params[7].detach()) * 0.5, atol=2e-2 * float(out.detach().abs().max()))
    with torch.no_grad():
        ev = AttentionLayer(*[p.detach() for p in params], n_heads=8, mask_flag=mask, compute_dtype="bf16")(x.detach())
    assert torch.equal(ev, out2.detach())
    # a stale graph (another training forward since) is refused
    with pytest.raises(Exception, match="another training forward"):
        out.sum().backward()


def test_attention_layer_attention_dropout_in_training_mode():
    """FullAttention(attention_dropout=p) (attn.py:36-39,54: nn.Dropout on A = softmax(scale * scores)) on the autograd path.  torch's RNG
    stream cannot be matched, so the mask is held to what defines it: with zero query / key projections A is uniform (1 / L), with a zero
    value weight every value row is the bias c, with an identity out_projection the layer's output row i, head h is s[i, h] * c where
    s = (kept probabilities of the row) / (1 - p).  So (1) the kept fraction is 1 - p and s has the binomial mean 1 and variance
    p / ((1 - p) L), rows and heads draw different masks; (2) the gradient of the value bias is sum_i s[i, h] * dout[i, :]: the BACKWARD
    regenerated exactly the mask of its forward; (3) a new forward draws a new mask; (4) eval() drops nothing, bit for bit."""
    from prego_amd.transformer import AttentionLayer
    d, H, B, L, p = 512, 8, 2, 256, 0.25
    dh = d // H
    g = torch.Generator(device="cuda").manual_seed(3)
    c = torch.rand(d, device="cuda", generator=g) + 0.5
    z = lambda *sh: torch.zeros(*sh, device="cuda")
    params = [z(d, d), z(d), z(d, d), z(d), z(d, d), c.clone(), torch.eye(d, device="cuda"), z(d)]
    params = [t.requires_grad_(True) for t in params]
    x = torch.randn(B, L, d, device="cuda", generator=g)
    dout = torch.rand(B, L, d, device="cuda", generator=g) + 0.5
    layer = AttentionLayer(*params, n_heads=H, mask_flag=False, attention_dropout=p)
    torch.manual_seed(11)
    out = layer(x)
    (out * dout).sum().backward()
    s = (out.detach() / c).reshape(B, L, H, dh)
    assert float((s.std(dim=-1) / s.mean(dim=-1).clamp_min(1e-6)).max()) < 2e-2          # one factor per (row, head): bf16 noise only
    s = s.mean(-1)                                                                        # [B, L, H]
    kept = s * (1 - p) * L                                                                # kept keys per row and head
    assert float((kept - kept.round()).abs().max()) < 0.6                                 # whole keys (bf16 probabilities: 1 / 256 is exact)
    frac = float(kept.sum() / (B * L * H * L))
    assert abs(frac - (1 - p)) < 5e-3, frac
    assert abs(float(s.mean()) - 1.0) < 5e-3
    var_ref = p / ((1 - p) * L)
    assert 0.8 * var_ref < float(s.var()) < 1.25 * var_ref, (float(s.var()), var_ref)
    assert float(s[0, 0].std()) > 0 and float(s[:, :, 0].std()) > 0                       # heads and rows draw their own masks
    # (2) d loss / d bv[k] = sum over rows of s[row, head(k)] * dout[row, k]
    ref_gbv = (s.unsqueeze(-1) * dout.reshape(B, L, H, dh)).sum((0, 1)).reshape(d)
    got_gbv = params[5].grad
    assert float((got_gbv - ref_gbv).abs().max() / ref_gbv.abs().max()) < 1e-2
    # (3) another forward: another mask; (4) eval(): none
    out2 = layer(x)
    assert not torch.equal(out2.detach(), out.detach())
    layer.eval()
    out3 = layer(x)
    ref = AttentionLayer(*[t.detach() for t in params], n_heads=H, mask_flag=False, compute_dtype="bf16")(x)
    assert torch.equal(out3.detach(), ref)
    with pytest.raises(Exception, match="attention_dropout"):
        AttentionLayer(*[t.detach() for t in params], n_heads=H, attention_dropout=1.0)


def test_noncausal_attention_vs_oracle_ragged_length():
    """L = 129 (window + cls token; not a multiple of any tile) without mask, against the numpy oracle"""
    from prego_amd.transformer import attention_layer
    sd = W.attention_layer_state_dict(2048, 21)
    x = W.normal((2, 129, 2048), 21, "nc.x")
    names = ("query_projection", "key_projection", "value_projection", "out_projection")
    np_args = [sd[n + s].astype(np.float64) for n in names for s in (".weight", ".bias")]
    ref = O.causal_attention_layer(x.astype(np.float64), *np_args, heads=8, mask_flag=False)
    args = [torch.from_numpy(sd[n + s]).cuda() for n in names for s in (".weight", ".bias")]
    out = attention_layer(torch.from_numpy(x).cuda(), *args, n_heads=8, mask_flag=False).cpu().numpy()
    assert np.abs(out - ref).max() < 1e-2 * max(1.0, np.abs(ref).max())


def test_vit_large_batch_uses_the_pingpong_epilogues_and_matches():
    """40 windows = 5 160 token rows: above the 4 096-row switch, so the QKV-split, residual and GELU epilogues run on the 256x256
    ping-pong kernel.  Windows 0-1 are the reference fixture's inputs (G5); every window must also agree with the same window
    pushed through the small-batch path (128x128 kernel epilogues)."""
    from prego_amd.registry import build_model
    import prego_amd.transformer  # noqa: F401
    g = np.load(os.path.join(G, "g5_vit_forward.npz"))
    cfg = _vit_cfg()
    m = build_model(cfg, "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.vit_state_dict(cfg, 20).items()})
    m.eval()
    B = 40
    rgb = torch.from_numpy(W.tsn_features((B, 128, 2048), 23, "big.rgb")).cuda()
    flow = torch.from_numpy(W.tsn_features((B, 128, 2048), 23, "big.flow")).cuda()
    rgb[:2] = torch.from_numpy(W.tsn_features((2, 128, 2048), 20, "g5.rgb")).cuda()
    flow[:2] = torch.from_numpy(W.tsn_features((2, 128, 2048), 20, "g5.flow")).cuda()
    with torch.no_grad():
        big = m(rgb, flow)["logits"].cpu().numpy()
        small = np.concatenate([m(rgb[i:i + 2], flow[i:i + 2])["logits"].cpu().numpy() for i in range(0, B, 2)])
    scale = max(1.0, np.abs(g["logits"]).max())
    assert np.abs(big[:2] - g["logits"]).max() < 1e-2 * scale
    assert np.array_equal(big[:2].argmax(-1), g["logits"].argmax(-1))
    assert np.abs(big - small).max() < 2e-3 * scale, np.abs(big - small).max()      # same math, different kernels / summation tiling


def test_causal_attention_layer_large_batch_matches_per_sample():
    """B = 5 x L = 1024 = 5 120 rows: the projections run on the ping-pong kernel (QKV epilogue); sample 0 is the G6 fixture input"""
    from prego_amd.transformer import attention_layer
    L = 1024
    g = np.load(os.path.join(G, f"g6_causal_attention_L{L}.npz"))
    sd = W.attention_layer_state_dict(2048, 20)
    x = W.normal((5, L, 2048), 24, "big.x")
    x[0] = W.normal((1, L, 2048), 20, f"g6.x.{L}")[0]
    names = ("query_projection", "key_projection", "value_projection", "out_projection")
    args = [torch.from_numpy(sd[n + s]).cuda() for n in names for s in (".weight", ".bias")]
    xt = torch.from_numpy(x).cuda()
    out = attention_layer(xt, *args, n_heads=8, mask_flag=True).cpu().numpy()
    scale = max(1.0, np.abs(g["out"]).max())
    assert np.abs(out[0][g["rows"]] - g["out"]).max() < 1e-2 * scale
    for b in (1, 4):
        one = attention_layer(xt[b:b + 1], *args, n_heads=8, mask_flag=True)[0].cpu().numpy()
        assert np.abs(out[b] - one).max() < 2e-3 * scale, (b, np.abs(out[b] - one).max())


@pytest.mark.parametrize("layers", [1, 2])
def test_last_block_token0_path_equals_all_rows(layers):
    """The last encoder block computes only token 0 (ViT.py:136 reads nothing else): same logits as running every row, and
    the 2-layer model (every row of block 1 feeds block 2's keys / values) against the reference fixture G5b."""
    from prego_amd.registry import build_model
    import prego_amd.transformer  # noqa: F401
    cfg = dict(_vit_cfg(), num_layers=layers)
    sd = W.vit_state_dict(cfg, 20)
    m = build_model(cfg, "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    rgb = torch.from_numpy(W.tsn_features((2, 128, 2048), 20, "g5.rgb")).cuda()
    flow = torch.from_numpy(W.tsn_features((2, 128, 2048), 20, "g5.flow")).cuda()
    with torch.no_grad():
        fast = m(rgb, flow)["logits"].cpu().numpy()
        m.debug_all_rows = True
        full = m(rgb, flow)["logits"].cpu().numpy()
    assert np.abs(fast - full).max() < 2e-3           # same arithmetic up to the GEMM kernel chosen for M = B rows
    g = np.load(os.path.join(G, f"g5b_vit_train_L{layers}.npz"))
    assert np.abs(fast - g["logits"]).max() < 1e-2
    assert np.abs(full - g["logits"]).max() < 1e-2


def test_attention_layer_fp32_mode_refuses_the_bf16_autograd_path():
    """compute_dtype='fp32' is the parity mode: with nn.Parameter weights outside torch.no_grad() the layer would take the bf16-operand
    training path - it raises instead of returning different numbers depending on grad mode (round-3 advisor); under no_grad it
    matches the fixture at the fp32 tolerance."""
    from prego_amd._lib import PregoError
    from prego_amd.transformer import AttentionLayer
    L = 128
    g = np.load(os.path.join(G, f"g6_causal_attention_L{L}.npz"))
    sd = W.attention_layer_state_dict(2048, 20)
    names = ("query_projection", "key_projection", "value_projection", "out_projection")
    params = [torch.nn.Parameter(torch.from_numpy(sd[n + s]).cuda()) for n in names for s in (".weight", ".bias")]
    layer = AttentionLayer(*params, n_heads=8, mask_flag=True, compute_dtype="fp32")
    x = torch.from_numpy(W.normal((1, L, 2048), 20, f"g6.x.{L}")).cuda()
    with pytest.raises(PregoError):
        layer(x)
    with torch.no_grad():
        out = layer(x)[0].cpu().numpy()
    assert np.abs(out[g["rows"]] - g["out"]).max() < 2e-5 * max(1.0, np.abs(g["out"]).max())


def test_attention_layer_stateless_op_equals_handle():
    """the stateless C-ABI op (weights converted per call) and the handle (converted once) run the same arithmetic"""
    import ctypes as C
    from prego_amd import _lib
    from prego_amd.transformer import AttentionLayer
    lib = _lib.load()
    d, H, B, L = 2048, 8, 2, 200
    sd = W.attention_layer_state_dict(d, 20)
    names = ("query_projection", "key_projection", "value_projection", "out_projection")
    ws = [torch.from_numpy(sd[n + s]).cuda() for n in names for s in (".weight", ".bias")]
    x = torch.from_numpy(W.normal((B, L, d), 20, "attn.handle.x")).cuda()
    a = AttentionLayer(*ws, n_heads=H, mask_flag=True, compute_dtype="bf16")(x)      # the stateless op is the bf16 form
    need = lib.prego_attention_layer_workspace_bytes(B, L, d)
    wsb = torch.empty(need, dtype=torch.uint8, device="cuda")
    out = torch.empty_like(x)
    _lib.check(lib.prego_attention_layer_forward(B, L, d, H, 1, C.c_void_p(x.data_ptr()), *[C.c_void_p(t.data_ptr()) for t in ws],
                                                 C.c_void_p(out.data_ptr()), C.c_void_p(wsb.data_ptr()), need, None))
    torch.cuda.synchronize()
    assert torch.equal(a, out)
    ref = O.causal_attention_layer(x.cpu().numpy().astype(np.float64), *[t.cpu().numpy().astype(np.float64) for t in ws], heads=H)
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-2


def test_vit_no_rgb_flow_only():
    """--no_rgb on the Transformer entry (ViT.py:118-123): the encoding Linear sees the flow stream only"""
    from prego_amd.registry import build_model
    import prego_amd.transformer  # noqa: F401
    cfg = _vit_cfg(no_rgb=True)
    sd = W.vit_state_dict(cfg, 20)
    assert sd["linear_encoding.weight"].shape == (2048, 2048)
    m = build_model(cfg, "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    x = W.tsn_features((2, 128, 2048), 20, "vit.norgb")
    with torch.no_grad():
        got = m(torch.zeros(2, 128, 0, device="cuda"), torch.from_numpy(x).cuda())["logits"].cpu().numpy()
    ref = O.vit_forward(sd, x, None, heads=8)["logits"]
    assert np.abs(got - ref).max() < 1e-2


@pytest.mark.parametrize("Nq,N,dh,causal", [
    (129, 129, 256, 0), (1, 129, 256, 0),            # ViTEnc window + cls token; the token-0-only last block
    (200, 200, 64, 1), (300, 300, 128, 1), (257, 257, 256, 1), (193, 193, 128, 0), (448, 448, 64, 0),   # 8-wave shape, ragged
    (1024, 1024, 256, 1),                            # BASELINE configs[3]: 8 causal query blocks of 16 ... 2 key tiles
    (64, 64, 128, 1), (130, 130, 64, 1), (129, 129, 256, 1), (1, 129, 256, 1),   # 4-wave shape, causal (incl. token 0 only)
])
def test_attention_forward_kernel_vs_fp64(Nq, N, dh, causal):
    """softmax(q k^T dh^-0.5 [+ triu mask]) v (Attention.py:30-38, attn.py:43-55) for every head dim and workgroup shape of the
    forward kernel, queries at positions 0..Nq-1, and its log-sum-exp output (what the backward pass recomputes P from).
    Inputs are bf16-exact, so only the kernel's own rounding shows (P and the output are rounded to bf16)."""
    import ctypes as C
    from prego_amd import _lib
    lib = _lib.load_debug()          # prego_debug_attention_fwd: only in libprego_amd_debug.so
    B, h = 2, 3
    rng = np.random.default_rng(Nq * 7 + N + dh + causal)

    def bf(a):
        return torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)
    q = bf(rng.standard_normal((B, h, Nq, dh)) * 1.5)
    k = bf(rng.standard_normal((B, h, N, dh)))
    v = bf(rng.standard_normal((B, h, N, dh)))
    qs = bf(q * dh ** -0.5)
    s = np.einsum("bhid,bhjd->bhij", qs, k)
    if causal:
        s = np.where(np.arange(N)[None, :] > np.arange(Nq)[:, None], -np.inf, s)
    mx = s.max(-1, keepdims=True)
    p = np.exp(s - mx)
    lse = (mx + np.log(p.sum(-1, keepdims=True)))[..., 0]
    want = np.einsum("bhij,bhjd->bhid", p / p.sum(-1, keepdims=True), v).transpose(0, 2, 1, 3).reshape(B, Nq, h * dh)

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).cuda()
    tq, tk, tv = dev(qs), dev(k), dev(v)
    out = torch.full((B, Nq, h * dh), float("nan"), dtype=torch.bfloat16, device="cuda")
    tl = torch.full((B, h, Nq), float("nan"), dtype=torch.float32, device="cuda")
    rc = lib.prego_debug_attention_fwd(B, Nq, N, h, dh, causal, tq.data_ptr(), tk.data_ptr(), tv.data_ptr(), out.data_ptr(),
                                       tl.data_ptr(), None)
    assert rc == 0
    got = out.float().cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    err = np.abs(got - want).max()
    assert err < 1e-2 * max(1.0, np.abs(want).max()), (err, np.abs(want).max())
    assert np.abs(tl.cpu().numpy() - lse).max() < 2e-3
    # without the lse output (the inference path) the result is the same, bit for bit
    out2 = torch.empty_like(out)
    assert lib.prego_debug_attention_fwd(B, Nq, N, h, dh, causal, tq.data_ptr(), tk.data_ptr(), tv.data_ptr(), out2.data_ptr(),
                                         None, None) == 0
    assert torch.equal(out, out2)


@pytest.mark.parametrize("window,T,layers", [(32, 300, 1), (128, 40, 1), (32, 70, 2)])
def test_vit_per_frame_eval_runner_vs_oracle_windows(window, T, layers):
    """prego_vit_forward_frames (ViTEnc.forward_frames): logits[t] = the reference forward on the `window` frames ending at t
    with zero feature rows in front of the video (dataset.py:53-55,96-103 at stride 1) - checked against oracle_np.vit_forward
    window by window, and against the batched prego_vit_forward on the same windows (same kernels behind the encoding: tight).
    linear_encoding runs once per frame inside the runner; one-layer models never materialise the residual stream."""
    from prego_amd.config import assembly101_cfg
    from prego_amd.registry import build_model
    import prego_amd.transformer  # noqa: F401
    cfg = assembly101_cfg(model="Transformer", window_size=window, patch_dim=1, num_heads=8, attn_dropout_rate=0.0, dropout=0.0,
                          num_layers=layers)
    sd = W.vit_state_dict(cfg, 20)
    m = build_model(cfg, "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    m.windows_per_batch = 64                     # several batches, a ragged last one
    rgb = W.tsn_features((T, 2048), 41, "vf.rgb")
    flow = W.tsn_features((T, 2048), 41, "vf.flow")
    got, arg = m.forward_frames(torch.from_numpy(rgb).cuda(), torch.from_numpy(flow).cuda())
    got, arg = got.cpu().numpy(), arg.cpu().numpy()
    assert got.shape == (T, 86) and np.array_equal(arg, got.argmax(1))
    # the windows, as the training loader would cut them
    pr = np.concatenate([np.zeros((window - 1, 2048), np.float32), rgb])
    pf = np.concatenate([np.zeros((window - 1, 2048), np.float32), flow])
    wr = np.stack([pr[t:t + window] for t in range(T)])
    wf = np.stack([pf[t:t + window] for t in range(T)])
    idx = np.unique(np.concatenate([np.arange(0, min(T, 6)), np.linspace(0, T - 1, 24).astype(int), [window - 2, window - 1, window] if T > window else []]).astype(int))
    ref = O.vit_forward(sd, wr[idx], wf[idx], 8, num_layers=layers)["logits"][:, 0]
    scale = np.abs(ref).max()
    assert np.abs(got[idx] - ref).max() < 1e-2 * max(1.0, scale), np.abs(got[idx] - ref).max()
    with torch.no_grad():
        bat = torch.cat([m(torch.from_numpy(wr[i:i + 64]).cuda(), torch.from_numpy(wf[i:i + 64]).cuda())["logits"][:, 0] for i in range(0, T, 64)]).cpu().numpy()
    assert np.abs(got - bat).max() < 2e-3 * max(1.0, scale), np.abs(got - bat).max()
    # the batched-eval interface Evaluate drives
    outs, args, _ = m.forward_clips([torch.from_numpy(rgb).cuda()], [torch.from_numpy(flow).cuda()])
    assert np.array_equal(outs[0].cpu().numpy(), got) and np.array_equal(args[0].cpu().numpy(), arg)
