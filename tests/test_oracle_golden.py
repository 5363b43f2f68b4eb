"""Pin the numpy oracle (oracle/oracle_np.py) against the golden vectors that
oracle/gen_golden.py produced from the real reference (CPU only)."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import oracle_np as O
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg, epic_tent_cfg

G = os.path.join(os.path.dirname(__file__), "golden")


def _ld(name):
    return np.load(os.path.join(G, name))


@pytest.mark.parametrize("tag,gain", [("plain", 1.0), ("peaky", 8.0)])
def test_g1_miniroad_eval_cfg1(tag, gain):
    """BASELINE config 1: 1 clip x 256 frames x 2048-d, zero flow, CPU."""
    g = _ld(f"g1_miniroad_eval_{tag}.npz")
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=gain)
    rgb = W.tsn_features((1, 256, 2048), 20, "g1.rgb")
    out = O.miniroad_forward(sd, rgb, None, keep=True)            # zero flow == skipped columns
    assert np.abs(out["logits"][0] - g["probs"]).max() < 2e-6   # fp64 oracle vs torch fp32
    assert np.abs(out["h_last"][0] - g["h_last"]).max() < 2e-6
    srt = np.sort(g["probs"], 1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-5
    assert np.array_equal(out["logits"][0].argmax(1)[safe], g["argmax"][safe])
    # the same with an explicit all-zero flow tensor (the reference's actual call)
    out2 = O.miniroad_forward(sd, rgb, np.zeros_like(rgb))
    assert np.abs(out2["logits"] - out["logits"]).max() < 1e-12


def test_g1c_trained_like_head_gain32():
    """head gain 32 (top-1 probabilities near 1): the oracle against the reference's output, 1024 frames, rgb + flow"""
    g = _ld("g1c_miniroad_eval_gain32.npz")
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=32.0)
    out = O.miniroad_forward(sd, W.tsn_features((1, 1024, 2048), 20, "g1c.rgb"), W.tsn_features((1, 1024, 2048), 20, "g1c.flow"))
    assert np.abs(out["logits"][0] - g["probs"]).max() < 2e-5       # fp64 oracle vs torch fp32 at logit scale ~30
    safe = g["margin"] > 1e-4
    assert np.array_equal(out["logits"][0].argmax(1)[safe], g["argmax"][safe])


def test_g2_long_T_4096_nonzero_flow():
    g = _ld("g2_miniroad_longT_4096.npz")
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    T = 4096
    rgb = W.tsn_features((1, T, 2048), 20, f"g2.rgb.{T}")
    flow = W.tsn_features((1, T, 2048), 20, f"g2.flow.{T}")
    out = O.miniroad_forward(sd, rgb, flow, dt=np.float32)["logits"][0]
    assert np.abs(out[g["sample_idx"]] - g["sample_probs"]).max() < 5e-5
    safe = g["margin"] > 2e-4
    assert np.array_equal(out.argmax(1)[safe], g["argmax"][safe].astype(np.int64))


def test_g3_intermediates():
    g = _ld("g3_miniroad_intermediates.npz")
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20)
    rgb = W.tsn_features((1, 8, 2048), 20, "g3.rgb")
    flow = W.tsn_features((1, 8, 2048), 20, "g3.flow")
    out = O.miniroad_forward(sd, rgb, flow, keep=True)
    for k, tol in (("y", 5e-6), ("e", 2e-5), ("h", 2e-6), ("raw_logits", 2e-6)):
        assert np.abs(out[k][0] - g[k]).max() < tol, k
    assert np.abs(out["logits"][0] - g["probs"]).max() < 1e-6


def _small_cfg():
    return assembly101_cfg(rgb_type="rgb_kinetics_bninception", no_flow=True, embedding_dim=128,
                           hidden_dim=64, num_classes=12, dropout=0.0, window_size=16, batch_size=4)


def _targets(B, T, C, seed, name):
    cls = (W.uniform01((B, T), seed, name) * C).astype(np.int64)
    tgt = np.zeros((B, T, C), dtype=np.float32)
    bi, ti = np.meshgrid(np.arange(B), np.arange(T), indexing="ij")
    tgt[bi, ti, cls] = 1.0
    return tgt


def test_g4_loss_grads_adamw_small():
    g = _ld("g4_miniroad_train_small.npz")
    cfg = _small_cfg()
    sd = {k: v.astype(np.float64) for k, v in W.miniroad_state_dict(cfg, 20).items()}
    rgb = W.tsn_features((4, 16, 1024), 20, "g4.rgb")
    tgt = _targets(4, 16, 12, 20, "g4.tgt")
    tgt[3, -1] = 0.0
    tgt[2, -1, 5] = 1.0
    m = {k: np.zeros_like(v) for k, v in sd.items()}
    v = {k: np.zeros_like(vv) for k, vv in sd.items()}
    for step in range(3):
        loss, grads = O.miniroad_loss_and_grads(sd, rgb, None, tgt)
        assert abs(loss - g["losses"][step]) < 2e-6
        if step == 0:
            for k in sd:
                ref = g["grad." + k]
                assert np.abs(grads[k] - ref).max() < 1e-6 + 1e-4 * np.abs(ref).max(), k
        for k in sd:
            sd[k], m[k], v[k] = O.adamw_step(sd[k], grads[k], m[k], v[k], step + 1)
        if step in (0, 2):
            for k in sd:
                assert np.abs(sd[k] - g[f"param{step + 1}." + k]).max() < 2e-6, k


@pytest.mark.parametrize("tag,hid,layers", [("h512", 512, 1), ("h2048", 2048, 1), ("h1024_l2", 1024, 2), ("h512_l2", 512, 2)])
def test_g10_other_hidden_sizes_and_two_gru_layers(tag, hid, layers):
    """rnn.py:31-38 takes any hidden_dim / num_layers: the oracle's stacked-GRU restatement against the reference at 512 / 2048 and 2 layers"""
    g = _ld(f"g10_miniroad_eval_{tag}.npz")
    cfg = assembly101_cfg(hidden_dim=hid, num_layers=layers)
    sd = W.miniroad_state_dict(cfg, seed=20, head_gain=8.0)
    for i, T in enumerate((96, 40)):
        rgb = W.tsn_features((1, T, 2048), 20, f"g10.{tag}.rgb.{i}")
        flow = W.tsn_features((1, T, 2048), 20, f"g10.{tag}.flow.{i}")
        out = O.miniroad_forward(sd, rgb, flow, keep=True)
        assert np.abs(out["logits"][0] - g[f"probs{i}"]).max() < 5e-6
        hn = out["h_last"] if layers == 2 else out["h_last"][None]
        assert np.abs(hn[:, 0] - g[f"h_n{i}"]).max() < 5e-6
        ok = g[f"margin{i}"] > 1e-5
        assert np.array_equal(out["logits"][0].argmax(1)[ok], g[f"argmax{i}"][ok])


def test_g4s_oadloss_reduction_sum():
    """OadLoss(reduction='sum') of the reference (criterions/loss.py:30-33) beside 'mean': value and gradient"""
    g = _ld("g4s_oadloss_sum.npz")
    lg, tg = g["logits"].astype(np.float64), g["target"].astype(np.float64)
    for red in ("mean", "sum"):
        assert abs(O.oad_loss(lg, tg, red) - float(g[f"loss_{red}"])) < 2e-6 * max(1.0, abs(float(g[f"loss_{red}"])))
        assert np.abs(O.oad_loss_grad(lg, tg, red) - g[f"dlogits_{red}"]).max() < 2e-7
    assert abs(float(g["loss_sum"]) - 6 * float(g["loss_mean"])) < 1e-4


def test_g4b_loss_grads_full_dims():
    g = _ld("g4b_miniroad_train_full.npz")
    cfg = assembly101_cfg(dropout=0.0)
    sd = W.miniroad_state_dict(cfg, 20)
    rgb = W.tsn_features((2, 8, 2048), 20, "g4b.rgb")
    flow = W.tsn_features((2, 8, 2048), 20, "g4b.flow")
    tgt = _targets(2, 8, 86, 20, "g4b.tgt")
    loss, grads = O.miniroad_loss_and_grads(sd, rgb, flow, tgt)
    assert abs(loss - float(g["loss"])) < 2e-6
    for k in sd:
        gg = grads[k].reshape(-1)
        assert abs(np.linalg.norm(gg) - float(g["norm." + k])) < 1e-4 * float(g["norm." + k]) + 1e-9, k
        ref = g["val." + k]
        assert np.abs(gg[g["idx." + k]] - ref).max() < 1e-7 + 1e-3 * np.abs(ref).max(), k


def _vit_cfg():
    return assembly101_cfg(model="Transformer", window_size=128, patch_dim=1, num_heads=8,
                           attn_dropout_rate=0.0, dropout=0.0)


def test_g5_vit_forward():
    g = _ld("g5_vit_forward.npz")
    cfg = _vit_cfg()
    sd = W.vit_state_dict(cfg, 20)
    rgb = W.tsn_features((2, 128, 2048), 20, "g5.rgb")
    flow = W.tsn_features((2, 128, 2048), 20, "g5.flow")
    out = O.vit_forward(sd, rgb, flow, heads=8, keep=True)
    assert out["logits"].shape == (2, 1, 86)
    assert np.abs(out["logits"] - g["logits"]).max() < 2e-5
    for k in ("ln1", "attn", "ln2", "ffn"):
        assert np.abs(out[k][:, g["rows"]] - g[k]).max() < 5e-5, k


def test_g5c_vit_forward_long_window_1024():
    """the oracle at BASELINE configs[3]'s long window (N = 1025 tokens, 12 classes) against the reference's own output"""
    g = _ld("g5c_vit_forward_w1024.npz")
    cfg = assembly101_cfg(model="Transformer", window_size=1024, patch_dim=1, num_heads=8, attn_dropout_rate=0.0, dropout=0.0, num_classes=12)
    sd = W.vit_state_dict(cfg, 20)
    rgb = W.tsn_features((2, 1024, 2048), 20, "g5c.rgb")
    flow = W.tsn_features((2, 1024, 2048), 20, "g5c.flow")
    out = O.vit_forward(sd, rgb, flow, heads=8, keep=True)
    assert out["logits"].shape == (2, 1, 12)
    assert np.abs(out["logits"] - g["logits"]).max() < 2e-5
    for k in ("attn", "ffn"):
        assert np.abs(out[k][:, g["rows"]] - g[k]).max() < 5e-5, k


@pytest.mark.parametrize("L", [128, 1024])
def test_g6_causal_attention(L):
    g = _ld(f"g6_causal_attention_L{L}.npz")
    sd = W.attention_layer_state_dict(2048, 20)
    x = W.normal((1, L, 2048), 20, f"g6.x.{L}")
    args = [sd[n + s].astype(np.float64) for n in ("query_projection", "key_projection", "value_projection", "out_projection")
            for s in (".weight", ".bias")]
    o = O.causal_attention_layer(x.astype(np.float64), *args, heads=8)
    assert np.abs(o[0][g["rows"]] - g["out"]).max() < 2e-5
    # causality property
    x2 = x.copy()
    x2[0, -1] += 1.0
    o2 = O.causal_attention_layer(x2.astype(np.float64), *args, heads=8)
    assert np.array_equal(o[0, :-1], o2[0, :-1])


G6B = [(512, 2, 192, True), (2048, 1, 128, True), (1024, 2, 100, False)]
_ATTN = ("query_projection", "key_projection", "value_projection", "out_projection")


@pytest.mark.parametrize("d,B,L,mask", G6B)
def test_g6b_attention_layer_gradients(d, B, L, mask):
    """the numpy backward of the AttentionLayer against autograd through the reference's module (attn.py:139-170)"""
    g = _ld(f"g6b_attention_grads_d{d}_L{L}_{'causal' if mask else 'full'}.npz")
    sd = W.attention_layer_state_dict(d, 20)
    x = W.normal((B, L, d), 20, f"g6b.x.{d}.{L}").astype(np.float64)
    G_ = W.normal((B, L, d), 20, f"g6b.g.{d}.{L}").astype(np.float64)
    args = [sd[n + s].astype(np.float64) for n in _ATTN for s in (".weight", ".bias")]
    dx, grads = O.causal_attention_layer_grads(x, *args, heads=8, dout=G_, mask_flag=mask)
    names = [n + s for n in _ATTN for s in (".weight", ".bias")]
    for k, got in [("x", dx)] + list(zip(names, grads)):
        ref_n, got_n = float(g["norm." + k]), float(np.linalg.norm(got))
        if k == "key_projection.bias":            # exactly zero in real arithmetic (a shift of every key moves a row's scores alike)
            assert got_n < 1e-6 and ref_n < 1e-4
            continue
        assert abs(got_n - ref_n) < 2e-5 * ref_n, (k, got_n, ref_n)
        assert np.abs(got.reshape(-1)[g["idx." + k]] - g["val." + k]).max() < 2e-5 * max(1.0, np.abs(g["val." + k]).max()), k


def test_g7_evaluate_json_and_argmax():
    g = json.load(open(os.path.join(G, "g7_evaluate.json")))
    cfg = epic_tent_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    for i, T in enumerate(g["lens"]):
        rgb = W.tsn_features((T, 2048), 20, f"g7.rgb.{i}")
        probs = O.miniroad_forward(sd, rgb[None], None)["logits"][0]
        seg = (np.arange(T) // 37) % 12
        tgt = np.zeros((T, 12), np.float32)
        tgt[np.arange(T), seg] = 1.0
        pred, gt = O.eval_argmax(probs, tgt)
        ref = g["output"][f"synth_video_{i}"]
        srt = np.sort(probs, 1)
        safe = (srt[:, -1] - srt[:, -2]) > 1e-5
        assert np.array_equal(pred[safe], np.array(ref["pred"])[safe])
        assert gt.tolist() == ref["gt"]


def test_g8_aggregate_known_answer():
    with gzip.open(os.path.join(G, "g8_output_miniROAD.json.gz"), "rt") as f:
        data = json.load(f)
    want = json.load(open(os.path.join(G, "g8_aggregated_data.json")))
    assert json.load(open(os.path.join(G, "g8_meta.json")))["shipped_equals_rerun"]
    assert O.aggregate(data) == want


def test_torch_port_matches_golden():
    """bench.py's cpu_baseline ("port") computes what the reference computes."""
    import torch
    from oracle.oracle_torch import TorchPort
    g = _ld("g1_miniroad_eval_peaky.npz")
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    port = TorchPort(sd, 1024)
    rgb = torch.from_numpy(W.tsn_features((1, 256, 2048), 20, "g1.rgb"))
    out = port.forward(rgb, torch.zeros_like(rgb))[0].numpy()
    assert np.abs(out - g["probs"]).max() < 1e-6
    assert np.array_equal(out.argmax(1), g["argmax"])


def test_g4c_train_16x128_real_shape():
    """the oracle's hand-written BPTT at the real training shape (16 windows x 128 frames) against the reference"""
    g = _ld("g4c_miniroad_train_16x128.npz")
    cfg = assembly101_cfg(dropout=0.0)
    sd = W.miniroad_state_dict(cfg, 20)
    rgb = W.tsn_features((16, 128, 2048), 20, "g4c.rgb")
    flow = W.tsn_features((16, 128, 2048), 20, "g4c.flow")
    tgt = _targets(16, 128, 86, 20, "g4c.tgt")
    loss, grads = O.miniroad_loss_and_grads(sd, rgb, flow, tgt)
    assert abs(loss - float(g["loss"])) < 5e-6
    for k in sd:
        gg = grads[k].reshape(-1)
        assert abs(np.linalg.norm(gg) - float(g["norm." + k])) < 2e-4 * float(g["norm." + k]) + 1e-9, k
        ref = g["val." + k]
        assert np.abs(gg[g["idx." + k]] - ref).max() < 1e-7 + 2e-3 * np.abs(ref).max(), k


@pytest.mark.parametrize("layers", [1, 2])
def test_g5b_vit_training_step(layers):
    """ViTEnc forward (1 and 2 layers) + OadLoss + full backward by hand against the reference's autograd"""
    g = _ld(f"g5b_vit_train_L{layers}.npz")
    cfg = dict(_vit_cfg(), num_layers=layers)
    sd = W.vit_state_dict(cfg, 20)
    rgb = W.tsn_features((2, 128, 2048), 20, "g5.rgb")
    flow = W.tsn_features((2, 128, 2048), 20, "g5.flow")
    tgt = _targets(2, 128, 86, 20, "g5b.tgt")
    loss, logits, grads = O.vit_loss_and_grads(sd, rgb, flow, tgt, heads=8, num_layers=layers)
    assert np.abs(logits - g["logits"]).max() < 3e-5
    assert abs(loss - float(g["loss"])) < 5e-6
    keys = [k[5:] for k in g.files if k.startswith("norm.")]
    assert set(keys) == set(k for k in sd if k != "position_encoding.position_ids")
    for k in keys:
        gg = grads[k].reshape(-1)
        ref_norm = float(g["norm." + k])
        assert abs(np.linalg.norm(gg) - ref_norm) < 5e-4 * ref_norm + 1e-9, (k, np.linalg.norm(gg), ref_norm)
        ref = g["val." + k]
        assert np.abs(gg[g["idx." + k]] - ref).max() < 1e-7 + 5e-3 * np.abs(ref).max(), k


@pytest.mark.parametrize("tag", ["a101", "epic"])
def test_g11_trained_weights_oracle_vs_reference_evaluate(tag):
    """G11 (round 6): weights that went through the reference's own training loop (oracle/train_g11.py), expected outputs from the
    reference's own Evaluate.  The oracle on the two short videos: probabilities at the sampled frames and every argmax above a 1e-5
    margin (fp64 oracle vs torch fp32)."""
    from prego_amd import workloads as WL
    g = _ld(f"g11_eval_{tag}.npz")
    cfg = {"a101": assembly101_cfg, "epic": epic_tent_cfg}[tag]()
    sd = W.g11_state_dict(tag)
    z = _ld(f"g11_weights_{tag}.npz")
    assert int(z["steps"]) >= 1000 and float(z["loss_curve"][-50:].mean()) < 0.5 * float(z["loss_curve"][:10].mean())     # it did train
    for i in (0, 1):
        T = int(g["lengths"][i])
        rgb, lab = WL.action_video(T, cfg["num_classes"], 20, f"g11.{tag}.eval.{i}")
        assert np.array_equal(lab, g[f"gt{i}"].astype(np.int64))
        out = O.miniroad_forward(sd, rgb[None], None)["logits"][0]
        assert np.abs(out[g[f"sample_idx{i}"]] - g[f"sample_probs{i}"]).max() < 2e-5
        safe = g[f"margin{i}"] > 1e-5
        assert np.array_equal(out.argmax(1)[safe], g[f"pred{i}"].astype(np.int64)[safe])
