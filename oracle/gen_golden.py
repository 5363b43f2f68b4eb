#!/usr/bin/env python3
"""Generate tests/golden/* by importing the REAL reference from /root/reference.

Runs only in the build container (the GPU box has no /root/reference).  Only
data is written: inputs are regenerated from seeds by prego_amd/weights.py, so
the fixtures hold expected OUTPUTS (plus tiny inputs where convenient).  No
reference source text is copied anywhere.

Import needs two stubs: `torchvision` (star-imported by
utils/group_transforms.py, unused on the path) and `ipdb` (datasets/dataset.py:5).

    python oracle/gen_golden.py            # everything (~2-3 min, one long-T case)
    python oracle/gen_golden.py g1 g4      # selected groups
"""
from __future__ import annotations

import gzip
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/step_recognition"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from prego_amd import weights as W            # noqa: E402
from prego_amd.config import assembly101_cfg, epic_tent_cfg  # noqa: E402


def _stub_modules():
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "ipdb"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    sys.modules["ipdb"].set_trace = lambda *a, **k: None
    sys.path.insert(0, REF)


def _load(model, sd):
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    return model


def g1_g2_g3():
    """MROAD eval: cfg1 (B=1,T=256), long-T drift, T=8 intermediates."""
    from model import build_model
    cfg = assembly101_cfg()
    for tag, gain in (("plain", 1.0), ("peaky", 8.0)):
        sd = W.miniroad_state_dict(cfg, seed=20, head_gain=gain)
        model = _load(build_model(cfg, "cpu"), sd).eval()
        rgb = W.tsn_features((1, 256, 2048), 20, "g1.rgb")
        flow = np.zeros_like(rgb)
        with torch.no_grad():
            out = model(torch.from_numpy(rgb), torch.from_numpy(flow))["logits"][0].numpy()
            # nn.GRU called directly for the final hidden state
            x = model.layer1(torch.cat((torch.from_numpy(rgb), torch.from_numpy(flow)), 2))
            hs, hl = model.gru(x, torch.zeros(1, 1, 1024))
        np.savez_compressed(os.path.join(OUT, f"g1_miniroad_eval_{tag}.npz"),
                            probs=out.astype(np.float32), argmax=out.argmax(1).astype(np.int32),
                            h_last=hl[0, 0].numpy().astype(np.float32), head_gain=np.float32(gain))
        print("g1", tag, out.shape, "min top1-top2 margin", float(np.min(np.sort(out, 1)[:, -1] - np.sort(out, 1)[:, -2])))
        if tag == "peaky":
            # G2: long-T drift with non-zero flow on the short one, zero flow on the long one
            for T in (4096, 31114):
                rgbL = W.tsn_features((1, T, 2048), 20, f"g2.rgb.{T}")
                flowL = W.tsn_features((1, T, 2048), 20, f"g2.flow.{T}") if T == 4096 else np.zeros_like(rgbL)
                with torch.no_grad():
                    o = model(torch.from_numpy(rgbL), torch.from_numpy(flowL))["logits"][0].numpy()
                idx = np.linspace(0, T - 1, 64).astype(np.int64)
                srt = np.sort(o, 1)
                np.savez_compressed(os.path.join(OUT, f"g2_miniroad_longT_{T}.npz"),
                                    argmax=o.argmax(1).astype(np.int16), margin=(srt[:, -1] - srt[:, -2]).astype(np.float32),
                                    sample_idx=idx, sample_probs=o[idx].astype(np.float32))
                print("g2", T, "done")
        if tag == "plain":
            # G3: intermediates at T=8, non-zero flow
            rgb8 = W.tsn_features((1, 8, 2048), 20, "g3.rgb")
            flow8 = W.tsn_features((1, 8, 2048), 20, "g3.flow")
            with torch.no_grad():
                x = torch.cat((torch.from_numpy(rgb8), torch.from_numpy(flow8)), 2)
                y = model.layer1[0](x)
                e = model.layer1(x)
                hs, _ = model.gru(e, torch.zeros(1, 1, 1024))
                model.train()
                p = model.dropout_p = None
                model.layer1[3].p = 0.0
                raw = model(torch.from_numpy(rgb8), torch.from_numpy(flow8))["logits"]
                model.eval()
                probs = model(torch.from_numpy(rgb8), torch.from_numpy(flow8))["logits"]
            np.savez_compressed(os.path.join(OUT, "g3_miniroad_intermediates.npz"),
                                y=y[0].numpy(), e=e[0].numpy(), h=hs[0].numpy(), raw_logits=raw[0].numpy(), probs=probs[0].numpy())
            print("g3 done")


def g1c():
    """MROAD eval with a trained-like head (f_classification.0.weight x 32: top-1 probabilities near 1, the regime in which a
    reduced-precision classifier flips argmaxes - round-3 verdict item 2): 1 clip x 1024 frames, rgb + non-zero flow."""
    from model import build_model
    cfg = assembly101_cfg()
    sd = W.miniroad_state_dict(cfg, seed=20, head_gain=32.0)
    model = _load(build_model(cfg, "cpu"), sd).eval()
    T = 1024
    rgb = W.tsn_features((1, T, 2048), 20, "g1c.rgb")
    flow = W.tsn_features((1, T, 2048), 20, "g1c.flow")
    with torch.no_grad():
        out = model(torch.from_numpy(rgb), torch.from_numpy(flow))["logits"][0].numpy()
    srt = np.sort(out, 1)
    np.savez_compressed(os.path.join(OUT, "g1c_miniroad_eval_gain32.npz"), probs=out.astype(np.float32),
                        argmax=out.argmax(1).astype(np.int32), margin=(srt[:, -1] - srt[:, -2]).astype(np.float32),
                        head_gain=np.float32(32.0))
    print("g1c", out.shape, "median top-1", float(np.median(srt[:, -1])), "frames with margin < 1e-3:", int((srt[:, -1] - srt[:, -2] < 1e-3).sum()))


def _small_cfg():
    # smallest feature size the reference's FEATURE_SIZES table offers is 1024 (rnn.py:6-16)
    return assembly101_cfg(rgb_type="rgb_kinetics_bninception", no_flow=True, embedding_dim=128,
                           hidden_dim=64, num_classes=12, dropout=0.0, window_size=16, batch_size=4)


def make_targets(B, T, C, seed, name, pad_rows=0):
    """one-hot per frame; first `pad_rows` rows all-zero (dataset.py:53-55,77-82)."""
    cls = (W.uniform01((B, T), seed, name) * C).astype(np.int64)
    tgt = np.zeros((B, T, C), dtype=np.float32)
    bi, ti = np.meshgrid(np.arange(B), np.arange(T), indexing="ij")
    tgt[bi, ti, cls] = 1.0
    if pad_rows:
        tgt[:, :pad_rows] = 0.0
    return tgt


def g4():
    """loss + grads + AdamW, via reference build_criterion / train semantics (train.py:20-24, main.py:62-67)."""
    from model import build_model
    from criterions import build_criterion
    # (a) reduced dims, all tensors stored
    cfg = _small_cfg()
    sd = W.miniroad_state_dict(cfg, seed=20)
    model = _load(build_model(cfg, "cpu"), sd).train()
    crit = build_criterion(cfg, "cpu")
    B, T = 4, 16
    rgb = W.tsn_features((B, T, 1024), 20, "g4.rgb")
    tgt = make_targets(B, T, 12, 20, "g4.tgt")
    tgt[3, -1] = 0.0                      # one all-zero last-frame target row (front padding case)
    tgt[2, -1, 5] = 1.0                   # one multi-label row (normalisation by L2 norm)
    optim = torch.optim.AdamW([{"params": model.parameters(), "initial_lr": cfg["lr"]}], lr=cfg["lr"],
                              weight_decay=cfg["weight_decay"])
    save = {}
    losses = []
    for step in range(3):
        out = model(torch.from_numpy(rgb), torch.from_numpy(np.zeros((B, T, 0), np.float32)))
        loss = crit(out, torch.from_numpy(tgt))
        optim.zero_grad(set_to_none=True)
        loss.backward()
        if step == 0:
            save["logits0"] = out["logits"].detach().numpy().copy()
            for k, p in model.named_parameters():
                save["grad." + k] = p.grad.numpy().copy()
        optim.step()
        losses.append(float(loss))
        if step in (0, 2):
            for k, p in model.named_parameters():
                save[f"param{step + 1}." + k] = p.detach().numpy().copy()
    save["losses"] = np.array(losses, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "g4_miniroad_train_small.npz"), **save)
    print("g4 small losses", losses)
    # (b) full dims with non-zero flow, B=2,T=8: loss + per-tensor grad norms + sampled entries
    cfg = assembly101_cfg(dropout=0.0)
    sd = W.miniroad_state_dict(cfg, seed=20)
    model = _load(build_model(cfg, "cpu"), sd).train()
    crit = build_criterion(cfg, "cpu")
    rgb = W.tsn_features((2, 8, 2048), 20, "g4b.rgb")
    flow = W.tsn_features((2, 8, 2048), 20, "g4b.flow")
    tgt = make_targets(2, 8, 86, 20, "g4b.tgt")
    out = model(torch.from_numpy(rgb), torch.from_numpy(flow))
    loss = crit(out, torch.from_numpy(tgt))
    loss.backward()
    save = {"loss": np.float64(float(loss))}
    for k, p in model.named_parameters():
        g = p.grad.numpy().reshape(-1)
        idx = np.linspace(0, g.size - 1, 256).astype(np.int64)
        save["norm." + k] = np.float64(np.linalg.norm(g.astype(np.float64)))
        save["idx." + k] = idx
        save["val." + k] = g[idx].copy()
    np.savez_compressed(os.path.join(OUT, "g4b_miniroad_train_full.npz"), **save)
    print("g4b full loss", float(loss))


def _vit_cfg(window=128, classes=86):
    return assembly101_cfg(model="Transformer", window_size=window, patch_dim=1, num_heads=8,
                           attn_dropout_rate=0.0, dropout=0.0, num_classes=classes)


def g5():
    """ViTEnc forward + sub-module outputs via forward hooks."""
    from model import build_model
    cfg = _vit_cfg()
    sd = W.vit_state_dict(cfg, seed=20)
    model = _load(build_model(cfg, "cpu"), sd).eval()
    B, T = 2, 128
    rgb = W.tsn_features((B, T, 2048), 20, "g5.rgb")
    flow = W.tsn_features((B, T, 2048), 20, "g5.flow")
    cap = {}
    hooks = [
        model.encoder.net[0].fn.norm.register_forward_hook(lambda m, i, o: cap.__setitem__("ln1", o.detach().numpy().copy())),
        model.encoder.net[0].fn.fn.register_forward_hook(lambda m, i, o: cap.__setitem__("attn", o.detach().numpy().copy())),
        model.encoder.net[1].fn.norm.register_forward_hook(lambda m, i, o: cap.__setitem__("ln2", o.detach().numpy().copy())),
        model.encoder.net[1].fn.fn.register_forward_hook(lambda m, i, o: cap.__setitem__("ffn", o.detach().numpy().copy())),
    ]
    with torch.no_grad():
        out = model(torch.from_numpy(rgb), torch.from_numpy(flow))["logits"].numpy()
    for h in hooks:
        h.remove()
    rows = np.array([0, 1, 63, 127, 128])
    np.savez_compressed(os.path.join(OUT, "g5_vit_forward.npz"), logits=out, rows=rows,
                        **{k: v[:, rows] for k, v in cap.items()})
    print("g5 logits", out.shape, float(np.abs(out).max()))


def g5c():
    """ViTEnc at the LONG window of BASELINE configs[3] (Epic-tent-O: window 1024 -> N = 1025 tokens, 12 classes, 8 heads of 256):
    logits of 2 windows + rows of the first block's LayerNorm / attention / FFN outputs from the reference itself."""
    from model import build_model
    cfg = _vit_cfg(window=1024, classes=12)
    sd = W.vit_state_dict(cfg, seed=20)
    model = _load(build_model(cfg, "cpu"), sd).eval()
    B, T = 2, 1024
    rgb = W.tsn_features((B, T, 2048), 20, "g5c.rgb")
    flow = W.tsn_features((B, T, 2048), 20, "g5c.flow")
    cap = {}
    hooks = [
        model.encoder.net[0].fn.fn.register_forward_hook(lambda m, i, o: cap.__setitem__("attn", o.detach().numpy().copy())),
        model.encoder.net[1].fn.fn.register_forward_hook(lambda m, i, o: cap.__setitem__("ffn", o.detach().numpy().copy())),
    ]
    with torch.no_grad():
        out = model(torch.from_numpy(rgb), torch.from_numpy(flow))["logits"].numpy()
    for h in hooks:
        h.remove()
    rows = np.array([0, 1, 511, 1023, 1024])
    np.savez_compressed(os.path.join(OUT, "g5c_vit_forward_w1024.npz"), logits=out, rows=rows, **{k: v[:, rows] for k, v in cap.items()})
    print("g5c logits", out.shape, float(np.abs(out).max()))


def g6():
    """Causal attention: AttentionLayer(FullAttention(mask_flag=True, attention_dropout=0)) - dead code in the
    reference (attn.py:35-57,139-170) but the only causal definition."""
    from model.transformer_models.attn import AttentionLayer, FullAttention
    d, H = 2048, 8
    sd = W.attention_layer_state_dict(d, seed=20)
    layer = AttentionLayer(FullAttention(mask_flag=True, attention_dropout=0.0), d, H).eval()
    _load(layer, sd)
    for L in (128, 1024):
        x = W.normal((1, L, d), 20, f"g6.x.{L}")
        xt = torch.from_numpy(x)
        with torch.no_grad():
            o = layer(xt, xt, xt, None)[0].numpy()
            # causality: perturb the last frame, outputs 0..L-2 must be bit-identical
            x2 = x.copy()
            x2[0, -1] += 1.0
            o2 = layer(torch.from_numpy(x2), torch.from_numpy(x2), torch.from_numpy(x2), None)[0].numpy()
        assert np.array_equal(o[:-1], o2[:-1])
        rows = np.unique(np.concatenate([np.arange(0, 4), np.linspace(0, L - 1, 28).astype(np.int64)]))
        np.savez_compressed(os.path.join(OUT, f"g6_causal_attention_L{L}.npz"), rows=rows, out=o[rows])
        print("g6", L, "done")


def g6b():
    """The same layer under autograd (attn.py:151-170 inside a backward pass): loss = sum(layer(x, x, x) * G) for a seeded G,
    gradients of x and of the eight projection parameters: norms + sampled entries, causal and unmasked, two head dims."""
    from model.transformer_models.attn import AttentionLayer, FullAttention
    for d, H, B, L, mask in ((512, 8, 2, 192, True), (2048, 8, 1, 128, True), (1024, 8, 2, 100, False)):
        sd = W.attention_layer_state_dict(d, seed=20)
        layer = AttentionLayer(FullAttention(mask_flag=mask, attention_dropout=0.0), d, H).eval()
        _load(layer, sd)
        x = torch.from_numpy(W.normal((B, L, d), 20, f"g6b.x.{d}.{L}")).requires_grad_(True)
        G = torch.from_numpy(W.normal((B, L, d), 20, f"g6b.g.{d}.{L}"))
        out = layer(x, x, x, None)
        (out * G).sum().backward()
        save = {"out_norm": np.float64(np.linalg.norm(out.detach().numpy().astype(np.float64)))}
        _grad_summary(layer, save)
        g = x.grad.numpy().reshape(-1)
        idx = np.linspace(0, g.size - 1, 512).astype(np.int64)
        save.update({"norm.x": np.float64(np.linalg.norm(g.astype(np.float64))), "idx.x": idx, "val.x": g[idx].copy()})
        np.savez_compressed(os.path.join(OUT, f"g6b_attention_grads_d{d}_L{L}_{'causal' if mask else 'full'}.npz"), **save)
        print("g6b", d, L, mask, {k: float(v) for k, v in save.items() if k.startswith("norm.")})


class _SynthEval(torch.utils.data.Dataset):
    def __init__(self, lens, C, seed):
        self.items = []
        for i, T in enumerate(lens):
            rgb = W.tsn_features((T, 2048), seed, f"g7.rgb.{i}")
            tgt = make_targets(1, T, C, seed, f"g7.tgt.{i}")[0]
            # piecewise-constant labels, like real step annotations
            seg = (np.arange(T) // 37) % C
            tgt = np.zeros((T, C), np.float32)
            tgt[np.arange(T), seg] = 1.0
            self.items.append((rgb, np.zeros_like(rgb), tgt, f"synth_video_{i}", 0, T))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        r, f, t, vid, s, e = self.items[i]
        return torch.from_numpy(r), torch.from_numpy(f), torch.from_numpy(t), vid, s, e


def g7():
    """Evaluate end to end (trainer/eval.py:30-84) on 3 synthetic Epic-tent-O-shaped videos."""
    import logging
    from model import build_model
    from trainer import build_eval
    cfg = epic_tent_cfg(eval="dummy.pth", video_list_path=os.path.join(REF, "data_info", "video_list.json"))
    sd = W.miniroad_state_dict(cfg, seed=20, head_gain=8.0)
    model = _load(build_model(cfg, "cpu"), sd)
    ev = build_eval(cfg)
    loader = torch.utils.data.DataLoader(_SynthEval([300, 517, 190], 12, 20), batch_size=1, shuffle=False)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        try:
            mAP = ev(model, loader, logging.getLogger("g7"), "cpu")
        except AttributeError:
            # eval.py:77 `(end - start).item()`: `start` is shadowed by the loader field;
            # with python ints it raises instead of printing nonsense.  mAP is computed before that line, so
            # recompute it the way eval.py:70-76 does.
            mAP = None
        js = json.load(open(os.path.join(tmp, "output_miniRoad", "output_miniROAD.json")))
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
    if mAP is None:
        from utils import perframe_average_precision
        names = json.load(open(cfg["video_list_path"]))["EPIC-TENT-O"]["class_index"]
        preds, gts = [], []
        model.eval()
        with torch.no_grad():
            for rgb, flow, tgt, vid, s, e in loader:
                preds += list(model(rgb, flow)["logits"].squeeze().numpy())
                gts += list(tgt.squeeze().numpy())
        mAP = perframe_average_precision(preds, gts, names, None, "AP")["mean_AP"]
    with open(os.path.join(OUT, "g7_evaluate.json"), "w") as f:
        json.dump({"mAP": float(mAP), "output": js, "lens": [300, 517, 190]}, f)
    print("g7 mAP", float(mAP))


def g8():
    """aggregate.py known-answer pair shipped by the reference: output_miniRoad/output_miniROAD.json ->
    data/output/aggregated_data.json.  Data files, stored gzipped; also re-run the reference aggregate() to
    confirm the shipped pair really is in/out of the shipped code."""
    src = "/root/reference/output_miniRoad/output_miniROAD.json"
    dst = "/root/reference/data/output/aggregated_data.json"
    sys.path.insert(0, "/root/reference/utils")
    import aggregate as ref_agg
    tmp = tempfile.mktemp(suffix=".json")
    ref_agg.aggregate(json.load(open(src)), tmp)
    rerun = json.load(open(tmp))
    os.remove(tmp)
    shipped = json.load(open(dst))
    print("g8 shipped pair reproduced by reference aggregate():", rerun == shipped)
    with gzip.open(os.path.join(OUT, "g8_output_miniROAD.json.gz"), "wt") as f:
        json.dump(json.load(open(src)), f, separators=(",", ":"))
    with open(os.path.join(OUT, "g8_aggregated_data.json"), "w") as f:
        json.dump(rerun, f, separators=(",", ":"))
    with open(os.path.join(OUT, "g8_meta.json"), "w") as f:
        json.dump({"shipped_equals_rerun": rerun == shipped}, f)


def _grad_summary(model, save, n=256):
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.numpy().reshape(-1)
        idx = np.linspace(0, g.size - 1, min(n, g.size)).astype(np.int64)
        save["norm." + k] = np.float64(np.linalg.norm(g.astype(np.float64)))
        save["idx." + k] = idx
        save["val." + k] = g[idx].copy()


def g4c():
    """Training step at the REAL shape of configs/miniroad_assembly101-O.yaml: B = 16 windows x T = 128 frames, full dims,
    rgb + non-zero flow, dropout 0 (torch's mask stream cannot be matched): loss + per-tensor grad norms + 256 sampled
    entries per tensor (trainer/train.py:20-23, 128 BPTT steps)."""
    from model import build_model
    from criterions import build_criterion
    cfg = assembly101_cfg(dropout=0.0)
    sd = W.miniroad_state_dict(cfg, seed=20)
    model = _load(build_model(cfg, "cpu"), sd).train()
    crit = build_criterion(cfg, "cpu")
    B, T = 16, 128
    rgb = W.tsn_features((B, T, 2048), 20, "g4c.rgb")
    flow = W.tsn_features((B, T, 2048), 20, "g4c.flow")
    tgt = make_targets(B, T, 86, 20, "g4c.tgt")
    out = model(torch.from_numpy(rgb), torch.from_numpy(flow))
    loss = crit(out, torch.from_numpy(tgt))
    loss.backward()
    save = {"loss": np.float64(float(loss)), "logits_last": out["logits"][:, -1, :].detach().numpy().copy()}
    _grad_summary(model, save)
    np.savez_compressed(os.path.join(OUT, "g4c_miniroad_train_16x128.npz"), **save)
    print("g4c loss", float(loss))


def g5b():
    """ViTEnc with num_layers = 2 (every row of layer 1 reaches the logits through layer 2's attention) and the TRAINING step of
    the `Transformer` registry entry: OadLoss on the [B,1,C] logits (criterions/loss.py:15-21), loss.backward(): loss +
    per-tensor grad norms + sampled entries, for 1 and 2 layers, window 128, B = 2, all dropouts 0."""
    from model import build_model
    from criterions import build_criterion
    B, T = 2, 128
    rgb = W.tsn_features((B, T, 2048), 20, "g5.rgb")
    flow = W.tsn_features((B, T, 2048), 20, "g5.flow")
    tgt = make_targets(B, T, 86, 20, "g5b.tgt")
    for layers in (1, 2):
        cfg = dict(_vit_cfg(), num_layers=layers)
        sd = W.vit_state_dict(cfg, seed=20)
        model = _load(build_model(cfg, "cpu"), sd).eval()
        with torch.no_grad():
            logits = model(torch.from_numpy(rgb), torch.from_numpy(flow))["logits"].numpy()
        model.train()
        crit = build_criterion(cfg, "cpu")
        out = model(torch.from_numpy(rgb), torch.from_numpy(flow))
        loss = crit(out, torch.from_numpy(tgt))
        loss.backward()
        save = {"logits": logits, "loss": np.float64(float(loss))}
        _grad_summary(model, save)
        np.savez_compressed(os.path.join(OUT, f"g5b_vit_train_L{layers}.npz"), **save)
        print("g5b layers", layers, "loss", float(loss), "logits", float(np.abs(logits).max()))


def g9():
    """Feeder (datasets/dataset.py:24-135) on a synthetic 2-video Epic-tent-O-shaped tree written to a temp dir: the window
    list of train mode (front pad window_size-1, stride 4, np.random phase), the whole-video items of test mode, and two
    sample items per mode.  The tree itself is regenerated from seeds by tests (prego_amd.weights), only outputs are stored."""
    from datasets import build_data_loader  # noqa: F401  (registers)
    from datasets.dataset_builder import DATA_LAYERS
    tmp = tempfile.mkdtemp()
    try:
        lens = {"vidA": 300, "vidB": 157}
        C = 12
        for sub in ("target_perframe", "rgb_anet_resnet50", "rgb_as_flow/rgb_anet_resnet50"):
            os.makedirs(os.path.join(tmp, sub))
        for vid, T in lens.items():
            rgb = W.tsn_features((T, 2048), 20, f"g9.rgb.{vid}")
            tgt = np.zeros((T, C), np.float32)
            tgt[np.arange(T), (np.arange(T) // 29) % C] = 1.0
            np.save(os.path.join(tmp, "rgb_anet_resnet50", vid + ".npy"), rgb)
            np.save(os.path.join(tmp, "rgb_as_flow/rgb_anet_resnet50", vid + ".npy"), rgb)
            np.save(os.path.join(tmp, "target_perframe", vid + ".npy"), tgt)
        vl = os.path.join(tmp, "video_list.json")
        json.dump({"EPIC-TENT-O": {"train_session_set": ["vidA", "vidB"], "test_session_set": ["vidB", "vidA"]}}, open(vl, "w"))
        cfg = epic_tent_cfg(root_path=tmp, video_list_path=vl)
        save = {}
        for mode in ("train", "test"):
            np.random.seed(20)
            ds = DATA_LAYERS[cfg["data_name"]](cfg, mode)
            wins = [(it[0], int(it[1]), int(it[2])) for it in ds.inputs]
            save[f"{mode}.vids"] = np.array([w[0] for w in wins])
            save[f"{mode}.start"] = np.array([w[1] for w in wins], np.int64)
            save[f"{mode}.end"] = np.array([w[2] for w in wins], np.int64)
            for j in (0, len(ds) - 1):
                r, f, t, vid, s_, e_ = ds[j]
                save[f"{mode}.item{j}.rgb_sum"] = np.float64(r.double().sum().item())
                save[f"{mode}.item{j}.rgb_rows"] = r.numpy()[[0, -1]][:, :16].copy()
                save[f"{mode}.item{j}.flow_abs_sum"] = np.float64(f.double().abs().sum().item())
                save[f"{mode}.item{j}.target_argmax"] = t.numpy().argmax(1).astype(np.int64)
                save[f"{mode}.item{j}.target_rowsum"] = t.numpy().sum(1)
                save[f"{mode}.item{j}.meta"] = np.array([str(vid), str(int(s_)), str(int(e_)), str(tuple(r.shape)), str(r.dtype), str(f.dtype), str(t.dtype)])
            save[f"{mode}.len"] = np.int64(len(ds))
        np.savez_compressed(os.path.join(OUT, "g9_feeder.npz"), **save)
        print("g9 train windows", int(save["train.len"]), "test items", int(save["test.len"]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def g4s():
    """OadLoss(cfg, reduction='sum') (criterions/loss.py:8-11,30-33) and its gradient, beside 'mean', on one batch of logits: 6 windows x
    5 frames x 86 classes, one all-zero last-frame target row (front padding) and one multi-label row."""
    from criterions.loss import OadLoss
    cfg = assembly101_cfg()
    B, T, C = 6, 5, cfg["num_classes"]
    logits = (W.uniform01((B, T, C), 20, "g4s.logits") * 8.0 - 4.0).astype(np.float32)
    tgt = make_targets(B, T, C, 20, "g4s.tgt")
    tgt[4, -1] = 0.0
    tgt[1, -1, 7] = 1.0
    save = {"logits": logits, "target": tgt}
    for red in ("mean", "sum"):
        lg = torch.from_numpy(logits).clone().requires_grad_(True)
        loss = OadLoss(cfg, reduction=red)({"logits": lg}, torch.from_numpy(tgt))
        loss.backward()
        save[f"loss_{red}"] = np.float64(loss.item())
        save[f"dlogits_{red}"] = lg.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g4s_oadloss_sum.npz"), **save)
    print("g4s", float(save["loss_mean"]), float(save["loss_sum"]))


def g10():
    """MROAD eval at the dimensions the shipped yamls do not use (rnn.py:31-38 takes any): hidden_dim 512 / 2048 and num_layers 2.
    Full feature and embedding sizes (2048 + 2048 -> 2048), 86 classes, head gain 8; two clips of 96 and 40 frames with non-zero flow;
    probabilities, argmax and the final state h_n [layers, H] of each clip."""
    from model import build_model
    for tag, hid, layers in (("h512", 512, 1), ("h2048", 2048, 1), ("h1024_l2", 1024, 2), ("h512_l2", 512, 2)):
        cfg = assembly101_cfg(hidden_dim=hid, num_layers=layers)
        sd = W.miniroad_state_dict(cfg, seed=20, head_gain=8.0)
        model = _load(build_model(cfg, "cpu"), sd).eval()
        save = {}
        for i, T in enumerate((96, 40)):
            rgb = W.tsn_features((1, T, 2048), 20, f"g10.{tag}.rgb.{i}")
            flow = W.tsn_features((1, T, 2048), 20, f"g10.{tag}.flow.{i}")
            with torch.no_grad():
                out = model(torch.from_numpy(rgb), torch.from_numpy(flow))["logits"][0].numpy()
                x = model.layer1(torch.cat((torch.from_numpy(rgb), torch.from_numpy(flow)), 2))
                _, hn = model.gru(x, torch.zeros(layers, 1, hid))
            srt = np.sort(out, 1)
            save[f"probs{i}"] = out.astype(np.float32)
            save[f"argmax{i}"] = out.argmax(1).astype(np.int32)
            save[f"margin{i}"] = (srt[:, -1] - srt[:, -2]).astype(np.float32)
            save[f"h_n{i}"] = hn[:, 0].numpy().astype(np.float32)
        np.savez_compressed(os.path.join(OUT, f"g10_miniroad_eval_{tag}.npz"), **save)
        print("g10", tag, "min margin", float(min(save["margin0"].min(), save["margin1"].min())))


def g4d():
    """Training step at the hidden sizes the shipped yamls do not use (rnn.py:31-38 takes any hidden_dim; round-5 verdict "missing 2"):
    hidden_dim 512 and 2048, full feature / embedding sizes, B = 5 windows x T = 24 frames (not a tile multiple), rgb + non-zero flow,
    one all-zero and one multi-label last-frame target row, dropout 0: loss, last-frame logits, per-tensor gradient norms + sampled
    entries (trainer/train.py:20-23)."""
    from model import build_model
    from criterions import build_criterion
    B, T = 5, 24
    for hid in (512, 2048):
        cfg = assembly101_cfg(dropout=0.0, hidden_dim=hid)
        sd = W.miniroad_state_dict(cfg, seed=20)
        model = _load(build_model(cfg, "cpu"), sd).train()
        crit = build_criterion(cfg, "cpu")
        rgb = W.tsn_features((B, T, 2048), 20, f"g4d.{hid}.rgb")
        flow = W.tsn_features((B, T, 2048), 20, f"g4d.{hid}.flow")
        tgt = make_targets(B, T, 86, 20, f"g4d.{hid}.tgt")
        tgt[1, -1] = 0.0
        tgt[2, -1, 7] = 1.0
        out = model(torch.from_numpy(rgb), torch.from_numpy(flow))
        loss = crit(out, torch.from_numpy(tgt))
        loss.backward()
        save = {"loss": np.float64(float(loss)), "logits_last": out["logits"][:, -1, :].detach().numpy().copy()}
        _grad_summary(model, save)
        np.savez_compressed(os.path.join(OUT, f"g4d_miniroad_train_h{hid}.npz"), **save)
        print("g4d", hid, "loss", float(loss))


def g4e():
    """Training step of a STACKED GRU (cfg['num_layers'] = 2: nn.GRU(2048, H, 2), rnn.py:32,38 - trainer/train.py trains whatever the
    constructor accepts): hidden_dim 1024 and 512, B = 5 windows x T = 24 frames, rgb + non-zero flow, an all-zero and a multi-label
    last-frame target row, dropout 0: loss, last-frame logits, norms + sampled entries of all 14 gradients."""
    from model import build_model
    from criterions import build_criterion
    B, T = 5, 24
    for hid in (1024, 512):
        cfg = assembly101_cfg(dropout=0.0, hidden_dim=hid, num_layers=2)
        sd = W.miniroad_state_dict(cfg, seed=20)
        model = _load(build_model(cfg, "cpu"), sd).train()
        crit = build_criterion(cfg, "cpu")
        rgb = W.tsn_features((B, T, 2048), 20, f"g4e.{hid}.rgb")
        flow = W.tsn_features((B, T, 2048), 20, f"g4e.{hid}.flow")
        tgt = make_targets(B, T, 86, 20, f"g4e.{hid}.tgt")
        tgt[1, -1] = 0.0
        tgt[2, -1, 7] = 1.0
        out = model(torch.from_numpy(rgb), torch.from_numpy(flow))
        loss = crit(out, torch.from_numpy(tgt))
        loss.backward()
        save = {"loss": np.float64(float(loss)), "logits_last": out["logits"][:, -1, :].detach().numpy().copy()}
        _grad_summary(model, save)
        assert "norm.gru.weight_ih_l1" in save
        np.savez_compressed(os.path.join(OUT, f"g4e_miniroad_train_l2_h{hid}.npz"), **save)
        print("g4e", hid, "loss", float(loss))


GROUPS = {"g4e": g4e, "g4d": g4d, "g10": g10, "g4s": g4s, "g1": g1_g2_g3, "g1c": g1c, "g4": g4, "g4c": g4c, "g5": g5, "g5b": g5b, "g5c": g5c, "g6": g6, "g6b": g6b, "g7": g7, "g8": g8, "g9": g9}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    _stub_modules()
    torch.manual_seed(20)
    torch.set_num_threads(8)
    which = sys.argv[1:] or list(GROUPS)
    for k in which:
        GROUPS[k]()
