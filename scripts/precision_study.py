"""Which stage of the MiniROAD eval path flips argmaxes under reduced-precision MFMA operands?  (round-3 verdict item 2)

CPU emulation (torch CPU fp32 matmuls on operands rounded to the stage's type, fp32 state / gates / LayerNorm / softmax as
in the kernels) of the HIP path with one precision switch per stage:

    x   features as the layer1 operand          w1   layer1 weight operand
    y   layer1 output as stored between kernels e    LayerNorm+ReLU output = W_ih operand      wih  W_ih operand
    gi  input projection as stored              h    state as the W_hh operand (every step)    whh  W_hh operand
    hr  relu(h) as the classifier operand       wc   classifier weight operand

Each switch is one of f32 / bf16 / f16 (round to nearest even) or a split kind f16x2 / bf16x2 (hi + lo of that type, products as
a_hi.b_lo + a_lo.b_hi + a_hi.b_hi: what the fp16x2 kernels of round 4 compute).  The reference is the fp64 oracle on the same weights.
Output: argmax mismatches, the largest reference margin among them, mismatches above a 1e-3 margin, max |dprob| per
configuration -> profiles/precision_study_r03.json.  Test infrastructure: imports the oracle, runs on the CPU only.

    python scripts/precision_study.py [out.json] [--long] [--only-x2] [--no-scale]     # --long adds G2 T = 31 114; --no-scale: split weights unscaled
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import oracle_np as O
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg, epic_tent_cfg

STAGES = ("x", "w1", "y", "e", "wih", "gi", "h", "whh", "hr", "wc")


def split(t, kind, scale=1.0):
    """(hi, lo) of t * scale in the 16-bit type of a split-operand kind ('f16x2' / 'bf16x2'), both as fp32 tensors"""
    dt = torch.float16 if kind == "f16x2" else torch.bfloat16
    ts = t * scale
    hi = ts.to(dt).to(torch.float32)
    lo = (ts - hi).to(dt).to(torch.float32)
    return hi, lo


def p2scale(t):
    """the power of two that puts max|t| in [8192, 16384) (csrc/rowwise.hip: x2_weight_scale_kernel)"""
    import math
    return 2.0 ** math.floor(math.log2(16384.0 / float(t.abs().max())))


SCALE_W = True       # split-operand weights are pre-scaled by a power of two (their lo halves stay out of fp16's subnormals)


def mm(a, ka, b, kb):
    """a [M, K] (kind ka) times b [N, K]^T (kind kb); split kinds take the three-product form a_hi.b_lo + a_lo.b_hi + a_hi.b_hi"""
    if ka in ("f16x2", "bf16x2"):
        ah, al = split(a, ka)
        s = p2scale(b) if SCALE_W else 1.0
        bh, bl = split(b, kb, s)
        return (ah @ bh.T + (ah @ bl.T + al @ bh.T)) / s
    return rnd(a, ka) @ rnd(b, kb).T


def rnd(t, kind):
    if kind == "f32":
        return t
    if kind == "bf16":
        return t.to(torch.bfloat16).to(torch.float32)
    if kind == "f16":
        return t.to(torch.float16).to(torch.float32)
    raise ValueError(kind)


def forward(sd, rgb, flow, cfgp):
    """rgb [T,Dr] (flow [T,Df] or None) -> probabilities [T,C] (float64 softmax of the fp32 logits)."""
    p = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in sd.items() if v.dtype != np.int64}
    x = torch.from_numpy(rgb)
    w1 = p["layer1.0.weight"]
    if flow is not None:
        x = torch.cat([x, torch.from_numpy(flow)], 1)
    else:
        w1 = w1[:, : x.shape[1]]
    H = p["gru.weight_hh_l0"].shape[1]
    y = mm(x, cfgp["x"], w1, cfgp["w1"]) + p["layer1.0.bias"]
    y = rnd(y, cfgp["y"])
    mu = y.mean(1, keepdim=True)
    var = ((y - mu) ** 2).mean(1, keepdim=True)
    e = torch.relu((y - mu) / torch.sqrt(var + 1e-5) * p["layer1.1.weight"] + p["layer1.1.bias"])
    gi = mm(e, cfgp["e"], p["gru.weight_ih_l0"], cfgp["wih"])
    b_ih, b_hh = p["gru.bias_ih_l0"], p["gru.bias_hh_l0"]
    bias2 = b_ih.clone()
    bias2[: 2 * H] += b_hh[: 2 * H]                       # the kernels fold b_hh of the r, z rows into the projection's bias
    gi = rnd(gi + bias2, cfgp["gi"])
    x2h = cfgp["h"] in ("f16x2", "bf16x2")
    if x2h:
        sw = p2scale(p["gru.weight_hh_l0"]) if SCALE_W else 1.0
        wh, wl = split(p["gru.weight_hh_l0"], cfgp["whh"], sw)
        wh, wl = wh.T.contiguous(), wl.T.contiguous()
    else:
        whh = rnd(p["gru.weight_hh_l0"], cfgp["whh"]).T.contiguous()
    bhn = b_hh[2 * H:]
    T = x.shape[0]
    h = torch.zeros(H)
    hs = torch.empty(T, H)
    for t in range(T):
        if x2h:
            hh, hl = split(h, cfgp["h"])
            gh = (hh @ wh + (hh @ wl + hl @ wh)) / sw
        else:
            gh = rnd(h, cfgp["h"]) @ whh
        r = torch.sigmoid(gi[t, :H] + gh[:H])
        z = torch.sigmoid(gi[t, H:2 * H] + gh[H:2 * H])
        n = torch.tanh(gi[t, 2 * H:] + r * (gh[2 * H:] + bhn))
        h = (1.0 - z) * n + z * h
        hs[t] = h
    logits = mm(torch.relu(hs), cfgp["hr"], p["f_classification.0.weight"], cfgp["wc"]) + p["f_classification.0.bias"]
    return O.softmax(logits.numpy().astype(np.float64))


def compare(got, ref):
    srt = np.sort(ref, 1)
    margin = srt[:, -1] - srt[:, -2]
    mism = got.argmax(1) != ref.argmax(1)
    return {"frames": int(mism.size), "argmax_mismatches": int(mism.sum()),
            "mismatches_above_1e-3_margin": int((mism & (margin > 1e-3)).sum()),
            "largest_margin_among_mismatches": float(margin[mism].max()) if mism.any() else 0.0,
            "max_abs_dprob": float(np.abs(got - ref).max())}


def configs():
    allk = lambda k: {s: k for s in STAGES}
    out = {"all_f32": allk("f32")}
    cur = allk("bf16")
    out["r02_default(all bf16, Y/GI stored bf16)"] = cur
    out["bf16_operands_fp32_Y_GI"] = dict(cur, y="f32", gi="f32")
    for s in STAGES:                                       # ablation: restore one stage to fp32
        out[f"bf16_but_{s}_f32"] = dict(cur, **{s: "f32"})
    for s in STAGES:                                       # isolation: only this stage reduced
        out[f"only_{s}_bf16"] = dict(allk("f32"), **{s: "bf16"})
    out["all_f16"] = allk("f16")
    out["f16_operands_fp32_Y_GI"] = dict(allk("f16"), y="f32", gi="f32")
    out["bf16_with_h_f16"] = dict(cur, h="f16", whh="f16")
    out["bf16_with_ff_f16"] = dict(cur, x="f16", w1="f16", y="f16", e="f16", wih="f16", gi="f16")
    out["bf16_with_head_f32"] = dict(cur, hr="f32", wc="f32")
    # round 4: split operands (hi + lo of the 16-bit type, three products).  (a) the classifier alone on split operands under 16-bit
    # everything else - the cheap fix the round-3 verdict proposed for the bf16 head; (b) the whole path on fp16 pairs = fp16x2
    out["bf16_with_head_bf16x2"] = dict(cur, hr="bf16x2", wc="bf16x2")
    out["f16_with_head_f16x2"] = dict(allk("f16"), hr="f16x2", wc="f16x2")
    out["f16_with_head_f32"] = dict(allk("f16"), hr="f32", wc="f32")
    out["fp16x2 (all operands split, Y / GI / relu(h) / W_c fp32)"] = dict(allk("f16x2"), y="f32", gi="f32", hr="f32", wc="f32")
    out["fp16x2_with_split_head"] = dict(allk("f16x2"), y="f32", gi="f32")
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_path = args[0] if args else "profiles/precision_study_r04.json"
    long_t = "--long" in sys.argv
    torch.set_num_threads(8)
    cases = []
    cfg = assembly101_cfg()
    sd8 = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    cases.append(("g1_peaky", sd8, W.tsn_features((256, 2048), 20, "g1.rgb").reshape(256, 2048), None))
    cases.append(("g2_T4096", sd8, W.tsn_features((4096, 2048), 20, "g2.rgb.4096"), W.tsn_features((4096, 2048), 20, "g2.flow.4096")))
    if long_t:
        cases.append(("g2_T31114", sd8, W.tsn_features((31114, 2048), 20, "g2.rgb.31114"), None))
    ecfg = epic_tent_cfg()
    esd = W.miniroad_state_dict(ecfg, 20, head_gain=8.0)
    g7 = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g7_evaluate.json")))
    for i, T in enumerate(g7["lens"]):
        cases.append((f"g7_video{i}", esd, W.tsn_features((T, 2048), 20, f"g7.rgb.{i}"), None))
    # trained-like head: gain 32 (verdict: "peaky / trained-like fixture")
    sd32 = W.miniroad_state_dict(cfg, 20, head_gain=32.0)
    cases.append(("g1_gain32", sd32, W.tsn_features((256, 2048), 20, "g1.rgb").reshape(256, 2048), None))
    rep = {"note": "CPU emulation of the HIP path, one precision switch per stage (scripts/precision_study.py); reference = fp64 oracle"}
    refs = {}
    for name, sd, rgb, flow in cases:
        refs[name] = O.miniroad_forward(sd, rgb[None], None if flow is None else flow[None])["logits"][0]
    global SCALE_W
    SCALE_W = "--no-scale" not in sys.argv
    allc = configs()
    if "--only-x2" in sys.argv:
        allc = {k: v for k, v in allc.items() if "x2" in k or k in ("all_f32", "all_f16", "r02_default(all bf16, Y/GI stored bf16)", "bf16_with_head_f32", "f16_with_head_f32")}
    for cname, cfgp in allc.items():
        tot = None
        per = {}
        for name, sd, rgb, flow in cases:
            c = compare(forward(sd, rgb, flow, cfgp), refs[name])
            per[name] = c
            if tot is None:
                tot = dict(c)
            else:
                for k in ("frames", "argmax_mismatches", "mismatches_above_1e-3_margin"):
                    tot[k] += c[k]
                for k in ("largest_margin_among_mismatches", "max_abs_dprob"):
                    tot[k] = max(tot[k], c[k])
        rep[cname] = {"total": tot, "per_fixture": per}
        print(f"{cname:45s} mism {tot['argmax_mismatches']:4d} (>1e-3: {tot['mismatches_above_1e-3_margin']:3d})  "
              f"worst margin {tot['largest_margin_among_mismatches']:.2e}  max|dp| {tot['max_abs_dprob']:.2e}", flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(rep, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
