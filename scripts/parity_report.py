"""Argmax bookkeeping of the north star ("identical argmax action sequences") against the reference fixtures, kept as data:
for G1 (cfg1, plain and peaky head), G1c (head gain 32), G2 (T = 4096 with flow, T = 31 114), G7 (Evaluate end
to end) and G11 (TRAINED weights, both shipped configs, four videos each incl. T = 31 114) in fp16, bf16, fp32 and fp16x2 (split-operand) modes: frames, argmax mismatches, the LARGEST reference top-1/top-2 margin among the mismatching frames (the
"smallest margin that was still violated" bound: every frame whose margin exceeds it agrees), max |dprob|.

    python scripts/parity_report.py gpurun_out/parity_r04.json [fp16x2,fp32]     # on the GPU box; copy the file to profiles/
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import oracle_np as O            # checker only
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg, epic_tent_cfg
from prego_amd.registry import build_model
import prego_amd.model  # noqa: F401

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def model(cfg, sd, dtype):
    m = build_model(dict(cfg, compute_dtype=dtype), "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.eval()


def entry(got_arg, ref_arg, margin, dprob):
    mism = np.asarray(got_arg) != np.asarray(ref_arg)
    return {"frames": int(mism.size), "argmax_mismatches": int(mism.sum()),
            "mismatches_above_1e-3_margin": int((mism & (np.asarray(margin) > 1e-3)).sum()),
            "largest_margin_among_mismatches": float(margin[mism].max()) if mism.any() else 0.0,
            "smallest_reference_margin": float(margin.min()), "max_abs_dprob": float(dprob)}


DTYPES = ("fp16", "bf16", "fp32", "fp16x2")


def main(out_path):
    rep = {"note": "fp16 / bf16 = 16-bit MFMA operands and 16-bit Y / GI between the kernels, fp32 accumulation; fp32 = exact-fp32 MFMA; fp16x2 = split fp16 operands "
                   "(hi + lo, three MFMA products per product), fp32 intermediates.  Margins are the reference's top-1 minus "
                   "top-2 probability at that frame; random-init weights put many frames below any usable margin."}
    cfg = assembly101_cfg()
    for dtype in DTYPES:
        g = np.load(os.path.join(G, "g1c_miniroad_eval_gain32.npz"))
        m = model(cfg, W.miniroad_state_dict(cfg, 20, head_gain=32.0), dtype)
        outs, args, _ = m.engine().forward_ragged([torch.from_numpy(W.tsn_features((1024, 2048), 20, "g1c.rgb")).cuda()],
                                                  [torch.from_numpy(W.tsn_features((1024, 2048), 20, "g1c.flow")).cuda()], want_argmax=True)
        m.engine().check()
        rep[f"g1c_gain32_{dtype}"] = entry(args[0].cpu().numpy(), g["argmax"], g["margin"], np.abs(outs[0].cpu().numpy() - g["probs"]).max())
        for tag, gain in (("plain", 1.0), ("peaky", 8.0)):
            g = np.load(os.path.join(G, f"g1_miniroad_eval_{tag}.npz"))
            m = model(cfg, W.miniroad_state_dict(cfg, 20, head_gain=gain), dtype)
            rgb = torch.from_numpy(W.tsn_features((1, 256, 2048), 20, "g1.rgb")).cuda()
            with torch.no_grad():
                got = m(rgb, torch.zeros_like(rgb))["logits"][0].cpu().numpy()
            m.engine().check()
            srt = np.sort(g["probs"], 1)
            rep[f"g1_{tag}_{dtype}"] = entry(got.argmax(1), g["probs"].argmax(1), srt[:, -1] - srt[:, -2], np.abs(got - g["probs"]).max())
        sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
        m = model(cfg, sd, dtype)
        for T, with_flow in ((4096, True), (31114, False)):
            g = np.load(os.path.join(G, f"g2_miniroad_longT_{T}.npz"))
            rgb = torch.from_numpy(W.tsn_features((T, 2048), 20, f"g2.rgb.{T}")).cuda()
            flow = [torch.from_numpy(W.tsn_features((T, 2048), 20, f"g2.flow.{T}")).cuda()] if with_flow else None
            outs, args, _ = m.engine().forward_ragged([rgb], flow, want_argmax=True)
            m.engine().check()
            got = outs[0].cpu().numpy()
            rep[f"g2_T{T}_{dtype}"] = entry(args[0].cpu().numpy(), g["argmax"].astype(np.int32), g["margin"],
                                            np.abs(got[g["sample_idx"]] - g["sample_probs"]).max())
        # G7: the three Evaluate videos (Epic-tent-O head, 12 classes)
        g7 = json.load(open(os.path.join(G, "g7_evaluate.json")))
        ecfg = epic_tent_cfg()
        esd = W.miniroad_state_dict(ecfg, 20, head_gain=8.0)
        em = model(ecfg, esd, dtype)
        tot = {"frames": 0, "argmax_mismatches": 0, "mismatches_above_1e-3_margin": 0, "largest_margin_among_mismatches": 0.0, "smallest_reference_margin": 1.0, "max_abs_dprob": 0.0}
        for i, T in enumerate(g7["lens"]):
            x = W.tsn_features((T, 2048), 20, f"g7.rgb.{i}")
            ref = O.miniroad_forward(esd, x[None], None)["logits"][0]
            outs, args, _ = em.engine().forward_ragged([torch.from_numpy(x).cuda()], None, want_argmax=True)
            em.engine().check()
            srt = np.sort(ref, 1)
            e = entry(args[0].cpu().numpy(), np.array(g7["output"][f"synth_video_{i}"]["pred"]), srt[:, -1] - srt[:, -2],
                      np.abs(outs[0].cpu().numpy() - ref).max())
            tot["frames"] += e["frames"]
            tot["argmax_mismatches"] += e["argmax_mismatches"]
            tot["mismatches_above_1e-3_margin"] += e["mismatches_above_1e-3_margin"]
            tot["largest_margin_among_mismatches"] = max(tot["largest_margin_among_mismatches"], e["largest_margin_among_mismatches"])
            tot["smallest_reference_margin"] = min(tot["smallest_reference_margin"], e["smallest_reference_margin"])
            tot["max_abs_dprob"] = max(tot["max_abs_dprob"], e["max_abs_dprob"])
        rep[f"g7_evaluate_{dtype}"] = tot
        # G11: TRAINED weights (the reference's own training loop, oracle/train_g11.py), expected outputs from the reference's Evaluate
        from prego_amd import workloads as WL
        for tag, gcfg in (("a101", assembly101_cfg()), ("epic", epic_tent_cfg())):
            g = np.load(os.path.join(G, f"g11_eval_{tag}.npz"))
            gm = model(gcfg, W.g11_state_dict(tag), dtype)
            vids = [WL.action_video(int(T), gcfg["num_classes"], 20, f"g11.{tag}.eval.{i}")[0] for i, T in enumerate(g["lengths"])]
            outs, args, _ = gm.engine().forward_ragged([torch.from_numpy(v).cuda() for v in vids], None, want_argmax=True)
            gm.engine().check()
            tot = None
            for i in range(len(vids)):
                got = outs[i].cpu().numpy()
                e = entry(args[i].cpu().numpy(), g[f"pred{i}"].astype(np.int32), g[f"margin{i}"],
                          np.abs(got[g[f"sample_idx{i}"]] - g[f"sample_probs{i}"]).max())
                if tot is None:
                    tot = e
                else:
                    for k in ("frames", "argmax_mismatches", "mismatches_above_1e-3_margin"):
                        tot[k] += e[k]
                    for k in ("largest_margin_among_mismatches", "max_abs_dprob"):
                        tot[k] = max(tot[k], e[k])
                    tot["smallest_reference_margin"] = min(tot["smallest_reference_margin"], e["smallest_reference_margin"])
            tot["reference_mAP"] = float(g["mAP"])
            rep[f"g11_trained_{tag}_{dtype}"] = tot
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    json.dump(rep, open(out_path, "w"), indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2:
        DTYPES = tuple(sys.argv[2].split(","))
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_r04.json")
