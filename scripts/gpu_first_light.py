"""first-light debug script (run on the GPU box): cfg1 vs golden with intermediate prints"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.model
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g = np.load(os.path.join(G, "g1_miniroad_eval_peaky.npz"))
for dtype in ("fp32", "bf16"):
    cfg = assembly101_cfg(compute_dtype=dtype)
    sd = W.miniroad_state_dict(cfg, 20, head_gain=8.0)
    m = build_model(cfg, "cuda:0")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    rgb = torch.from_numpy(W.tsn_features((1, 256, 2048), 20, "g1.rgb")).cuda()
    flow = torch.zeros_like(rgb)
    t = time.time()
    with torch.no_grad():
        out = m(rgb, flow)["logits"]
    try:
        m.engine().check()
    except Exception as e:
        print("CHECK FAILED", e)
    torch.cuda.synchronize()
    o = out[0].cpu().numpy()
    print(dtype, "time", time.time() - t, "max|dprob|", np.abs(o - g["probs"]).max(), "argmax mism", int((o.argmax(1) != g["argmax"]).sum()),
          "nan", int(np.isnan(o).sum()), "first rows err", np.abs(o - g["probs"]).max(1)[:6])
