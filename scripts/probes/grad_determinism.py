"""the training backward twice from the same state: every gradient must come out bit-identical (no race in the k-major GEMMs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_criterion, build_model
import prego_amd.loss, prego_amd.model  # noqa
dev = "cuda:0"
cfg = assembly101_cfg(compute_dtype="bf16", dropout=0.0)
m = build_model(cfg, dev); m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
crit = build_criterion(cfg, dev)
g = torch.Generator(device=dev); g.manual_seed(1)
rgb = torch.randn(16, 128, 2048, device=dev, generator=g).clamp_(min=0); flow = torch.randn(16, 128, 2048, device=dev, generator=g).clamp_(min=0)
tgt = torch.zeros(16, 128, 86, device=dev); tgt[:, :, 3] = 1
ref = None
for rep in range(6):
    m.train(); m.zero_grad(set_to_none=True)
    loss = crit(m(rgb, flow), tgt); loss.backward(); torch.cuda.synchronize()
    gr = {k: p.grad.clone() for k, p in m.named_parameters()}
    if ref is None:
        ref = gr
    else:
        bad = [k for k in gr if not torch.equal(gr[k], ref[k])]
        print("rep", rep, "differing tensors:", bad)
m.engine(train=True).check()
