"""Workload for `rocprofv3 --kernel-trace --stats` over the secondary paths (one path per invocation, so that each kernel-stats
file is one path): train (MiniROAD train step 16 x 128), vit (ViTEnc forward, 256 windows x 128), vit_train (ViTEnc train step,
16 windows x 128), attn (causal AttentionLayer B = 16, L = 1024), step (the streaming fast path, 1 and 16 streams).
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 scripts/secondary_profile.py vit"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_criterion, build_model
import prego_amd.loss, prego_amd.model, prego_amd.transformer  # noqa: F401
from prego_amd.optim import FusedAdamW
from prego_amd.transformer import attention_layer

which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
if which == "train":
    cfg = assembly101_cfg(compute_dtype="bf16")
    m = build_model(cfg, dev)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
    crit = build_criterion(cfg, dev)
    opt = FusedAdamW([{"params": list(m.parameters())}], lr=1e-4, weight_decay=0.05, model=m)
    rgb = torch.randn(16, 128, 2048, device=dev).clamp_(min=0)
    flow = torch.randn(16, 128, 2048, device=dev).clamp_(min=0)
    tgt = torch.zeros(16, 128, 86, device=dev)
    tgt[:, :, 3] = 1
    for _ in range(n + 2):
        m.train()
        loss = crit(m(rgb, flow), tgt)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
elif which in ("vit", "vit_train"):
    vcfg = assembly101_cfg(model="Transformer", window_size=128, patch_dim=1, num_heads=8, attn_dropout_rate=0.0, dropout=0.0)
    vm = build_model(vcfg, dev)
    vm.load_state_dict({k: torch.from_numpy(v) for k, v in W.vit_state_dict(vcfg, 20).items()})
    if which == "vit":
        vm.eval()
        xr, xf = torch.randn(256, 128, 2048, device=dev), torch.randn(256, 128, 2048, device=dev)
        with torch.no_grad():
            for _ in range(n + 2):
                vm(xr, xf)
    else:
        crit = build_criterion(vcfg, dev)
        opt = FusedAdamW([{"params": list(vm.parameters())}], lr=1e-4, weight_decay=0.05, model=vm)
        xr, xf = torch.randn(16, 128, 2048, device=dev), torch.randn(16, 128, 2048, device=dev)
        tgt = torch.zeros(16, 128, 86, device=dev)
        tgt[:, :, 3] = 1
        vm.train()
        for _ in range(n + 2):
            loss = crit(vm(xr, xf), tgt)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
elif which == "attn":
    sdA = W.attention_layer_state_dict(2048, 20)
    names = ("query_projection", "key_projection", "value_projection", "out_projection")
    wargs = [torch.from_numpy(sdA[k + s]).to(dev) for k in names for s in (".weight", ".bias")]
    x = torch.randn(16, 1024, 2048, device=dev)
    for _ in range(n + 2):
        attention_layer(x, *wargs, n_heads=8, mask_flag=True)
elif which == "step":
    # the online fast path (prego_miniroad_step): 200 frames of one stream (rgb + flow), then 200 frames of 16 streams
    cfg = assembly101_cfg(compute_dtype="bf16")
    m = build_model(cfg, dev)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.miniroad_state_dict(cfg, 20).items()})
    m.eval()
    eng = m.engine()
    for ns in (1, 16):
        x = torch.randn(ns, 2048, device=dev).clamp_(min=0)
        f = torch.randn(ns, 2048, device=dev).clamp_(min=0)
        h = torch.zeros(ns, 1024, device=dev)
        out, arg = torch.empty(ns, 86, device=dev), torch.empty(ns, dtype=torch.int32, device=dev)
        for _ in range(200):
            eng.step(x, f, h, out=out, argmax=arg)
    eng.check()
torch.cuda.synchronize()
print("done", which)
