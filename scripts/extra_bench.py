"""Secondary measurements for DESIGN.md (not the headline metric): training step (BASELINE config 3 shape, one GPU),
ViTEnc forward, causal attention at L = 1024 (config 4), batch-1 streaming latency (config 1 shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model, build_criterion
import prego_amd.model, prego_amd.loss, prego_amd.transformer
from prego_amd.transformer import attention_layer


def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

cfg = assembly101_cfg(compute_dtype="bf16")
sd = W.miniroad_state_dict(cfg, 20)
m = build_model(cfg, "cuda:0"); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
crit = build_criterion(cfg, "cuda:0")
opt = torch.optim.AdamW([{"params": m.parameters(), "initial_lr": 1e-4}], lr=1e-4, weight_decay=0.05)
B, T = 16, 128
rgb = torch.randn(B, T, 2048, device="cuda").clamp_(min=0); flow = torch.randn(B, T, 2048, device="cuda").clamp_(min=0)
tgt = torch.zeros(B, T, 86, device="cuda"); tgt[:, :, 3] = 1
def train_step():
    m.train(); loss = crit(m(rgb, flow), tgt); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
ms = timeit(train_step, 10, 3)
print(f"train step B=16 T=128 (fwd+loss+bwd+AdamW, dropout 0.2, bf16 operands): {ms:.2f} ms  (reference CPU probe: 11 100 ms)")
m.eval()
x1 = rgb[:1, :].repeat(1, 2, 1).contiguous(); f1 = flow[:1].repeat(1, 2, 1).contiguous()
with torch.no_grad():
    ms = timeit(lambda: m(x1, f1), 20, 3)
print(f"eval B=1 T=256 (BASELINE config 1 shape): {ms:.3f} ms = {256/ms*1e3:.0f} frames/s  (reference CPU probe: 59.3 ms)")
vcfg = assembly101_cfg(model="Transformer", window_size=128, patch_dim=1, num_heads=8, attn_dropout_rate=0.0, dropout=0.0)
vm = build_model(vcfg, "cuda:0"); vm.load_state_dict({k: torch.from_numpy(v) for k, v in W.vit_state_dict(vcfg, 20).items()}); vm.eval()
for Bv in (16, 256):
    xr = torch.randn(Bv, 128, 2048, device="cuda"); xf = torch.randn(Bv, 128, 2048, device="cuda")
    with torch.no_grad():
        ms = timeit(lambda: vm(xr, xf), 10, 3)
    fl = Bv * 7.69e9
    print(f"ViTEnc forward B={Bv} windows of 128: {ms:.2f} ms = {Bv/ms*1e3:.0f} windows/s, {fl/ms/1e9:.0f} TFLOP/s  (reference CPU probe: 53 windows/s)")
sdA = W.attention_layer_state_dict(2048, 20)
names = ("query_projection", "key_projection", "value_projection", "out_projection")
args = [torch.from_numpy(sdA[n + s]).cuda() for n in names for s in (".weight", ".bias")]
for Bc, L in ((1, 1024), (16, 1024)):
    x = torch.randn(Bc, L, 2048, device="cuda")
    ms = timeit(lambda: attention_layer(x, *args, n_heads=8, mask_flag=True), 10, 3)
    fl = Bc * (2 * L * 2048 * 2048 * 4 + 4 * 8 * L * L * 256 / 2)
    print(f"causal AttentionLayer B={Bc} L={L} d=2048 h=8: {ms:.3f} ms, {fl/ms/1e9:.0f} TFLOP/s (projections + causal attention)")
