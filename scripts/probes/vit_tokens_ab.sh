#!/bin/bash
export PREGO_AMD_DEBUG_LIB=1   # tuning knobs (PREGO_SPLIT_LAG*, PREGO_PLAN_SLOTS, PREGO_ATTN_NW, ...) are read by the debug library only (csrc/kernels.h: prego_tune_env)
# Runs ON THE GPU BOX: ViTEnc forward with the token rows written by the encoding GEMM's epilogue (default) vs the separate token kernel
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_transformer.py tests/test_gpu_vit_train.py -x -q 2>&1 | tail -2
for rep in 1 2; do for M in 0 1; do
  if [ $M = 1 ]; then export PREGO_VIT_TOKENS_KERNEL=1; else unset PREGO_VIT_TOKENS_KERNEL; fi
  echo "TOKENS_KERNEL=$M $(python - <<'PY'
import torch, time
from prego_amd import weights as W
from prego_amd.config import assembly101_cfg
from prego_amd.registry import build_model
import prego_amd.transformer
cfg = assembly101_cfg(model="Transformer", window_size=128, patch_dim=1, num_heads=8, attn_dropout_rate=0.0, dropout=0.0)
m = build_model(cfg, "cuda:0"); m.load_state_dict({k: torch.from_numpy(v) for k, v in W.vit_state_dict(cfg, 20).items()}); m.eval()
for B in (256, 1):
    xr, xf = torch.randn(B, 128, 2048, device="cuda"), torch.randn(B, 128, 2048, device="cuda")
    with torch.no_grad():
        for _ in range(5): m(xr, xf)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): m(xr, xf)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 20)
    print(f"B={B}: {best*1e3:.3f} ms", end="  ")
PY
)"
done; done
