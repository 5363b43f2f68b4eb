#!/bin/bash
export PREGO_AMD_DEBUG_LIB=1   # tuning knobs (PREGO_SPLIT_LAG*, PREGO_PLAN_SLOTS, PREGO_ATTN_NW, ...) are read by the debug library only (csrc/kernels.h: prego_tune_env)
# multi-tile recurrence steps: classic kernel (PREGO_GRU_NO_MT=1) vs the software-pipelined kernel, synth512 forced into 128 / 256 / 512
# slots = 1 / 2 / 4 clip tiles per group.  Runs on the GPU box.
for mt in 1 0; do
for sl in 128 256 512; do
  if [ $mt = 1 ]; then unset PREGO_GRU_NO_MT; else export PREGO_GRU_NO_MT=1; fi
  PREGO_PLAN_SLOTS=$sl timeout 300 python bench.py --steps 10 --warmup 3 --workload synth512 --no-cpu-baseline --no-secondary 2>/dev/null > /tmp/sw.json
  python - "$sl" "$mt" <<'PY'
import json, sys
sl, mt = int(sys.argv[1]), int(sys.argv[2])
d = json.loads(open("/tmp/sw.json").read().strip().splitlines()[-1])
g = d["rooflines"]["gru_recurrence"]
steps = 262144 // sl
print(json.dumps({"kernel": "pipelined" if mt else "classic", "slots": sl, "tiles_per_group": sl // 128, "pass_ms": round(d["ms_per_step"], 3),
                  "recurrence_ms": round(g["ms_per_step"], 3), "us_per_step": round(g["ms_per_step"] * 1e3 / steps, 3), "steps": steps, "output_sane": d["output_sane"]}))
PY
done
done
