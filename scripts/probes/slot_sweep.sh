#!/bin/bash
export PREGO_AMD_DEBUG_LIB=1   # tuning knobs (PREGO_SPLIT_LAG*, PREGO_PLAN_SLOTS, PREGO_ATTN_NW, ...) are read by the debug library only (csrc/kernels.h: prego_tune_env)
# recurrence cost per time step by live tiles per group (calibrates kStepCost in csrc/miniroad.cpp): synth512 workload,
# slots forced to 128 / 256 / 512 = 1 / 2 / 4 tiles per group
for sl in 128 256 512; do
  PREGO_PLAN_SLOTS=$sl python bench.py --steps 10 --warmup 3 --workload synth512 --no-cpu-baseline 2>/dev/null > /tmp/sw.json
  python - "$sl" <<'PY'
import json, sys
sl = int(sys.argv[1])
d = json.loads(open("/tmp/sw.json").read().strip().splitlines()[-1])
g = d["rooflines"]["gru_recurrence"]
steps = 262144 // sl
print(f"slots {sl}: pass {d['ms_per_step']:.2f} ms, recurrence {g['ms_per_step']:.2f} ms = {g['ms_per_step']*1e3/steps:.3f} us per step ({steps} steps)")
PY
done
