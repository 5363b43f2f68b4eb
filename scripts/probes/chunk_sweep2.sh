#!/bin/bash
export PREGO_AMD_DEBUG_LIB=1   # tuning knobs (PREGO_SPLIT_LAG*, PREGO_PLAN_SLOTS, PREGO_ATTN_NW, ...) are read by the debug library only (csrc/kernels.h: prego_tune_env)
# rows-per-chunk sweep on the bench workload, pack prefetch grid 512, two rounds
export PREGO_PACK_PREFETCH_GRID=${GRID:-512}
for i in 1 2; do
for r in 24576 32768 40960 49152 65536 98304; do
  python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-zero-flow --rows-per-chunk $r 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['rooflines']; print($r, round(d['ms_per_step'],2), 'rec', round(r['gru_recurrence'].get('ms_per_step',0),2), 'gemm', round(r['gemm'].get('ms_per_step',0),2), 'pack', round(r['pack'].get('ms_per_step',0),2))"
done
done
