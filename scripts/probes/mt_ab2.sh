#!/bin/bash
export PREGO_AMD_DEBUG_LIB=1   # tuning knobs (PREGO_SPLIT_LAG*, PREGO_PLAN_SLOTS, PREGO_ATTN_NW, ...) are read by the debug library only (csrc/kernels.h: prego_tune_env)
# same-box A/B of the multi-tile variants, three rounds, 512 slots (4 tiles) and 256 slots (2 tiles)
run() { PREGO_PLAN_SLOTS=$1 timeout 300 python bench.py --steps 10 --warmup 3 --workload synth512 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); g=d['rooflines']['gru_recurrence']; print('%s slots=%s us_per_step=%.3f pass_ms=%.3f' % ('$2', '$1', g['ms_per_step']*1e3/(262144//$1), d['ms_per_step']))"; }
for rep in 1 2 3; do
  for sl in 512 256; do
    unset PREGO_GRU_MT_SPEC; export PREGO_GRU_NO_MT=1; run $sl classic
    unset PREGO_GRU_NO_MT; export PREGO_GRU_MT_SPEC=0; run $sl check_first
    export PREGO_GRU_MT_SPEC=1; run $sl speculative
  done
done
