#!/bin/bash
export PREGO_AMD_DEBUG_LIB=1   # tuning knobs (PREGO_SPLIT_LAG*, PREGO_PLAN_SLOTS, PREGO_ATTN_NW, ...) are read by the debug library only (csrc/kernels.h: prego_tune_env)
# Runs ON THE GPU BOX: knob sweep of the split pass on one device (bench workload, forced R = 3; zero-flow forced R = 4)
cd $GRAFT_REPO_ROOT
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-secondary --steps 6 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['pass']['mode'], round(d['ms_per_step'],2), round(d['rooflines']['gru_recurrence']['us_per_timestep'],4), round(d.get('frames_per_s_zero_flow_fastpath',0)/1e6,2))"; }
export PREGO_SPLIT_PASS=3
run base
for v in nseg1 nseg4; do PREGO_AMD_LIB=$PWD/prego_amd/lib/alt/libprego_$v.so run $v; done
PREGO_SPLIT_LAG1=1 PREGO_SPLIT_LAG2=2 PREGO_SPLIT_LAG3=3 run lag123
PREGO_SPLIT_LAG1=2 PREGO_SPLIT_LAG2=4 PREGO_SPLIT_LAG3=5 run lag245
PREGO_SPLIT_LAG1=3 PREGO_SPLIT_LAG2=4 PREGO_SPLIT_LAG3=5 run lag345
run base
export PREGO_SPLIT_PASS=4
run r4_base
PREGO_SPLIT_LAG1=1 PREGO_SPLIT_LAG2=2 PREGO_SPLIT_LAG3=3 run r4_lag123
PREGO_SPLIT_LAG1=3 PREGO_SPLIT_LAG2=4 PREGO_SPLIT_LAG3=5 run r4_lag345
