#!/bin/bash
# Runs ON THE GPU BOX: what is this device (clocks, power cap, partition modes) and how do the two passes compare on it?
rocm-smi --showclocks --showpower --showmaxpower --showmemorypartition --showcomputepartition --showperflevel 2>&1 | grep -v "^=\|^$" | head -30
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|Name: +gfx" | head -8
for M in 0 3; do PREGO_SPLIT_PASS=$M timeout 200 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-zero-flow --steps 6 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('mode', d['pass']['mode'], d['ms_per_step'], 'ff/gemm ms', d['rooflines']['gemm']['ms_per_step'], 'gru ms', d['rooflines']['gru_recurrence']['ms_per_step'])"; done
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power" | head -6
