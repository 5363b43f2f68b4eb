#!/bin/bash
# Runs ON THE GPU BOX: same-device A/B of two builds of the library on the bench workload in fp16x2 (split-operand) mode
ALT=$GRAFT_REPO_ROOT/$1; shift
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for L in new old; do
    if [ $L = old ]; then export PREGO_AMD_LIB=$ALT; else unset PREGO_AMD_LIB; fi
    echo "$L $(python3 bench.py --dtype fp16x2 --no-cpu-baseline --no-secondary --no-zero-flow --steps 5 "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],2),'ms', d['pass']['mode'], 'rec ms', round(d['rooflines']['gru_recurrence']['ms_per_step'],2))")"
  done
done
