#!/bin/bash
# Runs ON THE GPU BOX: the bench workload with the split pass forced (PREGO_SPLIT_PASS=R), tree library and optionally an alt library ($1)
cd $GRAFT_REPO_ROOT
for L in new ${1:+old}; do
  if [ $L = old ]; then export PREGO_AMD_LIB=$GRAFT_REPO_ROOT/$1; else unset PREGO_AMD_LIB; fi
  for r in 1 2; do
  echo "$L $(PREGO_SPLIT_PASS=3 python3 bench.py --no-cpu-baseline --no-secondary --no-zero-flow --steps 10 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],2),'ms', d['pass'], 'rec us/step', round(d['rooflines']['gru_recurrence']['us_per_timestep'],4), 'ff ms', round(d['rooflines']['gemm']['ms_per_step'],2))")"
  done
done
