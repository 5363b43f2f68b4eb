#!/bin/bash
# Runs ON THE GPU BOX: the bench workload's forced split pass on the DEBUG library under a list of tuning-knob settings (one per argument,
# "NAME=VALUE[,NAME=VALUE]"; "-" = defaults).  usage: bash scripts/probes/split_knobs.sh - PREGO_SPLIT_CHUNK_SHIFT=5 ...
cd $GRAFT_REPO_ROOT
for K in "$@"; do
  (
  export PREGO_AMD_DEBUG_LIB=1 PREGO_SPLIT_PASS=3
  if [ "$K" != "-" ]; then IFS=, read -ra KV <<< "$K"; for kv in "${KV[@]}"; do export "$kv"; done; fi
  for r in 1 2; do
  echo "$K: $(python3 bench.py --no-cpu-baseline --no-secondary --no-zero-flow --steps 10 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],2),'ms', d['pass']['mode'], d['pass'].get('recurrence_xcds'), 'rec us/step', round(d['rooflines']['gru_recurrence']['us_per_timestep'],4), 'ff ms', round(d['rooflines']['gemm']['ms_per_step'],2))")"
  done
  )
done
