#!/bin/bash
# Runs ON THE GPU BOX: same-device A/B of two builds of the library on the bench workload, alternating: the tree's library against
# $1 (a .so built from another version of one source, PREGO_AMD_LIB).  usage: bash scripts/probes/ab_lib.sh prego_amd/lib_ab/libX.so [bench args]
ALT=$GRAFT_REPO_ROOT/$1; shift
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for L in new old; do
    if [ $L = old ]; then export PREGO_AMD_LIB=$ALT; else unset PREGO_AMD_LIB; fi
    echo "$L $(python3 bench.py --no-cpu-baseline --no-secondary --no-zero-flow --steps 10 "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],2),'ms', d['pass'], 'rec us/step', round(d['rooflines']['gru_recurrence']['us_per_timestep'],4))")"
  done
done
