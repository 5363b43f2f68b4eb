#!/bin/bash
# Runs ON THE GPU BOX: same-device A/B of two builds of the library on the configs[4] per-GPU share (512 clips x 512 frames: multi-tile recurrence)
ALT=$GRAFT_REPO_ROOT/$1; shift
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for L in new old; do
    if [ $L = old ]; then export PREGO_AMD_LIB=$ALT; else unset PREGO_AMD_LIB; fi
    echo "$L $(python3 bench.py --workload synth512 --no-cpu-baseline --no-secondary --steps 20 "$@" 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3),'ms', round(d['value']/1e6,2), 'M frames/s')")"
  done
done
