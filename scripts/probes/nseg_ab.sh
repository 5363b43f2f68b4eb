#!/bin/bash
# Runs ON THE GPU BOX: segments of the recurrence's gather (GRU_NSEG = 2 / 4 (tree) / 8), A/B builds in prego_amd/lib
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for N in 4 2 1; do
  if [ $N = 4 ]; then unset PREGO_AMD_LIB; else export PREGO_AMD_LIB=$GRAFT_REPO_ROOT/prego_amd/lib/libprego_nseg$N.so; fi
  echo "NSEG=$N $(python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2), round(d["rooflines"]["gru_recurrence"]["avg_launch_ms"],4))')"
done; done
