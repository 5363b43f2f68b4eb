#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the bench in split mode -> gpurun_out/prof_split_<R>/
R=${1:-3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_split_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PREGO_SPLIT_PASS=$R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-zero-flow --no-secondary > $OUT/log.txt 2>&1
S=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$S" ]; then head -8 "$S" | cut -c1-220; else tail -5 $OUT/log.txt; fi
find $OUT -name "*kernel_trace.csv" -size +20M -delete
