#!/bin/bash
# Runs ON THE GPU BOX: only the kernel-trace stats leg of collect_profiles.sh (tag $1)
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PREGO_SPLIT_PASS=3 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-zero-flow --no-secondary > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-300
find $OUT -name "*kernel_trace.csv" -size +20M -delete
