#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel trace of the data-parallel training step with the RCCL path forced in a one-rank group
#   gpurun -- 'bash scripts/collect_train_trace.sh r4f [bf16]'
TAG=${1:-r4_train}; COMP=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 PREGO_DP_FORCE_COLLECTIVE=1
cd /tmp; export TMPDIR=/tmp
ARGS="--mode train --steps 10 --warmup 3"
if [ -n "$COMP" ]; then ARGS="$ARGS --grad-compress $COMP"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/line.json 2> $OUT/err.log
cd $GRAFT_REPO_ROOT
python3 scripts/train_overlap_trace.py $OUT/trace $OUT/train_overlap_trace.json > $OUT/analysis.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/train_kernel_stats.csv \;
rm -rf $OUT/trace/*/*kernel_trace.csv.bak
tail -c 3000 $OUT/analysis.log
