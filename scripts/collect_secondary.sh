#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the secondary paths -> gpurun_out/prof_<tag>_<path>/
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
for P in train vit vit_train attn step; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_$P
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/secondary_profile.py $P > $OUT/log.txt 2>&1
  S=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== $P"; head -12 "$S" | cut -c1-220
  find $OUT -name "*kernel_trace.csv" -size +20M -delete
done
