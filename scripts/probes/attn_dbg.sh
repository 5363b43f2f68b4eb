#!/bin/bash
# Runs ON THE GPU BOX: what each part of the attention tile loop costs (diagnostic build only: PREGO_ATTN_DBG bits skip parts)
cd /tmp && export TMPDIR=/tmp
for D in 0 1 2 4 8 12 14 15; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/attn_dbg_$D
  mkdir -p $OUT
  PREGO_ATTN_DBG=$D rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/secondary_profile.py attn > $OUT/log.txt 2>&1
  S=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "DBG=$D $(grep flash $S | awk -F"\"," "{print \$2}" | cut -d, -f3)"
  rm -rf $OUT
done
