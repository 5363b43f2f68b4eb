#!/bin/bash
export PREGO_AMD_DEBUG_LIB=1   # tuning knobs (PREGO_SPLIT_LAG*, PREGO_PLAN_SLOTS, PREGO_ATTN_NW, ...) are read by the debug library only (csrc/kernels.h: prego_tune_env)
# Runs ON THE GPU BOX: ViTEnc forward (129-token windows) with the 4-wave (3 x 64 query slots) vs the 8-wave (2 x 128) attention shape
cd /tmp && export TMPDIR=/tmp
for NW in 4 8; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/vit_ab_$NW
  mkdir -p $OUT
  PREGO_ATTN_NW=$NW rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/secondary_profile.py vit > $OUT/log.txt 2>&1
  S=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "NW=$NW $(grep flash $S | awk -F'",' '{print $2}' | cut -d, -f1-3)"
  rm -rf $OUT
done
