#!/bin/bash
export PREGO_AMD_DEBUG_LIB=1   # tuning knobs (PREGO_SPLIT_LAG*, PREGO_PLAN_SLOTS, PREGO_ATTN_NW, ...) are read by the debug library only (csrc/kernels.h: prego_tune_env)
# Runs ON THE GPU BOX: what each part of the attention tile loop costs.  Needs a DIAGNOSTIC build of csrc/attention.hip that is not
# in the tree (results are wrong by construction): an extra kernel argument `dbg` read from PREGO_ATTN_DBG, bit 1 skips the in-loop
# LDS-DMA, bit 2 the softmax (P := constant), bit 4 the QK^T MFMAs, bit 8 the PV MFMAs.  Round-2 result (B=16, L=1024, 8 x 256, us):
# 0: 151, 1: 131, 2: 133, 4: 123, 8: 121, 12: 107, 14: 101, 15: 61 (flash_attention_v2 row).
cd /tmp && export TMPDIR=/tmp
for D in 0 1 2 4 8 12 14 15; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/attn_dbg_$D
  mkdir -p $OUT
  PREGO_ATTN_DBG=$D rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/secondary_profile.py attn > $OUT/log.txt 2>&1
  S=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "DBG=$D $(grep flash $S | awk -F"\"," "{print \$2}" | cut -d, -f3)"
  rm -rf $OUT
done
