#!/bin/bash
export PREGO_AMD_DEBUG_LIB=1   # tuning knobs (PREGO_SPLIT_LAG*, PREGO_PLAN_SLOTS, PREGO_ATTN_NW, ...) are read by the debug library only (csrc/kernels.h: prego_tune_env)
# Runs ON THE GPU BOX: A/B of the attention forward shapes (PREGO_ATTN_NW=4: 4 waves x 32 queries; default: 8 waves x 16 queries)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_transformer.py tests/test_gpu_vit_train.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for NW in 4 8; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/attn_ab_$NW
  mkdir -p $OUT
  PREGO_ATTN_NW=$NW rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/secondary_profile.py attn > $OUT/log.txt 2>&1
  S=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== NW=$NW"; head -6 "$S" | cut -c1-200
  find $OUT -name "*kernel_trace.csv" -size +20M -delete
done
