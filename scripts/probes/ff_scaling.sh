#!/bin/bash
# Runs ON THE GPU BOX: the feed-forward launch of the split pass ALONE (scripts/split_replay.py, debug library) on 7 .. 3 XCDs and with
# 2 / 4 / 8 units per super-round.  If launch time x XCDs is constant the launch is bound per XCD (LDS / L2 / its fabric port); if it
# grows with the XCD count, by something the XCDs share (Infinity Cache, HBM).  PREGO_SPLIT_SG changes how many workgroups share a
# weight slab through the XCD's L2 (fabric traffic per tile).
OUT=$GRAFT_REPO_ROOT/gpurun_out/ff_scaling
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for R in 1 2 3 4 5; do
  echo "== R=$R ($((8-R)) feed-forward XCDs)"
  PREGO_SPLIT_PASS=$R python3 scripts/split_replay.py --reps 2 2>&1 | grep -E "alone|Error|error" | tee -a $OUT/log.txt
done
for SG in 2 8; do
  echo "== R=3 SG=$SG"
  PREGO_SPLIT_SG=$SG PREGO_SPLIT_PASS=3 python3 scripts/split_replay.py --reps 2 2>&1 | grep -E "alone|Error|error" | tee -a $OUT/log.txt
done
