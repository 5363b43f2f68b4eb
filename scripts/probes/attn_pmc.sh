#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the causal attention forward (own passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
         "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  OUT=$GRAFT_REPO_ROOT/gpurun_out/attn_pmc_$i
  mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/secondary_profile.py attn > $OUT/log.txt 2>&1
  F=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "flash_attention" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()): print(f"{k:32s} {v / n:16.0f}  (per dispatch, {n} dispatches)")
PY
  find $OUT -name "*.csv" -size +5M -delete
done
