#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the two launches of a split pass replayed one at a time (scripts/split_replay.py) under
#   (a) rocprofv3 --kernel-trace --stats      -> their solo durations
#   (b) rocprofv3 --pmc FETCH_SIZE            -> HBM-side read bytes per launch (x 2 on gfx950, MI355X_MICROARCH.md HBM section)
#   (c) rocprofv3 --pmc WRITE_SIZE            -> write bytes per launch
# in three separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters never share a run with a trace domain other than
# the kernel trace).  Outputs under gpurun_out/prof_<tag>_split/; scripts/summarize_split_pmc.py turns them into profiles/.
TAG=${1:-r05}
shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_split
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scripts/split_replay.py "$@" > $OUT/trace.log 2>&1
tail -6 $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/scripts/split_replay.py "$@" > $OUT/pmc_fetch.log 2>&1
tail -3 $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/scripts/split_replay.py "$@" > $OUT/pmc_write.log 2>&1
tail -3 $OUT/pmc_write.log
# (d) the matrix pipe: cycles the MFMA unit was busy, MFMA operations issued (x 512 = flops), chip-active cycles; (e) LDS activity
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_mfma -- python3 $GRAFT_REPO_ROOT/scripts/split_replay.py "$@" > $OUT/pmc_mfma.log 2>&1
tail -3 $OUT/pmc_mfma.log
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_lds -- python3 $GRAFT_REPO_ROOT/scripts/split_replay.py "$@" > $OUT/pmc_lds.log 2>&1
tail -3 $OUT/pmc_lds.log
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT -name "*.csv" | head -20
