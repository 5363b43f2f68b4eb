#!/bin/bash
# Runs ON THE GPU BOX: does a split pass survive counter collection (which serialises kernel dispatches)?  -> gpurun_out/prof_split_pmc/
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_split_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PREGO_SPLIT_PASS=3
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/probes/split_check.py 64 3000 9000 0 2 > $OUT/log.txt 2>&1
tail -6 $OUT/log.txt | cut -c1-300
find $OUT -name "*.csv" -size +20M -delete
