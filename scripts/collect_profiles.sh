#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the bench, then two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950) on a shorter workload with identical per-launch shapes.
# Outputs under gpurun_out/prof_<tag>/ ; scripts/summarize_profiles.py turns them into profiles/*.
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the split pass is forced for this run (PREGO_SPLIT_PASS=3: every call after the handle's first): with 1 + 3 calls the default handle would
# still be timing its chunked pass and its split trial (DESIGN 5b "When"); scripts/collect_round.sh profiles the chunked pass separately
PREGO_SPLIT_PASS=3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-zero-flow --no-secondary > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-400
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-zero-flow --no-secondary --clips 64 --len-scale 0.25 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-zero-flow --no-secondary --clips 64 --len-scale 0.25 > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
# keep the merge small: the kernel traces are large
find $OUT -name "*kernel_trace.csv" -size +20M -delete
