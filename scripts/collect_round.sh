#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): everything profiles/ holds for one round, at the tree that travelled.
#   bash scripts/collect_round.sh r04          then, in the container:  python scripts/summarize_round.py r04
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 bench.py > $O/${TAG}_bench_default.log 2>&1;                                     tail -1 $O/${TAG}_bench_default.log > $O/${TAG}_bench_line.json
python3 bench.py --workload synth512 --no-cpu-baseline > $O/${TAG}_bench_synth512.log 2>&1; tail -1 $O/${TAG}_bench_synth512.log > $O/${TAG}_bench_line_synth512.json
python3 bench.py --mode train > $O/${TAG}_bench_train.log 2>&1;                           tail -1 $O/${TAG}_bench_train.log > $O/${TAG}_bench_line_train.json
python3 bench.py --mode train --local-batch 2 > $O/${TAG}_bench_train_b2.log 2>&1;        tail -1 $O/${TAG}_bench_train_b2.log > $O/${TAG}_bench_line_train_b2.json
python3 bench.py --dtype fp16x2 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_x2.log 2>&1; tail -1 $O/${TAG}_bench_x2.log > $O/${TAG}_bench_line_fp16x2.json
python3 scripts/parity_report.py $O/parity_${TAG}.json > $O/parity_${TAG}.log 2>&1
# split pass vs chunked pass on this box, alternating (same bench workload); job-time sums of the feed-forward launch (debug library)
for M in 0 auto 0 auto; do
  if [ $M = auto ]; then unset PREGO_SPLIT_PASS; else export PREGO_SPLIT_PASS=$M; fi
  echo "PREGO_SPLIT_PASS=${PREGO_SPLIT_PASS:-unset}"; python3 bench.py --no-cpu-baseline --no-secondary --no-zero-flow --steps 8 2>&1 | tail -1 | cut -c1-2200
done > $O/${TAG}_split_ab.log 2>&1
unset PREGO_SPLIT_PASS
PREGO_AMD_DEBUG_LIB=1 PREGO_SPLIT_STATS=1 PREGO_SPLIT_PASS=3 python3 scripts/probes/split_check.py 182 3342 22000 1 3 > $O/${TAG}_split_job_stats.log 2>&1
bash scripts/collect_profiles.sh $TAG > $O/${TAG}_collect_profiles.log 2>&1
bash scripts/collect_secondary.sh $TAG > $O/${TAG}_collect_secondary.log 2>&1
cd /tmp && export TMPDIR=/tmp
C=$O/prof_${TAG}_chunked
mkdir -p $C
PREGO_SPLIT_PASS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $C -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-zero-flow --no-secondary > $C/log.txt 2>&1
find $C -name "*kernel_trace.csv" -size +20M -delete
X=$O/prof_${TAG}_x2
mkdir -p $X
rocprofv3 --kernel-trace --stats --output-format csv -d $X -- python3 $R/bench.py --dtype fp16x2 --steps 2 --warmup 1 --no-cpu-baseline --no-zero-flow --no-secondary > $X/log.txt 2>&1
find $X -name "*kernel_trace.csv" -size +20M -delete
for f in $O/${TAG}_bench_line*.json; do echo "== $f"; cut -c1-600 $f; done
tail -5 $O/parity_${TAG}.log
