#!/bin/bash
# Runs ON THE GPU BOX: end-of-round evidence in one call (tests, smoke, profiles, bench lines, parity report).  TAG = round tag.
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | tail -3
bash scripts/collect_profiles.sh $TAG 2>&1 | tail -3
bash scripts/collect_secondary.sh $TAG > gpurun_out/secondary_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_${TAG}_line.json 2> gpurun_out/bench_$TAG.err; tail -c 400 gpurun_out/bench_${TAG}_line.json
python bench.py --workload synth512 --no-cpu-baseline --no-secondary > gpurun_out/bench_${TAG}_synth512.json 2>/dev/null; tail -c 300 gpurun_out/bench_${TAG}_synth512.json
python bench.py --mode train --steps 50 --warmup 10 > gpurun_out/bench_${TAG}_train.json 2>/dev/null; tail -c 300 gpurun_out/bench_${TAG}_train.json
python bench.py --mode train --steps 50 --warmup 10 --local-batch 2 > gpurun_out/bench_${TAG}_train_b2.json 2>/dev/null; tail -c 200 gpurun_out/bench_${TAG}_train_b2.json
python scripts/parity_report.py gpurun_out/parity_$TAG.json > gpurun_out/parity_$TAG.log 2>&1; tail -3 gpurun_out/parity_$TAG.log
