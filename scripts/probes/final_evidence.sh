#!/bin/bash
# Runs ON THE GPU BOX: end-of-round evidence in one call (tests, smoke, profiles, bench lines, parity report)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python __graft_entry__.py smoke 2>&1 | tail -2
bash scripts/collect_profiles.sh r02 2>&1 | tail -3
bash scripts/collect_secondary.sh r02 > gpurun_out/secondary_r02.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r02_line.json 2> gpurun_out/bench_r02.err; tail -c 600 gpurun_out/bench_r02_line.json
python bench.py --workload synth512 --no-cpu-baseline --no-secondary > gpurun_out/bench_r02_synth512.json 2>/dev/null; tail -c 300 gpurun_out/bench_r02_synth512.json
python scripts/parity_report.py > gpurun_out/parity_r02.log 2>&1; tail -3 gpurun_out/parity_r02.log
