#!/bin/bash
# Runs ON THE GPU BOX: the row jobs of the feed-forward launch (pack, LayerNorm; csrc/ff_pass.hip) of the tree against those of
# prego_amd/lib_ab/libold.so (scripts/build_old.sh ff_pass.hip): bit-identity tests, the launch's job sums in the pass and alone, the bench
# workload alternating, and the zero-flow (rgb only: the shipped Assembly101-O config) path.
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_split.py -x -q 2>&1 | tail -3
echo "== job sums of the feed-forward launch, tree"
PASS_CLOCK_FF=1 timeout 300 python3 scripts/probes/pass_clock.py 2>&1 | grep -i "feed-forward\|alone\|in the pass" | head
bash scripts/probes/ab_lib.sh prego_amd/lib_ab/libold.so
for r in 1 2; do
for L in new old; do
  if [ $L = old ]; then export PREGO_AMD_LIB=$GRAFT_REPO_ROOT/prego_amd/lib_ab/libold.so; else unset PREGO_AMD_LIB; fi
  echo "with zero-flow $L $(python3 bench.py --no-cpu-baseline --no-secondary --steps 10 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],2),'ms;  zero-flow M frames/s', round(d.get('frames_per_s_zero_flow_fastpath',0)/1e6,3))")"
done
done
