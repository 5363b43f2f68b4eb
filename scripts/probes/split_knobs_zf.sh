#!/bin/bash
# as split_knobs.sh, library decides the pass (no PREGO_SPLIT_PASS), zero-flow fast path timed as well
cd $GRAFT_REPO_ROOT
for K in "$@"; do
  (
  export PREGO_AMD_DEBUG_LIB=1
  if [ "$K" != "-" ]; then IFS=, read -ra KV <<< "$K"; for kv in "${KV[@]}"; do export "$kv"; done; fi
  echo "$K: $(python3 bench.py --no-cpu-baseline --no-secondary --steps 10 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],2),'ms', d['pass']['mode'], d['pass'].get('recurrence_xcds'), 'zero-flow M frames/s', round(d['frames_per_s_zero_flow_fastpath']/1e6,2))")"
  )
done
