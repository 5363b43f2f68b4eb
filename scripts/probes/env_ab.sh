#!/bin/bash
# same-device A/B of one environment knob on the bench workload: env_ab.sh VAR [rounds] [steps]; prints ms_per_step, alternating
VAR=$1; R=${2:-3}; STEPS=${3:-10}
for i in $(seq $R); do
  for v in 0 1; do
    if [ $v = 1 ]; then export $VAR=1; else unset $VAR; fi
    python3 bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-secondary --no-zero-flow 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['rooflines']; print('$VAR=$v', round(d['ms_per_step'],2), 'rec', round(r['gru_recurrence'].get('ms_per_step',0),2), 'gemm', round(r['gemm'].get('ms_per_step',0),2))"
  done
done
