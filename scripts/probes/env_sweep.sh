#!/bin/bash
# same-device sweep of one environment variable on the bench workload: env_sweep.sh VAR "v1 v2 ..." [rounds] [steps]  ("-" = unset)
VAR=$1; VALS=$2; R=${3:-2}; STEPS=${4:-10}
for i in $(seq $R); do
  for v in $VALS; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    python3 bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-secondary --no-zero-flow 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['rooflines']; print('$VAR=$v', round(d['ms_per_step'],2), 'rec', round(r['gru_recurrence'].get('ms_per_step',0),2), 'gemm', round(r['gemm'].get('ms_per_step',0),2), 'pack', round(r['pack'].get('ms_per_step',0),2))"
  done
done
