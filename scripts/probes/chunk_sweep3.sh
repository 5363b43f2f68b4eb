#!/bin/bash
# rows-per-chunk sweep of the bench pass (fp16 and fp16x2): gpurun -- 'bash scripts/probes/chunk_sweep3.sh'
for DT in fp16 fp16x2; do
  for R in 24576 32768 49152 65536 98304; do
    ST=6; [ $DT = fp16x2 ] && ST=3
    python bench.py --dtype $DT --rows-per-chunk $R --steps $ST --warmup 2 --no-secondary --no-cpu-baseline --no-zero-flow 2>/dev/null > /tmp/line.json
    python3 -c "
import json; d=json.load(open('/tmp/line.json')); print('$DT rows $R', round(d['ms_per_step'],2), {k:round(v['ms_per_step'],1) for k,v in d['rooflines'].items()})"
  done
done
