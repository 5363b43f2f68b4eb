#!/bin/bash
for r in 16384 32768 65536 131072 262144; do
  timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-zero-flow --rows-per-chunk $r 2>&1 | tail -1 > /tmp/line.json
  python - <<PY
import json
d=json.load(open('/tmp/line.json'))
print($r, round(d["value"]/1e6,2), "Mfps", round(d["ms_per_step"],1), "ms", {k:round(v["ms_per_step"],1) for k,v in d["rooflines"].items()})
PY
done
