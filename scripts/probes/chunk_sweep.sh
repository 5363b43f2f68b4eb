#!/bin/bash
# pass time vs pipeline chunk size (packed rows per chunk) on the BASELINE configs[1] workload
for r in 24576 32768 49152 65536 98304; do
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-zero-flow --rows-per-chunk $r 2>/dev/null > /tmp/cs.json
  python - "$r" <<'PY'
import json, sys
d = json.loads(open("/tmp/cs.json").read().strip().splitlines()[-1])
print(f"rows/chunk {sys.argv[1]}: {d['ms_per_step']:.2f} ms", {k: round(v['ms_per_step'], 2) for k, v in d['rooflines'].items()})
PY
done
