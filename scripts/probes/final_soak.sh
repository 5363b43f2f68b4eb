#!/bin/bash
# Runs ON THE GPU BOX: the whole GPU suite, then 20 split passes per configuration checked bit for bit against the chunked pass
cd $GRAFT_REPO_ROOT
timeout 2000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -2
for R in 3 4; do PREGO_SPLIT_PASS=$R timeout 500 python3 scripts/probes/split_check.py 182 3342 22000 1 20 2>&1 | grep -c "bit-identical to call 0: True"; done
SPLIT_DTYPE=bf16 PREGO_SPLIT_PASS=3 timeout 500 python3 scripts/probes/split_check.py 300 900 2900 0 20 2>&1 | grep -c "bit-identical to call 0: True"
