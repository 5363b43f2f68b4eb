#!/bin/bash
# Runs ON THE GPU BOX: split-pass job stats incl. the shader clock the feed-forward XCDs actually ran at
export PREGO_AMD_DEBUG_LIB=1 PREGO_SPLIT_STATS=1 PREGO_SPLIT_PASS=3
timeout 300 python3 $GRAFT_REPO_ROOT/scripts/probes/split_check.py 182 3342 22000 1 2 2>&1 | tail -3
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | head -4
