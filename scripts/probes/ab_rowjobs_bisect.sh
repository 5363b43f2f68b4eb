#!/bin/bash
# on the GPU box: which part of the rewritten LayerNorm job breaks bit-identity with the chunked pass?
cd $GRAFT_REPO_ROOT
for v in lngbglobal lnnofence; do
  echo "== $v"
  PREGO_AMD_LIB=$GRAFT_REPO_ROOT/prego_amd/lib_ab/lib$v.so timeout 600 python3 -m pytest "tests/test_gpu_split.py::test_split_pass_equals_chunked_pass_bit_for_bit" -x -q 2>&1 | tail -3
done
