for r in 1 2; do
for v in nseg1 nseg2 default; do
  if [ $v = default ]; then unset PREGO_AMD_LIB; else export PREGO_AMD_LIB=$PWD/prego_amd/lib/alt/libprego_$v.so; fi
  echo "== $v"; python scripts/gru_stamps.py 128 bf16 2>&1 | tail -1
done; done
