# A/B two builds on the same device: default library vs prego_amd/lib/alt/libprego_old.so (recurrence stamps, 128 clips)
for r in 1 2; do
for v in old default; do
  if [ $v = default ]; then unset PREGO_AMD_LIB; else export PREGO_AMD_LIB=$PWD/prego_amd/lib/alt/libprego_$v.so; fi
  echo "== $v"; python scripts/gru_stamps.py 128 bf16 2>&1 | tail -1
done; done
