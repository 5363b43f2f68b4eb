#!/bin/bash
# A/B baseline: prego_amd/lib_ab/libold.so = the library of git HEAD (sources of HEAD for the files given, the tree's objects for the rest).
#   bash scripts/build_old.sh gru_recurrence.hip [more files of prego_amd/csrc ...]      (run next to `python -m prego_amd.build`)
set -e
cd "$(dirname "$0")/.."
rm -rf gpurun_out/_oldsrc && mkdir -p gpurun_out/_oldsrc prego_amd/lib_ab
git archive HEAD prego_amd/csrc include | tar -x -C gpurun_out/_oldsrc
for f in "$@"; do
  x=""; case $f in *.cpp) x="-x hip";; esac
  (cd gpurun_out/_oldsrc && /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-result -Wno-inline-asm $x -c prego_amd/csrc/$f -o $(echo $f | tr . _).o) &
done
wait
objs=""
for o in prego_amd/lib/*.o; do
  case $o in *_dbg.o) continue;; esac
  b=$(basename $o)
  if [ -f gpurun_out/_oldsrc/$b ]; then objs="$objs gpurun_out/_oldsrc/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o prego_amd/lib_ab/libold.so $objs
echo prego_amd/lib_ab/libold.so
