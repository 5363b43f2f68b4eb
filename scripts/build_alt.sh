#!/bin/bash
# Build an A/B variant of the product library: ONE source recompiled with extra flags, linked with the tree's other objects.
#   bash scripts/build_alt.sh <name> <source in prego_amd/csrc> [extra hipcc flags...]   ->  prego_amd/lib_ab/lib<name>.so
# (lib_ab/*.so is git-ignored but travels to the GPU box; scripts/probes/ab_lib.sh alternates it against the tree's library)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p prego_amd/lib_ab
obj=prego_amd/lib_ab/${name}_$(echo $src | tr . _).o
x=""; case $src in *.cpp) x="-x hip";; esac
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-result -Wno-inline-asm "$@" $x -c prego_amd/csrc/$src -o $obj
objs=""
for o in prego_amd/lib/*.o; do
  case $o in *_dbg.o) continue;; esac
  if [ "$(basename $o)" = "$(echo $src | tr . _).o" ]; then objs="$objs $obj"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o prego_amd/lib_ab/lib${name}.so $objs
echo prego_amd/lib_ab/lib${name}.so
