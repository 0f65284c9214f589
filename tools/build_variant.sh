#!/bin/bash
# Probe builds: tools/build_variant.sh <name> <file.hip> [-DFLAG=..]...  ->  build_probe/lib_<name>.so
# = the product library with ONE translation unit recompiled under extra defines (phase masks, experimental knobs).
# Used with CHITU_HIP_LIB=<path> for same-box A/Bs; build_probe/ is git-ignored but travels with gpurun.
set -e
name=$1; src=$2; shift 2
root=$(cd $(dirname $0)/.. && pwd)
cd $root/chitu_amd/csrc
make -s -j8 >/dev/null
mkdir -p $root/build_probe/obj_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include "$@" -c $src -o $root/build_probe/obj_$name/${src%.hip}.o
objs=""
for o in *.o; do
  if [ "$o" = "${src%.hip}.o" ]; then objs="$objs $root/build_probe/obj_$name/$o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build_probe/lib_$name.so $objs
echo build_probe/lib_$name.so
