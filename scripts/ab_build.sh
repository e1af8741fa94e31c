#!/bin/bash
# Build variant libraries for same-session A/B runs:  scripts/ab_build.sh name "-DFLAG=.. -D.." [name2 "flags2" ...]
# -> .scratch/ab/<name>/libhpngs.so  (use with HPN_LIB=...)
set -e
cd "$(dirname "$0")/../highperformancengs_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  out=../../.scratch/ab/$name; mkdir -p $out/obj
  objs=""
  for f in kernels/*.hip *.hip; do
    o=$out/obj/$(echo $f | tr / _).o
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. -Ikernels -Wno-unused-function $flags -c $f -o $o &
    objs="$objs $o"
    while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.2; done
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libhpngs.so $objs -ldl
  rm -rf $out/obj
  echo "built $out/libhpngs.so ($flags)"
done
