#!/bin/bash
# One variant library for a same-session A/B:  scripts/ab_variant.sh name "-DFLAG .." file.hip [file2.hip ..]
# compiles only the named sources of highperformancengs_amd/csrc with the flags, links them with the tree's other objects
# (csrc/build, i.e. after `make shipped`) -> build_ab/<name>/libhpngs.so (travels to the GPU box; use with HPN_LIB=...).
set -e
name=$1; flags=$2; shift 2
R="$(cd "$(dirname "$0")/.." && pwd)"
cd $R/highperformancengs_amd/csrc
out=$R/build_ab/$name; mkdir -p $out/obj
skip=""
for f in "$@"; do
  o=$out/obj/$(basename $f .hip).o
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$R/include -I. -Ikernels -Wall -Wno-unused-function $flags -c $f -o $o &
  skip="$skip -e /$(basename $f .hip).o\$"
done
wait
others=$(ls build/kernels/*.o build/*.o | grep -v $skip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libhpngs.so $others $out/obj/*.o -ldl
rm -rf $out/obj
echo "built $out/libhpngs.so ($flags)"
