#!/bin/bash
# Same-session A/B of the depth kernels (index + sweep, two-pass K3 / K4, bedGraph text): the tree's library and variant builds under
# build_ab/<name>/ (scripts/ab_build.sh), two rounds each.  Output: gpurun_out/ab_sweep.txt
out=gpurun_out/ab_sweep.txt
: > $out
for round in 1 2; do
  for v in tree "$@"; do
    lib=""; [ $v != tree ] && lib=$PWD/build_ab/$v/libhpngs.so
    echo "== $v (round $round)" >> $out
    HPN_LIB=$lib timeout 300 python scripts/bench_depth_legs.py 2>/dev/null | grep kernel | sed 's/"algorithmic_bytes.*//' >> $out
  done
done
