#!/bin/bash
# Same-session A/B of K5 (k_window_add + k_window_rest): the tree's library and variant builds under build_ab/<name>/ (scripts/ab_build.sh),
# three rounds each, interleaved.  Output: gpurun_out/ab_k5.txt
out=gpurun_out/ab_k5.txt
: > $out
for round in 1 2 3; do
  for v in tree "$@"; do
    lib=""; [ $v != tree ] && lib=$PWD/build_ab/$v/libhpngs.so
    echo "== $v (round $round)" >> $out
    HPN_LIB=$lib timeout 300 python scripts/bench_kernels.py 7 k5 2>/dev/null | grep kernel >> $out
  done
done
