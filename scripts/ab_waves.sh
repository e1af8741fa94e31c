#!/bin/bash
# the inflate benchmarks with and without the wave-uniform state pinned to SGPRs (build_ab/nopin) -> gpurun_out/ab_waves.txt (round 3)
mkdir -p gpurun_out
out=gpurun_out/ab_waves.txt
: > $out
for v in tree nopin; do
  lib=""; [ $v != tree ] && lib=$PWD/build_ab/$v/libhpngs.so
  for w in 18 9 4 2; do
    echo "== $v waves/CU $w" >> $out
    HPN_INFLATE_WAVES=$w HPN_LIB=$lib timeout 300 python scripts/bench_gz_inflate.py 2>&1 | grep kernel_ms | cut -c1-200 >> $out
  done
done
