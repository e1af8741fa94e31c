#!/bin/bash
# A/B of the two inflate kernels: the tree's library against variant builds under build_ab/<name>/libhpngs.so
# (scripts/ab_variant.sh).  Output: gpurun_out/ab_inflate.txt
# AB_STRETCHES: stretches of the gzip benchmark, "tree:variant" (each library at its own chip-fill), default 6144:6144
mkdir -p gpurun_out /tmp/abw
out=gpurun_out/ab_inflate.txt
: > $out
ns=${AB_STRETCHES:-6144:6144}
g++ -O2 -std=c++17 scripts/bam_synth.cpp -o /tmp/abw/bam_synth -lz -lpthread
/tmp/abw/bam_synth /tmp/abw/a.bam --targets chr1:120000000:14000000 12 >/dev/null 2>&1
ls -l /tmp/abw/a.bam >> $out
for v in tree "$@"; do
  lib=""; n=${ns%%:*}; [ $v != tree ] && lib=$PWD/build_ab/$v/libhpngs.so && n=${ns##*:}
  echo "== $v" >> $out
  HPN_LIB=$lib timeout 300 python scripts/bench_gz_inflate.py $n >> $out 2>&1
  HPN_LIB=$lib timeout 300 python scripts/bench_inflate.py /tmp/abw/a.bam 4e9 check >> $out 2>&1
done
