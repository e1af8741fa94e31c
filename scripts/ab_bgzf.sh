#!/bin/bash
# timing-only variants of k_bgzf_inflate (results are wrong by construction): build_ab/<name>/libhpngs.so -> gpurun_out/ab_bgzf.txt
mkdir -p gpurun_out /tmp/abw
out=gpurun_out/ab_bgzf.txt
: > $out
g++ -O2 -std=c++17 scripts/bam_synth.cpp -o /tmp/abw/bam_synth -lz -lpthread
/tmp/abw/bam_synth /tmp/abw/a.bam --targets chr1:120000000:14000000 12 >/dev/null 2>&1
for v in tree "$@"; do
  lib=""; [ $v != tree ] && lib=$PWD/build_ab/$v/libhpngs.so
  echo "== $v" >> $out
  HPN_LIB=$lib timeout 300 python scripts/bench_inflate.py /tmp/abw/a.bam 4e9 2>&1 | grep kernel_ms | cut -c1-220 >> $out
done
