#!/bin/bash
# The histories of a gzip batch: one workgroup walking all stretches (HPN_GZ_WINDOWS=lds) against the three-step form (groups).
# Output: gpurun_out/ab_gz_windows.txt
out=gpurun_out/ab_gz_windows.txt
: > $out
for how in lds groups lds groups; do
  echo "== HPN_GZ_WINDOWS=$how" >> $out
  HPN_LIB=$PWD/highperformancengs_amd/testhooks/libhpngs.so HPN_GZ_WINDOWS=$how HPN_GZ_DEBUG=1 timeout 300 python scripts/bench_gz_inflate.py 2>&1 | grep -E "kernel|hpn_gz" | tail -3 >> $out
done
