#!/bin/bash
# Round 6, gzip route, second experiment: does the next batch's decode hide this batch's histories / translation / checks / framing
# (HPN_GZ_OVERLAP, test-hooks build), with the histories in the LDS-free form and with 20 instead of 24 decoders per CU (room for
# the other kernels beside them)?  -> gpurun_out/r06_gz/ab_gz_overlap.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06_gz; mkdir -p $O /dev/shm/gzp
CYC=${R06_CYCLES:-40}
python3 - <<PY
import os, sys
sys.path.insert(0, ".")
import torch
import highperformancengs_amd as hp
import bench_extra
from concurrent.futures import ThreadPoolExecutor
ctx = hp.Context(0)
texts = [bench_extra._fastq_text(ctx, 100_000, 150, 100 + k).tobytes() for k in range(50)]
ctx.close()
with ThreadPoolExecutor(16) as ex:
    cycle = b"".join(ex.map(bench_extra._gzip_one, texts))
cyc = $CYC
fd = os.open("/dev/shm/gzp/members.fq.gz", os.O_CREAT | os.O_WRONLY, 0o644)
os.ftruncate(fd, len(cycle) * cyc)
with ThreadPoolExecutor(16) as ex:
    list(ex.map(lambda c: os.pwrite(fd, cycle, c * len(cycle)), range(cyc)))
os.close(fd)
PY
cd /dev/shm/gzp
R=$GRAFT_REPO_ROOT
out=$O/${R06_OUT:-ab_gz_overlap.txt}
run() {  # label, bindir, env...
  label=$1; bin=$2; shift 2
  for rep in 1 2; do
    sleep 1.5
    s=$(date +%s.%N)
    row=$(env HPN_TIMING=1 "$@" $bin/fastq_count members.fq.gz 2> /tmp/err.txt | tail -1 | cut -f2-)
    e=$(date +%s.%N)
    echo "$label rep$rep wall $(echo "$e - $s" | bc -l | cut -c1-6) s | $row | $(grep 'gzip on the GPU' /tmp/err.txt | tail -1 | cut -c7-)" >> $out
  done
}
H=$R/highperformancengs_amd/testhooks/bin
run shipped $R/highperformancengs_amd/bin
run hooks_oversub3 $H HPN_GZ_OVERSUB=3
run hooks_overlap $H HPN_GZ_OVERLAP=1
run hooks_overlap_global $H HPN_GZ_OVERLAP=1 HPN_GZ_WINDOWS=global
run hooks_overlap_global_o3 $H HPN_GZ_OVERLAP=1 HPN_GZ_WINDOWS=global HPN_GZ_OVERSUB=3
run w20 $H LD_LIBRARY_PATH=$R/build_ab/w20h
run w20_overlap_global $H LD_LIBRARY_PATH=$R/build_ab/w20h HPN_GZ_OVERLAP=1 HPN_GZ_WINDOWS=global
run w20_overlap_global_o3 $H LD_LIBRARY_PATH=$R/build_ab/w20h HPN_GZ_OVERLAP=1 HPN_GZ_WINDOWS=global HPN_GZ_OVERSUB=3
run w20_overlap_groups $H LD_LIBRARY_PATH=$R/build_ab/w20h HPN_GZ_OVERLAP=1
cat $out
cd $R
for v in tree finddiag; do
  lib=""; [ $v != tree ] && lib=$R/build_ab/$v/libhpngs.so
  echo "== $v" >> $O/bench_gz_find2.txt
  HPN_LIB=$lib timeout 600 python3 scripts/bench_gz_find.py 2>&1 | grep -v amdgpu.ids >> $O/bench_gz_find2.txt
done
HPN_LIB=$R/build_ab/finddiag/libhpngs.so timeout 600 python3 scripts/bench_gz_find.py 18432 524288 2>&1 | grep -v amdgpu.ids >> $O/bench_gz_find2.txt
cat $O/bench_gz_find2.txt
rm -rf /dev/shm/gzp
