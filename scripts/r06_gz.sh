#!/bin/bash
# Round 6, the gzip route: same-session A/B of round 5's build (build_ab/r05) against the tree on the bench's 7.2 GB three-member
# file and on a C2-like file of many members; the search / CRC micro-benchmark; rocprofv3 over the tool.  -> gpurun_out/r06_gz/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BASE=${R06_BASE:-r05}   # the build the tree is compared with (build_ab/<name>/{libhpngs.so,bin})
O=$GRAFT_REPO_ROOT/gpurun_out/${R06_OUT:-r06_gz}; mkdir -p $O /dev/shm/gzp; : > $O/ab_gz_tool.txt; : > $O/bench_gz_find.txt
CYC=${R06_CYCLES:-40}
python3 - <<PY
import os, sys
sys.path.insert(0, ".")
import torch
import highperformancengs_amd as hp
import bench_extra
from concurrent.futures import ThreadPoolExecutor
ctx = hp.Context(0)
raw = bench_extra._fastq_text(ctx, 13_000_000, 150, 40).tobytes()
one = bench_extra._gz_single_member(raw, 256, 16)
with open("/dev/shm/gzp/gz3.fq.gz", "wb") as f:
    for _ in range(3):
        f.write(one)
del raw, one
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
print("members.fq.gz", len(cycle) * cyc / 1e9, "GB compressed,", sum(len(t) for t in texts) * cyc / 1e9, "GB of text")
PY
ls -l /dev/shm/gzp > $O/inputs.txt
cd /dev/shm/gzp
run() {  # label, bindir, env...
  label=$1; bin=$2; shift 2
  for f in gz3.fq.gz members.fq.gz; do
    for rep in 1 2; do
      sleep 1.5
      s=$(date +%s%N)
      row=$(env HPN_TIMING=1 "$@" $bin/fastq_count $f 2> /tmp/err.txt | tail -1)
      e=$(date +%s%N)
      echo "$label $f rep$rep wall $(( (e - s) / 1000000 )) ms | $row | $(grep 'gzip on the GPU' /tmp/err.txt | tail -1)" >> $O/ab_gz_tool.txt
    done
  done
}
R=$GRAFT_REPO_ROOT
run $BASE $R/build_ab/$BASE/bin
run tree $R/highperformancengs_amd/bin
run $BASE $R/build_ab/$BASE/bin
run tree $R/highperformancengs_amd/bin
cat $O/ab_gz_tool.txt
# ---- the search and the CRC alone ----
cd $R
for v in $BASE tree ${R06_FINDDIAG-finddiag}; do
  lib=""; [ $v != tree ] && lib=$R/build_ab/$v/libhpngs.so
  echo "== $v" >> $O/bench_gz_find.txt
  HPN_LIB=$lib timeout 600 python3 scripts/bench_gz_find.py >> $O/bench_gz_find.txt 2>&1
done
cat $O/bench_gz_find.txt
# ---- rocprofv3 over the tool ----
cd /dev/shm/gzp
for f in gz3 members; do
  rm -rf /tmp/prof_$f
  HPN_FULL_EXIT=1 HPN_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$f -o t -- $R/highperformancengs_amd/bin/fastq_count $f.fq.gz 2> /tmp/err_$f.txt > /dev/null
  cp $(find /tmp/prof_$f -name "*kernel_stats.csv" | head -1) $O/kernel_stats_gz_tool_$f.csv
  grep "hpn" /tmp/err_$f.txt | tail -3 >> $O/ab_gz_tool.txt
done
head -12 $O/kernel_stats_gz_tool_members.csv | cut -c1-160
rm -rf /dev/shm/gzp
