# Where the page cache of a file lies against where the readers run: a 16.5 GB FASTQ written by a process held on node 0 / node 1,
# read by fastq_count with its feeders next to the device (default), anywhere (HPN_NUMA=0), or the whole tool held on one node.
#   -> gpurun_out/r05/numa_pagecache_probe.txt
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O /tmp/r05pc; out=$PWD/$O/numa_pagecache_probe.txt; : > $out
B=$PWD/highperformancengs_amd/bin
python - <<'PY'
import sys
sys.path.insert(0, ".")
import torch, highperformancengs_amd as hp, bench_extra
ctx = hp.Context(0)
bench_extra._fastq_text(ctx, 13_000_000, 150, 40).tofile("/tmp/r05pc/quarter.fq")
ctx.close()
PY
n0=$(lscpu | grep -E "NUMA node0 CPU" | awk '{print $NF}'); n1=$(lscpu | grep -E "NUMA node1 CPU" | awk '{print $NF}')
for d in /sys/class/drm/card*/device; do [ -n "$(cat $d/numa_node 2>/dev/null)" ] && echo "$(basename $(dirname $d)) numa_node=$(cat $d/numa_node) $(cat $d/uevent | grep PCI_SLOT_NAME)" >> $out; done
cd /tmp/r05pc
t() { l=$1; shift; for i in 1 2; do s=$(date +%s%N); "$@" > /dev/null 2> err.txt; e=$(date +%s%N); echo "$l : $(( (e - s) / 1000000 )) ms   $(grep -E 'stream|copy\+frame' err.txt | sed 's/.*MB, //; s/.*MB  //' | cut -c1-110)" >> $out; done; }
for node in 0 1; do
  cpus=$( [ $node = 0 ] && echo $n0 || echo $n1 )
  rm -f big.fq
  taskset -c $cpus sh -c 'cat quarter.fq quarter.fq quarter.fq quarter.fq > big.fq'
  echo "== big.fq written by a process on node $node ($cpus)" >> $out
  export HPN_TIMING=1
  t "default (feeders next to the device)" $B/fastq_count big.fq
  HPN_NUMA=0 t "HPN_NUMA=0" $B/fastq_count big.fq
  HPN_NUMA=0 t "whole tool on node 0, HPN_NUMA=0" taskset -c $n0 $B/fastq_count big.fq
  HPN_NUMA=0 t "whole tool on node 1, HPN_NUMA=0" taskset -c $n1 $B/fastq_count big.fq
  HPN_NGPU=1 t "default, one context" $B/fastq_count big.fq
done
rm -rf /tmp/r05pc
cat $out
