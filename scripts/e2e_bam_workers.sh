#!/bin/bash
# bam2depth / bam_sliding_count on one synthetic BAM with 1, 2, 4 workers (HPN_NGPU) -- on a one-GPU box the workers share the
# device: this shows what the several-GPU routes cost, not what they gain.   scripts/e2e_bam_workers.sh [reads] [contigs]
set -e
cd "$(dirname "$0")/.."
reads=${1:-8000000}; contigs=${2:-4}
td=$(mktemp -d /tmp/hpn_w_XXXX)
g++ -O2 -std=c++17 scripts/bam_synth.cpp -o $td/bam_synth -lz -lpthread
$td/bam_synth $td/s.bam $reads $contigs $((reads*150/30/contigs)) 16
ls -la $td/s.bam | awk '{print "BAM bytes", $5}'
for tool in bam2depth bam_sliding_count; do
  for n in 1 2 4; do
    mkdir -p $td/o$n; cd $td/o$n; ln -sf ../s.bam .; ln -sf ../s.bam.bai .
    HPN_NGPU=$n $OLDPWD/highperformancengs_amd/bin/$tool -o x s.bam > /dev/null 2>&1   # warm
    t0=$(date +%s%N)
    HPN_NGPU=$n HPN_TIMING=1 $OLDPWD/highperformancengs_amd/bin/$tool -o x s.bam 2>&1 | grep -E "workers|ingest" | tr '\n' ' '
    t1=$(date +%s%N); echo "$tool HPN_NGPU=$n  $(( (t1 - t0) / 1000000 )) ms"
    cd $OLDPWD
  done
  cmp $td/o1/$( [ $tool = bam2depth ] && echo s.bam.1.bedGraph || echo x.txt ) $td/o4/$( [ $tool = bam2depth ] && echo s.bam.1.bedGraph || echo x.txt ) && echo "  outputs of 1 and 4 workers identical"
done
rm -rf $td
