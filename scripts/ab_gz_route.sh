# fastq_count on the 7.2 GB three-member .fastq.gz: the default route against its test-hooks variants (second decode context,
# batch sizes)  -> gpurun_out/r05/ab_gz_route.txt     (after KEEP_INPUTS=1 scripts/prof_r05_tools.sh, which makes /tmp/r05in/gz3.fq.gz)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O; out=$PWD/$O/ab_gz_route.txt; : > $out
B=$PWD/highperformancengs_amd/testhooks/bin
cd /tmp/r05in || exit 1
t() { for i in 1 2 3; do s=$(date +%s%N); env "$@" HPN_TIMING=1 $B/fastq_count gz3.fq.gz > /dev/null 2> err.txt; e=$(date +%s%N); echo "$* : $(( (e - s) / 1000000 )) ms   $(grep -E 'gzip on the GPU' err.txt | sed 's/\[hpn\] gzip on the GPU: //')" >> $out; done; }
t A=1
t HPN_GZ_OVERLAP=1
t HPN_GZ_BATCH=3072
t HPN_GZ_BATCH=2048
t HPN_GZ_BATCH=3072 HPN_GZ_OVERLAP=1
t HPN_GZ_STRETCH=524288
cat $out
rm -rf /tmp/r05in     # (boxes are reused: leave the disk as it was found)
