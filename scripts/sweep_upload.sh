# bam_sliding_count on the 10.6 GB BAM under read-thread counts and chunk sizes: what feeds the device fastest?  -> gpurun_out/r05/sweep_upload.txt
# (after KEEP_INPUTS=1 scripts/prof_r05_tools.sh, which makes /tmp/r05in/hg38.bam)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O; out=$PWD/$O/sweep_upload.txt; : > $out
B=$PWD/highperformancengs_amd/testhooks/bin      # (HPN_BAM_CHUNK / HPN_BAM_ROUNDS are test-hooks switches)
cd /tmp/r05in
t() { for i in 1 2; do s=$(date +%s%N); env "$@" HPN_TIMING=2 HPN_NGPU=1 $B/bam_sliding_count -w 20000 -o s hg38.bam > /dev/null 2> err.txt; e=$(date +%s%N); echo "$* : $(( (e - s) / 1000000 )) ms   $(grep -E 'ingest done|GPU stream open' err.txt | tr '\n' ' ')" >> $out; done; }
t A=1
t HPN_READ_THREADS=4
t HPN_READ_THREADS=8
t HPN_READ_THREADS=12
t HPN_READ_THREADS=16
t HPN_BAM_CHUNK=16777216 HPN_BAM_ROUNDS=44
t HPN_BAM_CHUNK=67108864 HPN_BAM_ROUNDS=11
t HPN_BAM_CHUNK=67108864 HPN_BAM_ROUNDS=11 HPN_READ_THREADS=12
t HPN_BAM_ROUNDS=11
t HPN_BAM_ROUNDS=44
cat $out
rm -rf /tmp/r05in     # (boxes are reused: leave the disk as it was found)
