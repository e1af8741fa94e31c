# fastq_count_kthread over eight 4.1 GB plain files with 2 .. 8 workers on the one device (each timed run a second behind the one before)
#   -> gpurun_out/r05/sweep_workers.txt     (after KEEP_INPUTS=1 scripts/prof_r05_plain.sh, which makes /tmp/r05pl/p*.fq)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O; out=$PWD/$O/sweep_workers.txt; : > $out
B=$PWD/highperformancengs_amd/bin
cd /tmp/r05pl || exit 1
for t in 2 3 4 8 2 3 4; do
  for i in 1 2; do sleep 1; s=$(date +%s%N); HPN_TIMING=1 $B/fastq_count_kthread -t $t -o m.tsv p0.fq p1.fq p2.fq p3.fq p4.fq p5.fq p6.fq p7.fq > /dev/null 2> err.txt; e=$(date +%s%N); echo "-t $t : $(( (e - s) / 1000000 )) ms  $(grep -c 'copy+frame' err.txt) files, slowest $(grep -o 'copy+frame+tally [0-9.]*' err.txt | sort -k2 -n | tail -1)" >> $out; done
done
rm -rf /tmp/r05pl
cat $out
