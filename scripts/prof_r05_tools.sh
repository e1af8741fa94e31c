# Round 5: the tools on the bench's two compressed inputs -- stamps (HPN_TIMING=2), wall times, rocprofv3 kernel stats.
#   bash scripts/prof_r05_tools.sh [tag]   -> gpurun_out/r05/tools_<tag>.txt, kernel_stats_<tool>_<tag>.csv
# (inputs are made once per box under /tmp/r05in: the 7.2 GB three-member .fastq.gz and the C4-shaped 10.6 GB BAM)
cd $GRAFT_REPO_ROOT
tag=${1:-a}
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O /tmp/r05in
out=$O/tools_$tag.txt
: > $out
if [ ! -f /tmp/r05in/gz3.fq.gz ]; then
python - <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import highperformancengs_amd as hp
import bench_extra, c4
ctx = hp.Context(0)
raw = bench_extra._fastq_text(ctx, 13_000_000, 150, 40).tobytes()
ctx.close()
one = bench_extra._gz_single_member(raw, 256, 16)
with open("/tmp/r05in/gz3.fq.gz", "wb") as f:
    for _ in range(3):
        f.write(one)
tg = c4.targets(lambda n, l: 30.0 if n in ("chr21", "chrM") else 3.0)
c4.synth("/tmp/r05in", "hg38.bam", tg, 15, soa=False)
PY
sync
fi
ls -l /tmp/r05in >> $out
B=$GRAFT_REPO_ROOT/highperformancengs_amd/bin
cd /tmp/r05in && export TMPDIR=/tmp
wall() { # label, command...
  l=$1; shift
  for i in 1 2 3; do rm -f d.1.depth s.txt hg38.bam.1.bedGraph; sleep ${PAUSE:-0}; s=$(date +%s%N); "$@" > /dev/null 2> /tmp/r05in/err.txt; e=$(date +%s%N); echo "$l run $i: $(( (e - s) / 1000000 )) ms" >> $out; done
  grep -E "^\[hpn" /tmp/r05in/err.txt | tail -${STAMPS:-40} >> $out
}
HPN_TIMING=2 wall "fastq_count gz3.fq.gz" $B/fastq_count gz3.fq.gz
HPN_TIMING=2 HPN_NGPU=1 wall "bam2depth" $B/bam2depth -w 20000 -o d hg38.bam
HPN_TIMING=2 HPN_NGPU=1 wall "bam_sliding_count" $B/bam_sliding_count -w 20000 -o s hg38.bam
prof() { n=$1; shift
  HPN_FULL_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05in/prof_$n -o t -- "$@" > /dev/null 2> /tmp/r05in/$n.err
  cp $(find /tmp/r05in/prof_$n -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${n}_$tag.csv; rm -rf /tmp/r05in/prof_$n
  echo "== rocprofv3 $n (top kernels)" >> $out; head -8 $O/kernel_stats_${n}_$tag.csv | cut -c1-60,100-400 >> $out
}
[ -z "$NOPROF" ] && { prof gz_tool $B/fastq_count gz3.fq.gz; HPN_NGPU=1 prof bam2depth $B/bam2depth -w 20000 -o d hg38.bam; HPN_NGPU=1 prof bam_sliding_count $B/bam_sliding_count -w 20000 -o s hg38.bam; }
cat $out
[ -z "$KEEP_INPUTS" ] && rm -rf /tmp/r05in /tmp/r05pl     # (boxes are reused: leave the disk as it was found)
