# Round 6: the tools on the bench's two compressed inputs -- wall times of round 5's build (build_ab/r05, when it is there) and of
# the tree in the SAME session, every output compared (md5), stamps, rocprofv3 kernel stats.
#   bash scripts/prof_r06_tools.sh [tag]   -> gpurun_out/r06/tools_<tag>.txt, kernel_stats_<tool>_<tag>.csv
# (inputs are made once per box under /tmp/r06in: the 7.2 GB three-member .fastq.gz and the C4-shaped 10.6 GB BAM)
cd $GRAFT_REPO_ROOT
tag=${1:-a}
O=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $O /tmp/r06in
out=$O/tools_$tag.txt
: > $out
if [ ! -f /tmp/r06in/hg38.bam ]; then
python3 - <<'PY'
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import highperformancengs_amd as hp
import bench_extra, c4
if not os.environ.get("NOGZ"):
    ctx = hp.Context(0)
    raw = bench_extra._fastq_text(ctx, 13_000_000, 150, 40).tobytes()
    ctx.close()
    one = bench_extra._gz_single_member(raw, 256, 16)
    with open("/tmp/r06in/gz3.fq.gz", "wb") as f:
        for _ in range(3):
            f.write(one)
tg = c4.targets(lambda n, l: 30.0 if n in ("chr21", "chrM") else 3.0)
c4.synth("/tmp/r06in", "hg38.bam", tg, 15, soa=False)
PY
sync
fi
ls -l /tmp/r06in >> $out
cd /tmp/r06in && export TMPDIR=/tmp
wall() { # label, command...
  l=$1; shift
  for i in 1 2 3; do rm -f d.1.depth s.txt hg38.bam.1.bedGraph; sleep ${PAUSE:-1.5}; s=$(date +%s%N); "$@" > /tmp/r06in/stdout.txt 2> /tmp/r06in/err.txt; e=$(date +%s%N); echo "$l run $i: $(( (e - s) / 1000000 )) ms" >> $out; done
  echo "$l outputs: $(cat /tmp/r06in/stdout.txt d.1.depth s.txt hg38.bam.1.bedGraph 2>/dev/null | md5sum | cut -c1-32)" >> $out
  grep -E "^\[hpn" /tmp/r06in/err.txt | tail -${STAMPS:-6} >> $out
}
for v in r05 tree; do
  B=$GRAFT_REPO_ROOT/highperformancengs_amd/bin
  [ $v = r05 ] && B=$GRAFT_REPO_ROOT/build_ab/r05/bin
  [ -x $B/bam2depth ] || continue
  [ -z "$NOGZ" ] && HPN_TIMING=1 wall "$v fastq_count gz3.fq.gz" $B/fastq_count gz3.fq.gz
  HPN_TIMING=1 HPN_NGPU=1 wall "$v bam2depth" $B/bam2depth -w 20000 -o d hg38.bam
  HPN_TIMING=1 HPN_NGPU=1 wall "$v bam_sliding_count" $B/bam_sliding_count -w 20000 -o s hg38.bam
done
B=$GRAFT_REPO_ROOT/highperformancengs_amd/bin
prof() { n=$1; shift
  HPN_FULL_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r06in/prof_$n -o t -- "$@" > /dev/null 2> /tmp/r06in/$n.err
  cp $(find /tmp/r06in/prof_$n -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${n}_$tag.csv; rm -rf /tmp/r06in/prof_$n
  echo "== rocprofv3 $n (top kernels)" >> $out; head -10 $O/kernel_stats_${n}_$tag.csv | cut -c1-60,100-400 >> $out
}
[ -z "$NOPROF" ] && { [ -z "$NOGZ" ] && prof gz_tool $B/fastq_count gz3.fq.gz; HPN_NGPU=1 prof bam2depth $B/bam2depth -w 20000 -o d hg38.bam; HPN_NGPU=1 prof bam_sliding_count $B/bam_sliding_count -w 20000 -o s hg38.bam; }
cat $out
[ -z "$KEEP_INPUTS" ] && rm -rf /tmp/r06in     # (boxes are reused: leave the disk as it was found)
