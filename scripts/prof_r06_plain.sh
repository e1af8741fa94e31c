# Round 6: the plain-text legs -- one 16.3 GB file through fastq_count, eight 4.08 GB files through fastq_count_kthread -t 8 --
# wall times and the tools' own stage lines (HPN_TIMING=2).   bash scripts/prof_r06_plain.sh [tag]  -> gpurun_out/r06/plain_<tag>.txt
cd $GRAFT_REPO_ROOT
tag=${1:-a}
O=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $O /tmp/r06pl
out=$O/plain_$tag.txt
: > $out
if [ ! -f /tmp/r06pl/big.fq ]; then
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch
import highperformancengs_amd as hp
import bench_extra
ctx = hp.Context(0)
for k in range(8):
    raw = bench_extra._fastq_text(ctx, 13_000_000, 150, 40 + k)
    raw.tofile(f"/tmp/r06pl/p{k}.fq")
    if k < 4:
        with open("/tmp/r06pl/big.fq", "ab") as f:
            raw.tofile(f)
ctx.close()
PY
sync
fi
ls -l /tmp/r06pl >> $out
B=$GRAFT_REPO_ROOT/highperformancengs_amd/bin
cd /tmp/r06pl
wall() { l=$1; shift
  for i in 1 2 3; do sleep ${PAUSE:-0}; s=$(date +%s%N); "$@" > /tmp/r06pl/out.txt 2> /tmp/r06pl/err.txt; e=$(date +%s%N); echo "$l run $i: $(( (e - s) / 1000000 )) ms" >> $out; done
  grep -E "^\[hpn" /tmp/r06pl/err.txt | tail -${STAMPS:-40} >> $out; tail -2 /tmp/r06pl/out.txt | cut -c1-200 >> $out
}
for v in r05 tree; do
  B=$GRAFT_REPO_ROOT/highperformancengs_amd/bin
  [ $v = r05 ] && B=$GRAFT_REPO_ROOT/build_ab/r05/bin
  [ -x $B/fastq_count ] || continue
  PAUSE=1.5 HPN_TIMING=1 wall "$v fastq_count big.fq (16.3 GB)" $B/fastq_count big.fq
  PAUSE=1.5 HPN_TIMING=1 wall "$v fastq_count_kthread -t 8 (8 x 4.08 GB)" $B/fastq_count_kthread -t 8 -o m.tsv p0.fq p1.fq p2.fq p3.fq p4.fq p5.fq p6.fq p7.fq
done
B=$GRAFT_REPO_ROOT/highperformancengs_amd/bin
# the eight-file run with every stage stamped (the timeline the review asked for)
HPN_TIMING=2 $B/fastq_count_kthread -t 8 -o m.tsv p0.fq p1.fq p2.fq p3.fq p4.fq p5.fq p6.fq p7.fq > /dev/null 2> $O/plain8_timeline.txt
cat $out
[ -z "$KEEP_INPUTS" ] && rm -rf /tmp/r06pl     # (boxes are reused: leave the disk as it was found)
