# Round 5: the plain-text legs -- one 16.3 GB file through fastq_count, eight 4.08 GB files through fastq_count_kthread -t 8 --
# wall times and the tools' own stage lines (HPN_TIMING=2).   bash scripts/prof_r05_plain.sh [tag]  -> gpurun_out/r05/plain_<tag>.txt
cd $GRAFT_REPO_ROOT
tag=${1:-a}
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O /tmp/r05pl
out=$O/plain_$tag.txt
: > $out
if [ ! -f /tmp/r05pl/big.fq ]; then
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch
import highperformancengs_amd as hp
import bench_extra
ctx = hp.Context(0)
for k in range(8):
    raw = bench_extra._fastq_text(ctx, 13_000_000, 150, 40 + k)
    raw.tofile(f"/tmp/r05pl/p{k}.fq")
    if k < 4:
        with open("/tmp/r05pl/big.fq", "ab") as f:
            raw.tofile(f)
ctx.close()
PY
sync
fi
ls -l /tmp/r05pl >> $out
B=$GRAFT_REPO_ROOT/highperformancengs_amd/bin
cd /tmp/r05pl
wall() { l=$1; shift
  for i in 1 2 3; do sleep ${PAUSE:-0}; s=$(date +%s%N); "$@" > /tmp/r05pl/out.txt 2> /tmp/r05pl/err.txt; e=$(date +%s%N); echo "$l run $i: $(( (e - s) / 1000000 )) ms" >> $out; done
  grep -E "^\[hpn" /tmp/r05pl/err.txt | tail -${STAMPS:-40} >> $out; tail -2 /tmp/r05pl/out.txt | cut -c1-200 >> $out
}
HPN_TIMING=2 wall "fastq_count big.fq (16.3 GB)" $B/fastq_count big.fq
HPN_TIMING=2 HPN_NGPU=1 wall "fastq_count big.fq HPN_NGPU=1" $B/fastq_count big.fq
HPN_TIMING=2 wall "fastq_count_kthread -t 8 (8 x 4.08 GB)" $B/fastq_count_kthread -t 8 -o m.tsv p0.fq p1.fq p2.fq p3.fq p4.fq p5.fq p6.fq p7.fq
HPN_NUMA=0 HPN_TIMING=1 wall "HPN_NUMA=0 fastq_count big.fq" $B/fastq_count big.fq
HPN_NUMA=0 HPN_TIMING=1 wall "HPN_NUMA=0 fastq_count_kthread -t 8" $B/fastq_count_kthread -t 8 -o m.tsv p0.fq p1.fq p2.fq p3.fq p4.fq p5.fq p6.fq p7.fq
cat $out
[ -z "$KEEP_INPUTS" ] && rm -rf /tmp/r05in /tmp/r05pl     # (boxes are reused: leave the disk as it was found)
