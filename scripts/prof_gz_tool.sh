# rocprofv3 --kernel-trace --stats over fastq_count on the bench's 7.2 GB three-member .fastq.gz (one context)  -> gpurun_out/r04/kernel_stats_gz_tool.csv
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04 /tmp/gzp
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
import torch
import highperformancengs_amd as hp
import bench_extra
ctx = hp.Context(0)
raw = bench_extra._fastq_text(ctx, 13_000_000, 150, 40).tobytes()
ctx.close()
one = bench_extra._gz_single_member(raw, 256, 16)
with open("/tmp/gzp/gz3.fq.gz", "wb") as f:
    for _ in range(3):
        f.write(one)
PY
cd /tmp/gzp && export TMPDIR=/tmp
HPN_TIMING=2 $GRAFT_REPO_ROOT/highperformancengs_amd/bin/fastq_count gz3.fq.gz 2>&1 | tail -30
HPN_FULL_EXIT=1 HPN_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gzp/prof -o t -- $GRAFT_REPO_ROOT/highperformancengs_amd/bin/fastq_count gz3.fq.gz 2> /tmp/gzp/err.txt > /dev/null
cp $(find /tmp/gzp/prof -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04/kernel_stats_gz_tool.csv
grep "hpn" /tmp/gzp/err.txt | tail -5
