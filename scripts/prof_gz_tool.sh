# rocprofv3 --kernel-trace --stats over fastq_count on a single-member .fastq.gz (4.1 GB of text): the gzip route's kernels side by side
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python scripts/e2e_gz_big.py > /tmp/e2e0.txt 2>&1
cd /tmp && export TMPDIR=/tmp
HPN_FULL_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gzt -o t -- $GRAFT_REPO_ROOT/highperformancengs_amd/bin/fastq_count /tmp/big.fq.gz > /dev/null 2>&1
f=$(find /tmp/prof_gzt -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r03/kernel_stats_gz_tool.csv
