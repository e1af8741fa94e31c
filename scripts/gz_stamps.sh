# fastq_count on the 7.2 GB .fastq.gz of scripts/prof_gz_tool.sh, six times with HPN_TIMING=2: where a slow run loses its time
cd /tmp/gzp 2>/dev/null || exit 1
for i in 1 2 3 4 5 6; do echo "== run $i"; HPN_TIMING=2 $GRAFT_REPO_ROOT/highperformancengs_amd/bin/fastq_count gz3.fq.gz 2>&1 | grep -E "t=|gzip on"; done
