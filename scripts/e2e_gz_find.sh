#!/bin/bash
# the gzip route's wall on one 2.4 GB single-member file: the tool before the producer thread (build_ab/old_tool) and now,
# block starts found by the cores / by the device
python scripts/e2e_gz_big.py ${1:-1.3e7} > /tmp/e2e0.txt 2>&1   # makes /tmp/big.fq.gz
run() { env "$@" 2>&1 | grep -v amdgpu | grep "hpn_gz\|gzip on\|Finished" | cut -c1-230; }
for r in 1; do
  echo "== old tool"; run HPN_TIMING=1 HPN_GZ_DEBUG=1 build_ab/old_tool/fastq_count /tmp/big.fq.gz
  echo "== new, cores search"; run HPN_GZ_FIND=host HPN_TIMING=1 HPN_GZ_DEBUG=1 highperformancengs_amd/testhooks/bin/fastq_count /tmp/big.fq.gz
  echo "== new, device searches"; run HPN_GZ_FIND=device HPN_TIMING=1 HPN_GZ_DEBUG=1 highperformancengs_amd/testhooks/bin/fastq_count /tmp/big.fq.gz
done
