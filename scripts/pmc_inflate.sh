#!/bin/bash
# instruction mix of the two inflate kernels (rocprofv3 --pmc, one pass per group) -> gpurun_out/pmc_inflate.txt
export TMPDIR=/tmp
mkdir -p gpurun_out /tmp/abw
out=gpurun_out/pmc_inflate.txt
G1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SMEM"
G2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
[ -f /tmp/abw/a.bam ] || { g++ -O2 -std=c++17 scripts/bam_synth.cpp -o /tmp/abw/bam_synth -lz -lpthread; /tmp/abw/bam_synth /tmp/abw/a.bam --targets chr1:120000000:14000000 12 >/dev/null 2>&1; }
echo "== k_gz_sym_inflate (scripts/bench_gz_inflate.py)" > $out
timeout 600 python3 scripts/pmc.py k_gz_sym_inflate "$G1" "$G2" -- python3 scripts/bench_gz_inflate.py >> $out 2>&1
echo "== k_bgzf_inflate (scripts/bench_inflate.py a.bam)" >> $out
timeout 600 python3 scripts/pmc.py k_bgzf_inflate "$G1" "$G2" -- python3 scripts/bench_inflate.py /tmp/abw/a.bam 4e9 >> $out 2>&1
