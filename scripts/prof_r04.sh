# Round 4's evidence, collected in one session on the GPU box:  bash scripts/prof_r04.sh   -> gpurun_out/r04/  (copy to profiles/r04/)
#   bench.json                        the default bench.py line (headline + extra legs)
#   kernel_stats_*.csv                rocprofv3 --kernel-trace --stats for every kernel a bench leg quotes: K1 (headline), K1L / K1L + Nucleotide /
#                                     K2 / K3 + K4 sweep / K5 (bench_kernels.py), k_bedgraph_text + K3 / K4 two-pass (bench_depth_legs.py),
#                                     k_gz_sym_inflate / k_gz_windows_lds / k_gz_translate (bench_gz_inflate.py), k_bgzf_inflate (bench_inflate.py)
#   pmc_k1.txt                        FETCH_SIZE / WRITE_SIZE of the headline kernel in separate passes (profiles/traffic.json)
#   pmc_k5.txt, pmc_inflate.txt       instruction mix and wait shares
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O /tmp/abw
python bench.py > $O/bench.json 2> $O/bench.err
g++ -O2 -std=c++17 scripts/bam_synth.cpp -o /tmp/abw/bam_synth -lz -lpthread
/tmp/abw/bam_synth /tmp/abw/a.bam --targets chr1:120000000:14000000 12 >/dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
prof() {   # name, command...
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o $n -- "$@" > $O/${n}_under_rocprof.txt 2>/dev/null
  cp $O/$n/*/${n}_kernel_stats.csv $O/kernel_stats_$n.csv 2>/dev/null || cp $O/$n/${n}_kernel_stats.csv $O/kernel_stats_$n.csv
  rm -rf $O/$n
}
prof bench_k1 python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline
prof kernels python3 $GRAFT_REPO_ROOT/scripts/bench_kernels.py 5
prof depth_legs python3 $GRAFT_REPO_ROOT/scripts/bench_depth_legs.py
prof gz_inflate python3 $GRAFT_REPO_ROOT/scripts/bench_gz_inflate.py
prof bgzf_inflate python3 $GRAFT_REPO_ROOT/scripts/bench_inflate.py /tmp/abw/a.bam 4e9 check
cd $GRAFT_REPO_ROOT
timeout 600 python3 scripts/pmc.py k_tally_scan "FETCH_SIZE" "WRITE_SIZE" -- python3 bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $O/pmc_k1.txt 2>&1
timeout 400 python3 scripts/pmc.py k_window_add,k_window_rest "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES" "FETCH_SIZE" -- python3 scripts/bench_kernels.py 3 k5 > $O/pmc_k5.txt 2>&1
timeout 600 python3 scripts/pmc.py k_bgzf_inflate "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_LDS" -- python3 scripts/bench_inflate.py /tmp/abw/a.bam 4e9 > $O/pmc_inflate.txt 2>&1
ls -la $O
