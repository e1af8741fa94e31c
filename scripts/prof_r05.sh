# Round 5's evidence, collected in one session on the GPU box:  bash scripts/prof_r05.sh   -> gpurun_out/r05/  (copy to profiles/r05/)
#   bench.json                        the default bench.py line (headline + frac_ragged + extra legs)
#   kernel_stats_*.csv                rocprofv3 --kernel-trace --stats: K1 (headline), the two inflate benchmarks, the other kernels
#   pmc_k1.txt                        FETCH_SIZE / WRITE_SIZE of the headline kernel in separate passes (-> profiles/traffic.json)
#   pmc_inflate.txt                   instruction mix of k_bgzf_inflate and k_gz_sym_inflate (per launch; windows: see the README)
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O /tmp/abw
[ -z "$NOBENCH" ] && python bench.py > $O/bench.json 2> $O/bench.err
g++ -O2 -std=c++17 scripts/bam_synth.cpp -o /tmp/abw/bam_synth -lz -lpthread
[ -f /tmp/abw/a.bam ] || /tmp/abw/bam_synth /tmp/abw/a.bam --targets chr1:120000000:14000000 12 >/dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
prof() {   # name, command...
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o $n -- "$@" > $O/${n}_under_rocprof.txt 2>/dev/null
  cp $O/$n/*/${n}_kernel_stats.csv $O/kernel_stats_$n.csv 2>/dev/null || cp $O/$n/${n}_kernel_stats.csv $O/kernel_stats_$n.csv
  rm -rf $O/$n
}
prof bench_k1 python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline --no-ragged
prof gz_inflate python3 $GRAFT_REPO_ROOT/scripts/bench_gz_inflate.py
prof bgzf_inflate python3 $GRAFT_REPO_ROOT/scripts/bench_inflate.py /tmp/abw/a.bam 4e9 check
prof kernels python3 $GRAFT_REPO_ROOT/scripts/bench_kernels.py 5
cd $GRAFT_REPO_ROOT
timeout 600 python3 scripts/pmc.py k_tally_scan "FETCH_SIZE" "WRITE_SIZE" -- python3 bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-ragged > $O/pmc_k1.txt 2>&1
G1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"
G2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
echo "== k_bgzf_inflate (scripts/bench_inflate.py a.bam: 3.85 GB inflated per launch)" > $O/pmc_inflate.txt
timeout 600 python3 scripts/pmc.py k_bgzf_inflate "$G1" "$G2" -- python3 scripts/bench_inflate.py /tmp/abw/a.bam 4e9 >> $O/pmc_inflate.txt 2>&1
echo "== k_gz_sym_inflate (scripts/bench_gz_inflate.py: 7.57 GB of text per launch)" >> $O/pmc_inflate.txt
timeout 600 python3 scripts/pmc.py k_gz_sym_inflate "$G1" "$G2" -- python3 scripts/bench_gz_inflate.py >> $O/pmc_inflate.txt 2>&1
ls -la $O
rm -rf /tmp/abw
