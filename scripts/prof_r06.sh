# Round 6's evidence, collected in one session on the GPU box:  bash scripts/prof_r06.sh   -> gpurun_out/r06/  (copy to profiles/r06/)
#   bench.json                        the default bench.py line (headline + frac_ragged + extra legs incl. the raw BAM route)
#   kernel_stats_*.csv                rocprofv3 --kernel-trace --stats: K1 (headline), the BAM legs (SoA and raw route), the text front end on
#                                     1 GiB of device text, the gzip micro-benchmark (decode + histories + translation)
#   pmc_k1.txt                        FETCH_SIZE / WRITE_SIZE of the headline kernel in separate passes (-> profiles/traffic.json)
#   pmc_raw_route.txt                 FETCH_SIZE / WRITE_SIZE per launch of the raw route's kernels (k_raw_*, k_depth_*<RawRecs>, k_window_add)
set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $O
[ -z "$NOBENCH" ] && python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
prof() {   # name, command...
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o $n -- "$@" > $O/${n}_under_rocprof.txt 2>/dev/null
  cp $O/$n/*/${n}_kernel_stats.csv $O/kernel_stats_$n.csv 2>/dev/null || cp $O/$n/${n}_kernel_stats.csv $O/kernel_stats_$n.csv
  rm -rf $O/$n
}
prof bench_k1 python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline --no-ragged
prof bam_legs python3 $GRAFT_REPO_ROOT/scripts/bench_raw_legs.py 2
prof text_inplace python3 $GRAFT_REPO_ROOT/scripts/bench_text_inplace.py 3380000 9
prof gz_inflate python3 $GRAFT_REPO_ROOT/scripts/bench_gz_inflate.py
cd $GRAFT_REPO_ROOT
timeout 600 python3 scripts/pmc.py k_tally_scan "FETCH_SIZE" "WRITE_SIZE" -- python3 bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-ragged > $O/pmc_k1.txt 2>&1
timeout 900 python3 scripts/pmc.py k_raw_starts,k_raw_count,k_raw_scan,k_raw_index,k_raw_fields,k_window_add,k_depth_index,k_depth_sweep,k_depth_tiles,k_depth_scan "FETCH_SIZE" "WRITE_SIZE" -- python3 scripts/bench_raw_legs.py 1 raw > $O/pmc_raw_route.txt 2>&1
ls -la $O
