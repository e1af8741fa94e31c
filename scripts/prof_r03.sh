# Round 3's evidence, collected in one session on the GPU box -> gpurun_out/r03/ (copy to profiles/r03/)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python bench.py > gpurun_out/r03/bench.json 2> gpurun_out/r03/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/k1 -o k1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03/bench_k1_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/depth -o depth -- python3 $GRAFT_REPO_ROOT/scripts/sweep_only.py 5 > $GRAFT_REPO_ROOT/gpurun_out/r03/sweep_only.txt 2>/dev/null
cd $GRAFT_REPO_ROOT
timeout 500 python3 scripts/pmc.py k_tally_scan "FETCH_SIZE" "WRITE_SIZE" -- python3 bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/r03/pmc_k1.txt 2>&1
timeout 400 python3 scripts/pmc.py k_depth "FETCH_SIZE" "WRITE_SIZE" -- python3 scripts/sweep_only.py 0 > gpurun_out/r03/pmc_depth.txt 2>&1
HPN_LIB=$GRAFT_REPO_ROOT/highperformancengs_amd/diag/libhpngs.so python3 scripts/sweep_only.py 0 2>&1 | grep "^tile" | sort -t' ' -k2n > gpurun_out/r03/sweep_stamps.txt
ls -la gpurun_out/r03
