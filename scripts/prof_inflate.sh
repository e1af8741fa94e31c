# rocprofv3 --kernel-trace --stats of the two inflate benchmarks -> gpurun_out/r03/kernel_stats_{gz,bgzf}_inflate.csv
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03 /tmp/abw
[ -f /tmp/abw/a.bam ] || { g++ -O2 -std=c++17 scripts/bam_synth.cpp -o /tmp/abw/bam_synth -lz -lpthread; /tmp/abw/bam_synth /tmp/abw/a.bam --targets chr1:120000000:14000000 12 >/dev/null 2>&1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gz -o gz -- python3 $GRAFT_REPO_ROOT/scripts/bench_gz_inflate.py > $GRAFT_REPO_ROOT/gpurun_out/r03/bench_gz_inflate_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bgzf -o bgzf -- python3 $GRAFT_REPO_ROOT/scripts/bench_inflate.py /tmp/abw/a.bam 4e9 > $GRAFT_REPO_ROOT/gpurun_out/r03/bench_bgzf_inflate_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
for n in gz bgzf; do f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1); head -1 $f > gpurun_out/r03/kernel_stats_${n}_inflate.csv; grep "hpn::" $f >> gpurun_out/r03/kernel_stats_${n}_inflate.csv; done
ls -la gpurun_out/r03
