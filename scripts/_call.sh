mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_bam_raw_gpu.py tests/test_bam_gpu.py tests/test_c4_files_gpu.py -q -m gpu 2>&1 | tail -3
NOGZ=1 timeout 900 bash scripts/prof_r06_tools.sh g 2>&1 | grep -E "run [0-9]|outputs|k_raw_count|k_raw_index" | cut -c1-60,100-230
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o pl -- python3 $GRAFT_REPO_ROOT/scripts/bench_raw_legs.py 1 raw > /tmp/pl.txt 2>/dev/null; grep -h "k_raw" $(find /tmp/pl -name "*kernel_stats.csv") | cut -c1-40,200-330
