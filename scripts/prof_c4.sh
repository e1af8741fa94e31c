# rocprofv3 --kernel-trace --stats of bam2depth / bam_sliding_count on the C4-shaped BAM (one worker) -> gpurun_out/r03/kernel_stats_c4_*.csv
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python - <<'PY'
import os, sys, tempfile, time
sys.path.insert(0, "tests")
import c4
os.makedirs("/tmp/c4p", exist_ok=True)
tg = c4.targets(lambda n, l: 30.0 if n in ("chr21", "chrM") else 3.0)
c4.synth("/tmp/c4p", "hg38.bam", tg, 15, soa=False)
PY
cd /tmp/c4p && export TMPDIR=/tmp
for tool in bam2depth bam_sliding_count; do
  HPN_FULL_EXIT=1 HPN_NGPU=1 HPN_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4p/prof_$tool -o t -- $GRAFT_REPO_ROOT/highperformancengs_amd/bin/$tool -w 20000 -o o hg38.bam 2> /tmp/c4p/$tool.err > /dev/null
  f=$(find /tmp/c4p/prof_$tool -name "*kernel_stats.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/r03/kernel_stats_c4_$tool.csv
  grep -v "^chr" /tmp/c4p/$tool.err | tail -12; ls /tmp/c4p | head -20
done
