# rocprofv3 --kernel-trace --stats of bam2depth / bam_sliding_count (one worker) on the C4-shaped BAM in both layouts: samtools' (no record
# crosses a BGZF block) and htsjdk's (records packed across blocks, BAM_SYNTH_PACKED=1)   -> gpurun_out/r04/kernel_stats_c4_<layout>_<tool>.csv
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
for layout in aligned packed; do
rm -rf /tmp/c4p && mkdir -p /tmp/c4p
LAYOUT=$layout python - <<'PY'
import os, sys
sys.path.insert(0, "tests")
import c4
tg = c4.targets(lambda n, l: 30.0 if n in ("chr21", "chrM") else 3.0)
c4.synth("/tmp/c4p", "hg38.bam", tg, 15, soa=False, env={"BAM_SYNTH_PACKED": "1"} if os.environ["LAYOUT"] == "packed" else None)
PY
cd /tmp/c4p && export TMPDIR=/tmp
for tool in bam2depth bam_sliding_count; do
  HPN_FULL_EXIT=1 HPN_NGPU=1 HPN_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c4p/prof_$tool -o t -- $GRAFT_REPO_ROOT/highperformancengs_amd/bin/$tool -w 20000 -o o hg38.bam 2> /tmp/c4p/$tool.err > /dev/null
  f=$(find /tmp/c4p/prof_$tool -name "*kernel_stats.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/r04/kernel_stats_c4_${layout}_$tool.csv
  echo "== $layout $tool"; grep -v "^chr" /tmp/c4p/$tool.err | tail -6
done
cd $GRAFT_REPO_ROOT
done
