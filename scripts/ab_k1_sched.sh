# K1's chunk order, same session: pair chunks last (811) against spread among the byte chunks (812), on the headline batch and on
# the batch cut to mixed lengths (bench.py's frac / frac_ragged), plus the uniform 100..151 leg of bench_extra  -> gpurun_out/r05/ab_k1_sched.txt
mkdir -p gpurun_out/r05; out=gpurun_out/r05/ab_k1_sched.txt; : > $out
for v in 811 812 811 812; do
  echo "== HPN_K1_VARIANT=$v" >> $out
  HPN_LIB=$PWD/highperformancengs_amd/testhooks/libhpngs.so HPN_K1_VARIANT=$v python bench.py --no-extra --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('frac', r['frac'], 'kernel_ms', r['kernel_ms'], '| ragged frac', r['frac_ragged'], 'kernel_ms', r['ragged']['kernel_ms'])" >> $out
done
for v in 811 812; do
  echo "== uniform 100..151, HPN_K1_VARIANT=$v" >> $out
  HPN_LIB=$PWD/highperformancengs_amd/testhooks/libhpngs.so HPN_K1_VARIANT=$v python -c "
import sys; sys.path.insert(0,'.')
import torch, highperformancengs_amd as hp, bench_extra
ctx=hp.Context(0); legs=[]; bench_extra._fastq_kernel_legs(ctx,3,legs)
for l in legs: print(l['kernel'][:60], l['kernel_ms'], l['frac'])
" 2>/dev/null >> $out
done
cat $out
