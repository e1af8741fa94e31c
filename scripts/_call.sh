timeout 900 python -m pytest tests/test_gz_inflate_gpu.py tests/test_bgzf_inflate_gpu.py tests/test_bam_raw_gpu.py tests/test_fuzz_gpu.py tests/test_bam_gpu.py -x -q -m gpu 2>&1 | tail -5
mkdir -p gpurun_out/r06_gz
for v in tree finddiag; do lib=""; [ $v != tree ] && lib=$PWD/build_ab/$v/libhpngs.so; echo "== $v"; HPN_LIB=$lib timeout 600 python3 scripts/bench_gz_find.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_gz/bench_gz_find3.txt
NOGZ=1 timeout 900 bash scripts/prof_r06_tools.sh a 2>&1 | tail -70
