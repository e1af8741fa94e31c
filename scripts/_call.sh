AB_STRETCHES=6144:6144 bash scripts/ab_inflate.sh before
grep -v "^Traceback\|^  File\|^    \|amdgpu.ids" gpurun_out/ab_inflate.txt | cut -c1-330
timeout 1500 python -m pytest tests/test_gz_inflate_gpu.py tests/test_bgzf_inflate_gpu.py -x -q -m gpu 2>&1 | tail -3
