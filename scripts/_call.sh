mkdir -p gpurun_out/r06
timeout 1500 python3 scripts/c2_gz_1e9.py > gpurun_out/r06/c2_gz_1e9_b.json 2> gpurun_out/r06/c2_gz_1e9_b.err
tail -c 1500 gpurun_out/r06/c2_gz_1e9_b.json; tail -5 gpurun_out/r06/c2_gz_1e9_b.err
rm -rf /dev/shm/c2gz_*
