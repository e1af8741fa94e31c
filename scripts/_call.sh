mkdir -p gpurun_out/r06
timeout 1500 python3 scripts/soak_raw_walk.py 4000 2>&1 | tail -5 | tee gpurun_out/r06/soak_raw_walk.txt
