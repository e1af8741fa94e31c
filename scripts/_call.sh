mkdir -p gpurun_out/r06
timeout 1500 python3 scripts/soak_text_lines.py 300 | tee gpurun_out/r06/soak_text_lines.txt
