timeout 1500 python -m pytest tests/test_bam_raw_gpu.py tests/test_bam_gpu.py tests/test_c4_files_gpu.py tests/test_depth_sweep_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "depth or wig or sliding or bam" 2>&1 | tail -4
NOGZ=1 timeout 900 bash scripts/prof_r06_tools.sh c 2>&1 | grep -E "run [0-9]|outputs|k_raw|k_bgzf" | cut -c1-60,100-260
mkdir -p gpurun_out/r06
( time timeout 1500 python bench.py > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err ) 2>&1 | tail -4
tail -5 gpurun_out/r06/bench.err
python - <<'PY'
import json
b=json.loads(open("gpurun_out/r06/bench.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step","roofline")})
for l in b["extra"]["kernel_legs"]:
    print({k:v for k,v in l.items() if k in ("kernel","kernel_ms","frac","frac_bytes_touched","failed","ms_per_launch","add_ms","finish_ms","fields_ms","window_add_ms","records_per_ms_window_add","identical_to_soa_route")})
for l in b["extra"]["end_to_end"]:
    print({k:v for k,v in l.items() if k in ("leg","seconds","gbases_per_s","outputs_identical","link_frac","input_GBps","why")})
PY
