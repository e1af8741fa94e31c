mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_abi_c.py tests/test_c4_files_gpu.py tests/test_bam_raw_gpu.py -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_cli_gpu.py -q -m gpu -k "depth or wig or sliding" 2>&1 | tail -2
NOGZ=1 NOPROF=1 timeout 900 bash scripts/prof_r06_tools.sh d 2>&1 | grep -E "run [0-9]|outputs" | cut -c1-200
timeout 900 python3 scripts/pmc.py k_raw_starts,k_raw_count,k_raw_scan,k_raw_index,k_raw_fields,k_window_add,k_depth_index,k_depth_sweep "FETCH_SIZE" "WRITE_SIZE" -- python3 scripts/bench_raw_legs.py 1 raw 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/pmc_raw_route.txt
echo "=== c2"; timeout 1200 python3 scripts/c2_gz_1e9.py > gpurun_out/r06/c2_gz_1e9.json 2> gpurun_out/r06/c2.err
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r06/c2_gz_1e9.json"))
print(j["outputs_identical"])
for r in j["runs"]: print(r["seconds"], r["gbases_per_s"], r["row_identical"], r["stderr"][-1])
PY
echo "=== c4"; ( time timeout 2400 python3 scripts/c4_full.py > gpurun_out/r06/c4_full.json 2> gpurun_out/r06/c4.err ) 2>&1 | tail -3
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r06/c4_full.json"))
print(j.get("input"), j.get("outputs_identical"), j.get("skipped"))
for r in j.get("runs", []): print(r["run"], r["seconds"], r.get("first_run_seconds"), r.get("outputs_identical"), (r.get("kernels") or {}).get("top", [])[:2])
PY
