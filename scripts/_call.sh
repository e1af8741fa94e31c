mkdir -p gpurun_out/r06
echo "=== soak (damaged streams)"; timeout 900 python3 scripts/soak_inflate_damaged.py 200 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r06/soak_inflate.txt
echo "=== soak (valid streams)"; timeout 900 python3 scripts/soak_inflate.py 2>&1 | grep -v amdgpu.ids | tail -5 | tee -a gpurun_out/r06/soak_inflate.txt
echo "=== scale8 dry run"; SCALE8_GB=4 SCALE8_DEPTH=1 timeout 1500 bash scripts/scale8.sh > gpurun_out/r06/scale8_dry.log 2>&1; tail -3 gpurun_out/r06/scale8_dry.log | cut -c1-300
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/scale8.json"))
print("devices", j["devices"], j["note"]); print(j["rccl_of_the_c_tools"])
for b in j["bench"]: print(b["gpus"], b["rc"], b["wall_s"], (b["line"] or {}).get("value"), (b["line"] or {}).get("config",{}).get("rccl_ranks"))
for f in j["fastq_count"]:
    print(f["input"]); 
    for r in f["runs"]: print("  ", r)
print(json.dumps(j["bam"])[:1500])
PY
