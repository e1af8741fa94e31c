mkdir -p gpurun_out/r06
python bench.py > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err
python3 - <<'PY'
import json
b=json.loads(open("gpurun_out/r06/bench.json").read().strip().splitlines()[-1])
print({k:b[k] for k in ("value","ms_per_step")}, b["roofline"]["frac"], b["roofline"].get("frac_ragged"), b["cpu_baseline"]["value"])
for l in b["extra"]["end_to_end"]:
    h=l.get("hpngs") or {}
    print(l["leg"][:90], h.get("seconds"), l.get("outputs_identical"), l.get("link_frac"))
PY
