"""Prints the runs of a c4_full.py / c2_gz_1e9.py JSON one per line:  python scripts/c4_full_show.py profiles/r06/c4_full.json"""
import json, sys
j = json.load(open(sys.argv[1]))
print(j.get("input"), j.get("input_made_in_s"), "identical:", j.get("outputs_identical"), j.get("skipped"))
for r in j.get("runs", []):
    print({k: v for k, v in r.items() if k not in ("kernels", "stderr_tail")})
    k = r.get("kernels")
    if k:
        print("   device_ms", k["device_ms"], [(x["kernel"][:24], x["calls"], x["total_ms"]) for x in k["top"][:5]])
print("targets identical:", sum(t["identical"] for t in j.get("targets", [])), "of", len(j.get("targets", [])))
