#!/usr/bin/env python3
"""Writes scripts/README.md: every file here with the first line of its header, grouped.   python3 scripts/mk_index.py"""
import os

d = os.path.dirname(os.path.abspath(__file__))


def desc(f):
    txt = open(os.path.join(d, f), errors="replace").read().splitlines()
    out = ""
    for i, l in enumerate(txt[:12]):
        s = l.strip()
        if s.startswith("#!"):
            continue
        if s.startswith('"""'):
            out = s.strip('"').strip()
            if not out and i + 1 < len(txt):
                out = txt[i + 1].strip()
            break
        if s.startswith("#") or s.startswith("//"):
            out = s.lstrip("#/ ").strip()
            if out:
                break
    out = out.replace("|", "/")
    return (out[:177] + "...") if len(out) > 180 else out


groups = [
    ("Evidence of the current round (what `profiles/r06/` was made with)",
     ["prof_r06.sh", "prof_r06_tools.sh", "prof_r06_plain.sh", "r06_gz.sh", "r06_gz2.sh", "pmc.py", "bench_raw_legs.py", "bench_text_inplace.py",
      "bench_gz_find.py", "bench_gz_inflate.py", "bench_inflate.py", "sweep_only.py"]),
    ("BASELINE configurations at their stated sizes", ["c2_gz_1e9.py", "c4_full.py", "c4_full_show.py", "c4_30x_big.py", "scale8.sh", "scale8.py"]),
    ("Same-session A/B of variant builds",
     ["ab_variant.sh", "ab_inflate.sh", "ab_bgzf.sh", "ab_sweep.sh", "ab_k5.sh", "ab_depth.py", "ab_gz_route.sh", "ab_gz_windows.sh", "ab_k1_sched.sh",
      "ab_waves.sh", "ab_build.sh"]),
    ("Soaks and sanitizers (beyond the test suite)",
     ["soak_inflate.py", "soak_inflate_damaged.py", "soak_raw_walk.py", "soak_text_lines.py", "sanitize_host.sh", "sanitize_shard.sh", "emu_raw_chain.py"]),
    ("Inputs and housekeeping", ["bam_synth.cpp", "mk_index.py"]),
]
seen = set(sum((g[1] for g in groups), []))
rest = [f for f in sorted(os.listdir(d)) if os.path.isfile(os.path.join(d, f)) and f not in seen and f not in ("_call.sh", "README.md")]
out = ["# scripts/", "",
       "Nothing here is part of the product (`highperformancengs_amd/`) or of the tests; these are the measurement, A/B and soak programs",
       "the files under `profiles/rNN/` were made with, kept so that every figure in DESIGN.md and `docs/kernels/` can be run again.",
       "`scripts/micro/` (its own README) holds the single-question micro-benchmarks.  They run on the GPU box from the repository",
       "root (`gpurun -- 'bash scripts/<name>.sh'`); variant libraries go to `build_ab/<name>/libhpngs.so` (`ab_variant.sh`; `ab_build.sh`",
       "is its predecessor and builds every source) and are selected with `HPN_LIB=`.  This index: `python3 scripts/mk_index.py`.", ""]
for title, files in groups:
    out += ["## " + title, "", "| file | what |", "|---|---|"]
    out += [f"| `{f}` | {desc(f) or '(see its header)'} |" for f in files]
    out.append("")
out += ["## Probes of earlier rounds (cited by `profiles/r01` ... `r05` and the per-kernel pages)", "", "| file | what |", "|---|---|"]
out += [f"| `{f}` | {desc(f) or '(see its header)'} |" for f in rest]
out.append("")
open(os.path.join(d, "README.md"), "w").write("\n".join(out))
