#!/usr/bin/env python3
"""The BAM kernel legs of bench.py on their own (bench_extra._bam_kernel_legs: the SoA legs and the raw-record route the tools
take), for rocprofv3 / PMC passes:   python3 scripts/bench_raw_legs.py [reps] [raw]   -> one JSON object per leg ("raw": the raw-record legs alone)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import highperformancengs_amd as hp  # noqa: E402
import bench_extra  # noqa: E402

ctx = hp.Context(0)
legs = []
bench_extra._bam_kernel_legs(ctx, int(sys.argv[1]) if len(sys.argv) > 1 else 3, legs, raw_only=len(sys.argv) > 2 and sys.argv[2] == "raw")
for l in legs:
    print(json.dumps(l), flush=True)
ctx.close()
