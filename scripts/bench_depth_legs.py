"""The BAM kernel legs of bench_extra.py alone (K3, K4, the sweep, bedGraph text, K5): quick A/B runs."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import highperformancengs_amd as hp
import bench_extra

ctx = hp.Context(0)
only = "--fastq" not in sys.argv
legs = bench_extra.kernel_legs(ctx, reps=5, fastq=not only) if "fastq" in bench_extra.kernel_legs.__code__.co_varnames else bench_extra.kernel_legs(ctx)
for l in legs:
    print(json.dumps(l))
