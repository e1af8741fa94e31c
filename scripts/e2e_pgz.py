#!/usr/bin/env python3
"""Single-member .fastq.gz end to end: the two-pass parallel inflater (pgz_reader.hpp) vs the serial quick decoder vs
zlib inside the same tools, and the reference on the same file.  Same session, best of 2."""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN, REF = os.path.join(ROOT, "highperformancengs_amd", "testhooks", "bin")   # (the build that reads test / timing switches), os.path.join(ROOT, "oracle", "_ref")
L = C.CDLL(os.path.join(ROOT, "oracle", "liborc.so"))
L.orc_synth_write_fastq.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
reads, rl = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000, 150
td = tempfile.mkdtemp(prefix="hpn_e2e_")
one = os.path.join(td, "one.fq.gz")
L.orc_synth_write_fastq(one.encode(), 5, 0, reads, rl, rl, 1)
print(f"{reads} reads x {rl}, {os.path.getsize(one)/1e6:.0f} MB compressed, one gzip member", flush=True)


def t(cmd, env):
    best, out = 1e9, b""
    for _ in range(2):
        t0 = time.perf_counter()
        p = subprocess.run(cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, **env})
        best = min(best, time.perf_counter() - t0)
        out = p.stdout
    return best, out


for tool, args in (("fastq_count", []), ("fastq_count", ["-L"]), ("fastq_trim", ["-s", "5", "-e", "120", "-o", "t", "-i"])):
    outs = []
    for who, exe, env in (("reference", os.path.join(REF, tool), {}),
                          ("hpngs GPU two-pass inflate", os.path.join(BIN, tool), {}),
                          ("hpngs host two-pass (16 thr)", os.path.join(BIN, tool), {"HPN_GZ_GPU": "0"}),
                          ("hpngs host two-pass, 4 thr", os.path.join(BIN, tool), {"HPN_GZ_GPU": "0", "HPN_GZ_THREADS": "4"}),
                          ("hpngs serial decoder", os.path.join(BIN, tool), {"HPN_NO_PGZ": "1"}),
                          ("hpngs zlib", os.path.join(BIN, tool), {"HPN_NO_PGZ": "1", "HPN_FAST_INFLATE": "0"})):
        if not os.access(exe, os.X_OK):
            continue
        dt, out = t([exe] + args + [one], env)
        if "GPU" in who:
            p = subprocess.run([exe] + args + [one], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, **env, "HPN_TIMING": "1"})
            print("      " + " | ".join(l for l in p.stderr.decode().splitlines() if "gzip" in l), flush=True)
        if tool == "fastq_trim":
            out = subprocess.run("md5sum < t.trim.fastq", shell=True, cwd=td, stdout=subprocess.PIPE).stdout
        outs.append(out)
        print(f"{tool + ' ' + ' '.join(args):34s} {who:30s} {dt:7.3f} s  {reads*rl/dt/1e9:6.3f} Gbases/s", flush=True)
    print("   outputs identical:", len(set(outs)) == 1, flush=True)
subprocess.run(["rm", "-rf", td])
