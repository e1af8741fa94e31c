#!/bin/bash
# The host-side readers (two-pass gzip, member-parallel gzip, quick decoder) under ThreadSanitizer and
# AddressSanitizer + UBSan, on sound, multi-member and damaged files.  CPU only (the GPU pool has no sanitizers).
#   bash scripts/sanitize_host.sh        -> prints warnings / errors found (expected: 0 everywhere)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
LINK="-L$ROOT/highperformancengs_amd -lhpngs -lz -lpthread -Wl,-rpath,$ROOT/highperformancengs_amd -Wl,-rpath-link,/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib"
g++ -O1 -g -std=c++17 -DHPN_TEST_HOOKS -fsanitize=thread -I$ROOT/include $ROOT/highperformancengs_amd/csrc/tools/hpn_ingest_dump.cpp -o $T/tsan $LINK
g++ -O1 -g -std=c++17 -DHPN_TEST_HOOKS -fsanitize=address,undefined -I$ROOT/include $ROOT/highperformancengs_amd/csrc/tools/hpn_ingest_dump.cpp -o $T/asan $LINK
python3 - "$T" <<'PY'
import gzip, sys
import numpy as np
T = sys.argv[1]
rng = np.random.default_rng(8)
n = 30000
seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), (n, 100))
qual = rng.integers(35, 74, (n, 100), dtype=np.uint8)
text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, seq[i].tobytes(), qual[i].tobytes()) for i in range(n))
one = gzip.compress(text, 6)
open(T + "/one.fq.gz", "wb").write(one)
open(T + "/multi.fq.gz", "wb").write(b"".join(gzip.compress(text[i:i + 400000], 6) for i in range(0, len(text), 400000)))
for k in range(6):
    b = bytearray(one)
    for _ in range(3):
        p = int(rng.integers(0, len(b)))
        b[p] ^= 1 << int(rng.integers(0, 8))
    if k % 2:
        del b[int(rng.integers(len(b) // 2, len(b))):]
    open(T + f"/bad{k}.gz", "wb").write(b)
# bgzip's container (round 6: the BGZF reader checks length and CRC-32 where the bytes are text), sound and damaged
import struct, zlib
blocks = []
for a in range(0, len(text), 30000):
    piece = text[a:a + 30000]
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(piece) + co.flush()
    blocks.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece)))
eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
open(T + "/sound.bgz", "wb").write(b"".join(blocks) + eof)
for k in range(4):
    bl = [bytearray(x) for x in blocks]
    j = int(rng.integers(1, len(bl) - 1))
    if k == 0:
        bl[j][len(bl[j]) // 2] ^= 8
    elif k == 1:
        bl[j][-4:] = struct.pack("<I", 29999)
    elif k == 2:
        bl[j][-8] ^= 1
    blob = b"".join(bytes(x) for x in bl) + eof
    if k == 3:
        blob = blob[:len(blob) * 2 // 3]
    open(T + f"/badbgz{k}.bgz", "wb").write(blob)
PY
export ASAN_OPTIONS=detect_leaks=0 TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0"
for f in one.fq.gz multi.fq.gz bad0.gz bad1.gz bad2.gz bad3.gz bad4.gz bad5.gz sound.bgz badbgz0.bgz badbgz1.bgz badbgz2.bgz badbgz3.bgz; do
  for env in "HPN_PGZ_FORCE=1 HPN_PGZ_CHUNK=20000 HPN_GZ_THREADS=3" "HPN_PGZ_FORCE=1 HPN_PGZ_CHUNK=30000 HPN_GZ_THREADS=5" "HPN_GZ_THREADS=4"; do
    for s in tsan asan; do
      env $env $T/$s cat $T/$f > /dev/null 2> $T/err.txt || true
      c=$(grep -c "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error" $T/err.txt || true)
      echo "$f [$env] $s: $c"
    done
  done
done
rm -rf "$T"
