#!/bin/bash
# The host side of "one FASTQ over several lanes" (csrc/host/text_shard.hpp: reader, dispatcher, lane threads, the board of line
# counts, the ordered writer) under ThreadSanitizer and AddressSanitizer + UBSan.  CPU only: the few ABI calls the route makes
# are stood in for by tests/stub/shard_harness.cpp, everything else is the product's code.
#   bash scripts/sanitize_shard.sh      -> one line per run: "<what>: <reports> sanitizer reports, result ok|WRONG" (expected: 0, ok)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
g++ -O1 -g -std=c++17 -DHPN_TEST_HOOKS -fsanitize=thread -I$ROOT/include $ROOT/tests/stub/shard_harness.cpp -o $T/tsan -lz -lpthread
g++ -O1 -g -std=c++17 -DHPN_TEST_HOOKS -fsanitize=address,undefined -I$ROOT/include $ROOT/tests/stub/shard_harness.cpp -o $T/asan -lz -lpthread
python3 - "$T" <<'PY'
import gzip, sys
import numpy as np
T = sys.argv[1]
rng = np.random.default_rng(3)
recs = []
for i in range(6000):
    l = int(rng.integers(0, 200))
    recs.append(b"@r%d %s\n%s\n+\n%s\n" % (i, bytes(rng.integers(48, 123, int(rng.integers(0, 30)), dtype=np.uint8)),
                                              bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), l)), bytes(rng.integers(33, 75, l, dtype=np.uint8))))
text = b"".join(recs)
open(T + "/a.fq", "wb").write(text)
open(T + "/a.fq.gz", "wb").write(b"".join(gzip.compress(text[i:i + 300000], 6) for i in range(0, len(text), 300000)))
open(T + "/nonl.fq", "wb").write(text[:-1])
open(T + "/trunc.fq", "wb").write(text[:len(text) // 2 + 17])
big = b"@big\n" + b"A" * 100000 + b"\n+\n" + b"I" * 100000 + b"\n"
open(T + "/big.fq", "wb").write(b"".join(recs[:3000]) + big + b"".join(recs[3000:]))
lens = [len(r.split(b"\n")[1]) for r in recs]
open(T + "/want.txt", "w").write("%d %d\n" % (len(recs), sum(lens)))
S, E = 3, 90
open(T + "/want.trim", "wb").write(b"".join(b"\n".join([r.split(b"\n")[0], r.split(b"\n")[1][S:E], b"+", r.split(b"\n")[3][S:E], b""]) for r in recs))
PY
export ASAN_OPTIONS=detect_leaks=0 TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0"
read WR WB < $T/want.txt
for s in tsan asan; do
  for lanes in 2 3 5; do
    for chunk in 8192 20000 65536; do
      for f in a.fq a.fq.gz nonl.fq; do
        out=$(HPN_TEXT_CHUNK=$chunk $T/$s count $T/$f $lanes 2> $T/err.txt || true)
        c=$(grep -c "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error" $T/err.txt || true)
        # a final line without its newline may be handed to a lane that is not the last: then the route says "irregular" (1) and adds nothing
        ok=WRONG; [ "$out" = "0 0 $WR $WB" ] && ok=ok; [ "$f" = nonl.fq ] && [ "$out" = "0 1 0 0" ] && ok=ok
        echo "count $f lanes=$lanes chunk=$chunk $s: $c sanitizer reports, result $ok ($out)"
      done
      out=$(HPN_TEXT_CHUNK=$chunk $T/$s count $T/trunc.fq $lanes 2> $T/err.txt || true)
      c=$(grep -c "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error" $T/err.txt || true)
      ok=WRONG; [ "$out" = "0 1 0 0" ] && ok=ok
      echo "count trunc.fq (must be abandoned) lanes=$lanes chunk=$chunk $s: $c sanitizer reports, result $ok ($out)"
      out=$(HPN_TEXT_CHUNK=$chunk $T/$s trim $T/a.fq $lanes 3 90 $T/got.trim 2> $T/err.txt || true)
      c=$(grep -c "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error" $T/err.txt || true)
      ok=WRONG; cmp -s $T/got.trim $T/want.trim && [ "$out" = "0 0 $WR" ] && ok=ok
      echo "trim a.fq lanes=$lanes chunk=$chunk $s: $c sanitizer reports, result $ok ($out)"
    done
    # one record longer than a piece's tail in mid-file: its lane finds the piece irregular while the others wait for output slabs queued
    # behind it -- the route must come back ("0 1 ...") and not hang, with the irregular lane slowed too
    for slow in 0 300; do
      out=$(HPN_STUB_IRREGULAR_SLEEP_MS=$slow HPN_TEXT_CHUNK=65536 timeout 120 $T/$s trim $T/big.fq $lanes 3 90 $T/got.trim 2> $T/err.txt || echo HUNG)
      c=$(grep -c "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error" $T/err.txt || true)
      ok=WRONG; [ "${out:0:3}" = "0 1" ] && ok=ok
      echo "trim big.fq (must be abandoned, not hang) lanes=$lanes slow=$slow $s: $c sanitizer reports, result $ok ($out)"
    done
  done
done
rm -rf $T
