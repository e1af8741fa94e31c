# one big plain FASTQ from the page cache: how many pread threads feed one PCIe link best?
set -e
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import numpy as np, os, sys
sys.path.insert(0,'tests'); import orc
n, L = 2000000, 150
seq, qual, off = orc.synth_soa(99, 0, n, L, L)
s, q = seq.reshape(n, L), qual.reshape(n, L)
nl = np.full((n,1), 10, np.uint8)
names = np.frombuffer(b"".join(b"@r%09d" % i for i in range(n)), np.uint8).reshape(n, 11)
plus = np.tile(np.frombuffer(b"\n+\n", np.uint8), (n,1))
rec = np.concatenate([names, nl, s, plus, q, nl], axis=1).tobytes()
with open('/tmp/big.fq','wb') as fh:
    for _ in range(24): fh.write(rec)
print('bytes', os.path.getsize('/tmp/big.fq'))
PY
B=highperformancengs_amd/testhooks/bin
cat /tmp/big.fq > /dev/null
for t in 4 6 8 12 16; do
  for c in 33554432 67108864; do
    echo "== read threads $t chunk $c"
    HPN_READ_THREADS=$t HPN_TEXT_CHUNK=$c HPN_TIMING=1 $B/fastq_count /tmp/big.fq 2>&1 | grep -E "hpn\] /tmp|Finished"
  done
done
echo "== 2 lanes, 8 threads"; HPN_NGPU=2 HPN_READ_THREADS=8 HPN_TIMING=1 $B/fastq_count /tmp/big.fq 2>&1 | grep -E "one input|Finished"
