# big plain file: one context vs HPN_NGPU lanes on the one device
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
rec = np.concatenate([names, nl, s, plus, q, nl], axis=1)
with open('/tmp/big.fq','wb') as fh:
    for _ in range(8): fh.write(rec.tobytes())
print('bytes', os.path.getsize('/tmp/big.fq'))
PY
B=highperformancengs_amd/bin
cat /tmp/big.fq > /dev/null
for i in 1 2; do
 HPN_TIMING=1 $B/fastq_count -H /tmp/big.fq 2>&1 | grep -v "^\[hpn\] worker" 
 HPN_NGPU=2 HPN_TIMING=1 $B/fastq_count /tmp/big.fq 2>&1 | grep -v "^\[hpn\] worker"
 HPN_NGPU=4 HPN_TIMING=1 $B/fastq_count /tmp/big.fq 2>&1 | grep -v "^\[hpn\] worker"
done
( time $B/fastq_trim -i /tmp/big.fq -s 5 -e 140 -o /tmp/t1 ) 2>&1 | tail -8
( time HPN_NGPU=3 HPN_TIMING=1 $B/fastq_trim -i /tmp/big.fq -s 5 -e 140 -o /tmp/t3 ) 2>&1 | tail -8
cmp /tmp/t1.trim.fastq /tmp/t3.trim.fastq && echo trim identical
