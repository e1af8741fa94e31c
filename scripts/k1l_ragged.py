"""K1 / K1L on ragged reads (lengths 100..151): wall and kernel time of one pass with and without the Quality matrix, totals checked (probe)."""
import sys, time
sys.path.insert(0, ".")
import torch, numpy as np
import highperformancengs_amd as hp
ctx = hp.Context(0)
for n in (2_000_000, 20_000_000):
    g0 = torch.Generator(device="cuda").manual_seed(11)
    rl = torch.randint(100, 152, (n,), device="cuda", generator=g0, dtype=torch.int64)
    ro = torch.zeros(n + 1, dtype=torch.int64, device="cuda"); torch.cumsum(rl, 0, out=ro[1:])
    tot = int(ro[-1].item())
    dq = torch.randint(35, 75, (tot,), device="cuda", generator=g0, dtype=torch.uint8)
    for fl in (0, 1):
        t0 = time.time(); ctx.fastq_tally_dev(dq, ro, n, flags=fl); r = ctx.fastq_tally_fetch(qual_hist=bool(fl)); dt = time.time() - t0
        print(n, "flags", fl, "wall %.3f s" % dt, "kernel %.3f ms" % ctx.last_kernel_ms(0), r.total == tot)
