"""TEST-ONLY backend for bench.py (selected with HPN_BENCH_BACKEND=stub.bench_backend:Backend and tests/ on PYTHONPATH): lets
the launcher and the rank arithmetic of bench.py run on a machine without GPUs -- N fresh processes, gloo instead of RCCL, the
ranks' tallies from the oracle.  What it proves: `--gpus N` makes N ranks, rank 0's line says n_gpus = N, the counts are the
whole job's.  It measures nothing (the JSON line carries "backend": "stub-cpu (tests)")."""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

import orc
from highperformancengs_amd import shard


class StubCtx:
    """What shard.ShardedTally and bench.py need of api.Context: the 'native' all-reduce is a gloo all-reduce of the count
    vector, where hpn_allreduce_u64 runs RCCL on the context's stream."""

    def __init__(self):
        self.vec, self.ms = None, 0.0

    def comm_init(self, rank, world, uid):
        assert uid == bytes(range(128))      # rank 0's id reached this rank
        self.world = world

    def fastq_tally_dev(self, d_qual, d_off, n, flags=0):
        t0 = time.perf_counter()
        rc, c = orc.count_soa(d_qual, d_off)
        assert rc == 0
        s = c.summary()
        self.vec = torch.from_numpy(shard.pack_counts(c.seqlen, s.sum, s.q20, s.q30, qual_hist=c.quality if flags else None))
        self.ms = (time.perf_counter() - t0) * 1e3

    def tally_devptr(self):
        return self.vec

    def allreduce_u64(self, vec, words):
        assert len(vec) == words
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)

    def fastq_tally_fetch(self, qual_hist=False):
        class R:
            pass
        r, u = R(), shard.unpack_counts(self.vec.numpy())
        r.seqlen, r.total, r.q20, r.q30, r.qual_hist = u["seqlen"], u["total"], u["q20"], u["q30"], u.get("qual_hist")
        return r

    def last_kernel_ms(self, family=0):
        return self.ms

    def sync(self):
        pass


class Backend:
    name, device, dist_backend = "stub-cpu (tests)", "cpu", "gloo"

    def n_devices(self):
        return int(os.environ.get("HPN_STUB_DEVICES", "8"))

    def open(self, local):
        self.ctx = StubCtx()
        return self.ctx

    def dist_kwargs(self):
        return {}

    def unique_id(self):
        return bytes(range(128))

    def resident_batch(self, n, L, rank):
        first = shard.weak_shard_first(rank, n)
        seq, qual, off = orc.synth_soa(12345, first, n, L, L)
        return qual, np.asarray(off), n, first

    def sync(self):
        pass

    def rccl_ranks(self):
        return dist.get_world_size() if dist.is_initialized() else 1
