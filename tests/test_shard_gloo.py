"""CPU, world_size 2 over gloo: the N > 1 path of fastq_count -- record-block shards, one
sum all-reduce of the count vector, min/max re-derived from the reduced histogram.
Per-rank tallies come from the oracle here (no GPU); on GPUs the same vector comes from
hpn_fastq_tally_devptr and the same all-reduce runs over RCCL (bench.py)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    import orc
    from highperformancengs_amd import shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, last = shard.shard_range(n_total, rank, world)
    seq, qual, off = orc.synth_soa(4321, first, last - first, 20, 200)   # counter-based: any shard on its own
    rc, c = orc.count_soa(qual, off)
    s = c.summary()
    v = torch.from_numpy(shard.pack_counts(c.seqlen, s.sum, s.q20, s.q30, qual_hist=c.quality,
                                           nuc_hist=np.zeros((5, 512), np.uint64)))
    shard.allreduce_counts(v)
    if rank == 0:
        q.put(v.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_reduce_to_the_whole():
    import orc
    from highperformancengs_amd import shard
    n_total, world = 3001, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = shard.unpack_counts(q.get(timeout=120))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    seq, qual, off = orc.synth_soa(4321, 0, n_total, 20, 200)
    rc, want = orc.count_soa(qual, off)
    s = want.summary()
    assert np.array_equal(got["seqlen"], want.seqlen) and np.array_equal(got["qual_hist"], want.quality)
    assert (got["total"], got["q20"], got["q30"]) == (s.sum, s.q20, s.q30)
    rep = shard.summarise(got["seqlen"], got["total"], got["q20"], got["q30"])
    assert (rep["reads"], rep["min_len"], rep["max_len"]) == (s.reads, s.min_len, s.max_len) and rep["bases"] == s.bases


def test_shard_ranges_partition_the_records():
    from highperformancengs_amd import shard
    for n in (0, 1, 7, 8, 1000, 10**9 + 3):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_summarise_min_rule():
    # a non-empty length-0 bin never becomes the minimum (statSeqLen, fastq_count.c:63-74)
    from highperformancengs_amd import shard
    h = np.zeros(512, np.uint64)
    h[0], h[4], h[6] = 2, 1, 1
    rep = shard.summarise(h, 10, 10, 8)
    assert (rep["reads"], rep["min_len"], rep["max_len"], rep["bases"]) == (4, 4, 6, 10.0)


# ---- bench.py's rank arithmetic (shard.ShardedTally) driven on CPU with a stub context -----------------------------

class _StubCtx:
    """What ShardedTally needs of api.Context, on the CPU: tallies come from the oracle, the 'native' all-reduce is a
    gloo all-reduce of the context's count vector (where hpn_allreduce_u64 would run RCCL on its stream)."""

    def __init__(self, rank, fail_init):
        self.rank, self.fail_init, self.inits, self.vec = rank, fail_init, 0, None

    def comm_init(self, rank, world, uid):
        assert len(uid) == 128 and uid == bytes(range(128))      # rank 0's id reached this rank
        self.inits += 1
        if self.fail_init:
            raise RuntimeError("no RCCL on this rank")

    def fastq_tally_dev(self, d_qual, d_off, n, flags=0):
        import orc
        from highperformancengs_amd import shard
        rc, c = orc.count_soa(d_qual, d_off)
        s = c.summary()
        self.vec = torch.from_numpy(shard.pack_counts(c.seqlen, s.sum, s.q20, s.q30,
                                                      qual_hist=c.quality if flags else None))

    def tally_devptr(self):
        return self.vec

    def allreduce_u64(self, vec, words):
        assert len(vec) == words
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)

    def fastq_tally_fetch(self, qual_hist=False):
        from highperformancengs_amd import shard

        class R:
            pass
        r, u = R(), shard.unpack_counts(self.vec.numpy())
        r.seqlen, r.total, r.q20, r.q30, r.qual_hist = u["seqlen"], u["total"], u["q20"], u["q30"], u.get("qual_hist")
        return r


def _bench_worker(rank, world, port, n_per_rank, fail_rank, full, q):
    import orc
    from highperformancengs_amd import shard
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = _StubCtx(rank, fail_init=(rank == fail_rank))
    job = shard.ShardedTally(ctx, rank, world, device="cpu", full_matrix=full)
    mode = job.setup(lambda: bytes(range(128)))
    first = shard.weak_shard_first(rank, n_per_rank)
    seq, qual, off = orc.synth_soa(99, first, n_per_rank, 60, 60)
    out = job.step(qual, off, n_per_rank)
    job.check_closed_form(out, n_per_rank, 60)
    slowest = shard.max_over_ranks(1.0 + rank, "cpu")
    q.put((rank, mode, ctx.inits, out["total"], out["q20"], out["q30"], out["seqlen"].copy(),
           out.get("qual_hist", np.zeros(1)).copy(), slowest))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("fail_rank,full,want_mode", [(-1, False, "rccl-native"), (1, False, "torch.distributed"),
                                                      (0, True, "torch.distributed"), (-1, True, "rccl-native")])
def test_bench_rank_arithmetic_two_ranks(fail_rank, full, want_mode):
    """Every rank gets rank 0's unique id, ONE failing hpn_comm_init moves both ranks to the torch.distributed sum (no
    rank is left alone in a collective), both routes give the whole job's counts on every rank, time = slowest rank."""
    import orc
    world, n_per_rank = 2, 700
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_bench_worker, args=(r, world, port, n_per_rank, fail_rank, full, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    seq, qual, off = orc.synth_soa(99, 0, world * n_per_rank, 60, 60)
    rc, want = orc.count_soa(qual, off)
    s = want.summary()
    for rank, mode, inits, total, q20, q30, seqlen, qh, slowest in got:
        assert mode == want_mode and inits == 1
        assert (total, q20, q30) == (s.sum, s.q20, s.q30) and np.array_equal(seqlen, want.seqlen)
        if full:
            assert np.array_equal(qh, want.quality)
        assert slowest == 2.0
