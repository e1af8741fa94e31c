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
