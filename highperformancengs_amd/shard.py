"""Record-block sharding across the GPUs of one node and the final reduction.

The reference is single-process; its nearest seam is reduceStats
(fastq_count_kthread.c:180-210): an element-wise sum of per-file accumulators.
Here the units are contiguous record blocks, one per rank; every
rank tallies its block into a private count vector and ONE sum all-reduce of
that small vector (515 words; 68,611 with the matrices) finishes the job.  No
data-path collective exists: records never move between GPUs.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def shard_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """[first, last) record block of `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def pack_counts(seqlen, total, q20, q30, qual_hist=None, nuc_hist=None) -> np.ndarray:
    """Count vector in the HPN_TALLY_W_* layout of include/hpngs.h (int64 for torch)."""
    n = _lib.W_BAD + 1 if qual_hist is None and nuc_hist is None else _lib.TALLY_WORDS
    v = np.zeros(n, np.int64)
    v[_lib.W_SEQLEN:_lib.W_SEQLEN + _lib.LEN_BINS] = np.asarray(seqlen).astype(np.int64)
    v[_lib.W_TOTAL], v[_lib.W_Q20], v[_lib.W_Q30] = int(total), int(q20), int(q30)
    if qual_hist is not None:
        v[_lib.W_QUAL:_lib.W_NUC] = np.asarray(qual_hist).reshape(-1).astype(np.int64)
    if nuc_hist is not None:
        v[_lib.W_NUC:_lib.TALLY_WORDS] = np.asarray(nuc_hist).reshape(-1).astype(np.int64)
    return v


def unpack_counts(v: np.ndarray) -> dict:
    v = np.asarray(v)
    out = {"seqlen": v[_lib.W_SEQLEN:_lib.W_SEQLEN + _lib.LEN_BINS].astype(np.uint64),
           "total": int(v[_lib.W_TOTAL]), "q20": int(v[_lib.W_Q20]), "q30": int(v[_lib.W_Q30])}
    if len(v) >= _lib.TALLY_WORDS:
        out["qual_hist"] = v[_lib.W_QUAL:_lib.W_NUC].reshape(_lib.QUAL_ROWS, _lib.LEN_BINS).astype(np.uint64)
        out["nuc_hist"] = v[_lib.W_NUC:_lib.TALLY_WORDS].reshape(_lib.NUC_CODES, _lib.LEN_BINS).astype(np.uint64)
    return out


def allreduce_counts(vec, group=None):
    """Sum a count vector over all ranks with torch.distributed (RCCL on GPUs, gloo on CPU).
    `vec` is a torch int64 tensor; reduced in place and returned."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    return vec


def summarise(seqlen, total, q20, q30) -> dict:
    """statSeqLen + the report columns of fastq_count.c:63-74,127 from reduced counts
    (min/max are re-derived from the reduced histogram: no min/max collective)."""
    seqlen = np.asarray(seqlen, np.uint64)
    nz = np.nonzero(seqlen)[0]
    reads = int(seqlen.sum())
    bases = float(sum(float(seqlen[l]) * float(l) for l in nz))
    nz_pos = nz[nz > 0]
    return {"reads": reads, "bases": bases,
            "min_len": int(nz_pos[0]) if len(nz_pos) else 0,  # a non-empty bin 0 never becomes the minimum
            "max_len": int(nz[-1]) if len(nz) else 0,
            "q20_pct": 100.0 * q20 / total if total else float("nan"),
            "q30_pct": 100.0 * q30 / total if total else float("nan")}


# ---- the N-rank job as bench.py runs it (and as tests/test_shard_gloo.py drives it on CPU with a stub context) ----

def weak_shard_first(rank: int, n_per_rank: int) -> int:
    """Global index of the first record of `rank`'s block when every rank holds n_per_rank records
    (weak scaling, BASELINE configs[4]: 8 x 1e9); the counter-based generator makes any block on its own."""
    return rank * n_per_rank


def max_over_ranks(seconds: float, device="cpu") -> float:
    """The job's time is its slowest rank's."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(x: float, device="cpu") -> list:
    """[x of rank 0, x of rank 1, ...] on every rank (bench.py: each rank's mean kernel time)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [float(x)]
    mine = torch.tensor([x], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


class ShardedTally:
    """One rank's side of the sharded fastq_count: tally the resident block, sum the count vector over the ranks,
    fetch it.  The sum is ONE all-reduce of W_BAD+1 (or TALLY_WORDS) u64 words: on the context's stream through the
    library's own RCCL binding (hpn_comm_init / hpn_allreduce_u64) when every rank could initialise it, else through
    torch.distributed on the fetched vector.  `ctx` is an api.Context (or, in the CPU tests, an object with the same
    five methods); `device` is where the small voting / fallback tensors live ("cuda" on GPUs, "cpu" under gloo)."""

    def __init__(self, ctx, rank: int, world: int, device="cuda", full_matrix: bool = False):
        self.ctx, self.rank, self.world, self.device, self.full = ctx, rank, world, device, full_matrix
        self.allreduce = "none"
        self.flags = _lib.TALLY_QUAL_HIST if full_matrix else 0
        self.words = _lib.TALLY_WORDS if full_matrix else _lib.W_BAD + 1

    def setup(self, unique_id_fn) -> str:
        """Agree on the all-reduce: rank 0's ncclUniqueId goes to every rank, each rank tries hpn_comm_init, and ONE
        failure anywhere (MIN vote) moves every rank to torch.distributed, so that no rank waits in a collective
        the others never enter."""
        import torch
        import torch.distributed as dist
        if self.world <= 1:
            return self.allreduce
        mine = 1
        try:
            uid = torch.zeros(_lib.UNIQUE_ID_BYTES, dtype=torch.uint8, device=self.device)
            if self.rank == 0:
                uid.copy_(torch.frombuffer(bytearray(unique_id_fn()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            self.ctx.comm_init(self.rank, self.world, bytes(uid.cpu().numpy().tobytes()))
        except Exception as e:  # noqa: BLE001
            mine = 0
            self.why = str(e)
        ok = torch.tensor([mine], device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        self.allreduce = "rccl-native" if int(ok.item()) == 1 else "torch.distributed"
        return self.allreduce

    def step(self, d_qual, d_off, n: int) -> dict:
        """One pass over the resident block + the reduction: the job's counts (identical on every rank)."""
        import torch
        self.ctx.fastq_tally_dev(d_qual, d_off, n, flags=self.flags)
        if self.allreduce == "rccl-native":
            self.ctx.allreduce_u64(self.ctx.tally_devptr(), self.words)
        res = self.ctx.fastq_tally_fetch(qual_hist=self.full)
        if self.allreduce == "torch.distributed":
            v = torch.from_numpy(pack_counts(res.seqlen, res.total, res.q20, res.q30,
                                             qual_hist=res.qual_hist if self.full else None)).to(self.device)
            allreduce_counts(v)
            return unpack_counts(v.cpu().numpy())
        out = {"seqlen": np.array(res.seqlen, np.uint64), "total": int(res.total), "q20": int(res.q20), "q30": int(res.q30)}
        if self.full:
            out["qual_hist"] = np.array(res.qual_hist, np.uint64)
        return out

    def check_closed_form(self, out: dict, n_per_rank: int, read_len: int):
        """What every rank's block of the synthetic generator must add up to (exact)."""
        assert out["total"] == self.world * n_per_rank * read_len, (out["total"], self.world, n_per_rank, read_len)
        assert int(out["seqlen"][read_len]) == self.world * n_per_rank
        assert int(np.asarray(out["seqlen"]).sum()) == self.world * n_per_rank
