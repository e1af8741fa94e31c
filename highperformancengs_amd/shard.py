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
