"""Synthetic BAM record batches (structure of arrays) for tests: SURVEY.md §8d's mix of
CIGARs and flags, coordinate-sorted, without going through a BAM file."""
import numpy as np

from highperformancengs_amd.bamio import BamSoA, parse_cigar

CIGARS = ["150M", "40M2I108M", "60M5D90M", "10S140M", "75M200N75M", "5H100M45S", "50=50X50M", "1M", "30M1D1I118M"]
QLEN = [150, 150, 150, 150, 150, 145, 150, 1, 149]
FLAGS = np.array([0, 16, 0, 16, 0, 16, 0, 16, 99, 147, 4, 256, 512, 1024, 2048, 1 | 4], np.uint32)


def make_soa(n, refs, seed, sort=True, max_start_frac=1.0, cigars=None):
    rng = np.random.default_rng(seed)
    lens = np.array([l for _, l in refs], np.int64)
    tid = rng.choice(len(refs), size=n, p=lens / lens.sum()).astype(np.int32)
    pos = (rng.random(n) * (lens[tid] * max_start_frac)).astype(np.int32)
    if sort:
        order = np.lexsort((pos, tid))
        tid, pos = tid[order], pos[order]
    flag = FLAGS[rng.integers(0, len(FLAGS), n)]
    cg_sets = [np.array(parse_cigar(c), np.uint32) for c in (cigars or CIGARS)]
    qlens = QLEN if cigars is None else [sum(w >> 4 for w in cs if (w & 0xf) in (0, 1, 4, 7, 8)) for cs in cg_sets]
    pick = rng.integers(0, len(cg_sets), n)
    ncig = np.array([len(c) for c in cg_sets])[pick]
    cigar_off = np.zeros(n + 1, np.uint32)
    np.cumsum(ncig, out=cigar_off[1:])
    cigar = np.concatenate([cg_sets[k] for k in pick]) if n else np.zeros(0, np.uint32)
    l_qseq = np.array(qlens, np.int32)[pick]
    nb = (l_qseq + 1) // 2
    seq_off = np.zeros(n + 1, np.uint64)
    np.cumsum(nb, out=seq_off[1:])
    codes = np.array([1, 2, 4, 8, 15, 2, 4, 1, 8, 3, 5], np.uint8)  # A C G T N ... and a few ambiguity codes
    tot = int(seq_off[-1])
    seq4 = (codes[rng.integers(0, len(codes), tot)] << 4) | codes[rng.integers(0, len(codes), tot)]
    # zero the padding nibble of odd-length reads half of the time, garbage otherwise
    return BamSoA(refs=list(refs), tid=tid, pos=pos, flag=flag.astype(np.uint32), l_qseq=l_qseq,
                  cigar_off=cigar_off, cigar=cigar.astype(np.uint32), seq_off=seq_off,
                  seq4=np.ascontiguousarray(seq4, np.uint8))


def fmt_bedgraph(name, runs):
    return b"".join(b"%s\t%d\t%d\t%d\n" % (name.encode(), s, e, d) for s, e, d in runs.tolist())


def fmt_depth(name, tlen, W, win_sum):
    """output_bins (bam2depth.c:238-246): bins[k]/W with %.2f; bins are exact integers in double."""
    nm = name.encode()
    return b"".join(b"%s\t%d\t%d\t%s\n" % (nm, W * k, min(W * (k + 1), tlen), ("%.2f" % (float(int(win_sum[k])) / W)).encode())
                    for k in range(tlen // W + 1))
