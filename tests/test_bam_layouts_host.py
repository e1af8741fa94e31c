"""CPU: the input tooling of the packed-BAM tests and the algorithm of the device's record index.

* bamio.repack_bam writes htsjdk's layout (records packed across BGZF blocks) with a .bai whose virtual offsets are those of the
  NEW blocks: every chunk begin / linear-index entry must be the start of a record of the right target when followed through
  the block table -- the GPU tests of the per-target workers (.bai offsets into the middle of a block) stand on it.
* scripts/emu_raw_chain.py restates what kernels/bam_raw.hip does (guess per block, walk, prove in file order, carry the record a
  call ends in): over every cut of the packed golden file the two calls together must index every record."""
import gzip
import os
import struct
import sys

import pytest

from conftest import golden_path
from highperformancengs_amd import bamio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def _block_table(path):
    raw, o, table, acc = open(path, "rb").read(), 0, {}, 0
    while o < len(raw):
        bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
        isz = struct.unpack_from("<I", raw, o + bsize - 4)[0]
        table[o] = acc                      # compressed offset of the block -> offset of its first byte in the inflated stream
        acc += isz
        o += bsize
    return table


@pytest.mark.parametrize("block", [777, 20000, 65280])
def test_repacked_bam_has_the_same_records_and_an_index_of_its_own_blocks(tmp_path, block):
    src = golden_path("bam", "rand.bam")
    dst = str(tmp_path / "p.bam")
    n = bamio.repack_bam(src, dst, block)
    a, b = bamio.read_bam_records(src), bamio.read_bam_records(dst)
    assert n == len(a.tid) == len(b.tid)
    for f in ("tid", "pos", "flag", "l_qseq", "cigar", "seq4"):
        assert (getattr(a, f) == getattr(b, f)).all(), f
    assert gzip.open(src).read() == gzip.open(dst).read()          # the same uncompressed stream, byte for byte
    data, table = gzip.open(dst).read(), _block_table(dst)
    starts = {}
    o = bamio_header_len(data)
    while o < len(data):
        starts[o] = struct.unpack_from("<i", data, o + 4)[0]       # record start -> refID
        o += 4 + struct.unpack_from("<i", data, o)[0]
    bai = open(dst + ".bai", "rb").read()
    assert bai[:4] == b"BAI\1"
    n_ref = struct.unpack_from("<i", bai, 4)[0]
    p, seen = 8, 0
    for t in range(n_ref):
        n_bin = struct.unpack_from("<i", bai, p)[0]
        p += 4
        for _ in range(n_bin):
            _bin, n_chunk = struct.unpack_from("<Ii", bai, p)
            p += 8
            for _ in range(n_chunk):
                beg, end = struct.unpack_from("<QQ", bai, p)
                p += 16
                at = table[beg >> 16] + (beg & 0xffff)
                assert starts.get(at) == t, (t, beg)                # a chunk begins at a record of its target
                assert end > beg
                seen += 1
        n_intv = struct.unpack_from("<i", bai, p)[0]
        p += 4
        for _ in range(n_intv):
            v = struct.unpack_from("<Q", bai, p)[0]
            p += 8
            if v:
                assert starts.get(table[v >> 16] + (v & 0xffff)) == t
    assert p == len(bai) and seen > 0


def bamio_header_len(data):
    l_text = struct.unpack_from("<i", data, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]
    p += 4
    for _ in range(n_ref):
        p += 8 + struct.unpack_from("<i", data, p)[0]
    return p


@pytest.mark.parametrize("block,lo,hi", [(97, 396, 404), (333, 1, 9), (1000, 30, 40), (31, 500, 504)])
def test_guess_walk_prove_and_carry_over_cuts_of_a_packed_file(block, lo, hi):
    import emu_raw_chain as E
    data = gzip.open(golden_path("bam", "rand.bam")).read()
    hl = bamio_header_len(data)
    rec = [hl]
    while rec[-1] < len(data):
        rec.append(rec[-1] + 4 + struct.unpack_from("<i", data, rec[-1])[0])
    first, nblk = hl // block, -(-len(data) // block)
    for cut in range(lo, hi):
        mid = first + cut
        a0, a1 = first * block, min(mid * block, len(data))
        blocks = [(i * block - a0, min(block, len(data) - i * block)) for i in range(first, mid)]
        f, n, tail, offs = E.index(data[a0:a1], blocks, hl - a0)
        n_a = sum(1 for k in range(len(rec) - 1) if rec[k + 1] <= a1)
        assert f & 3 == 0 and n == n_a and tail == a1 - rec[n_a] and offs == [r - a0 for r in rec[:n_a]], (block, cut)
        front = data[a1 - tail:a1]
        blocks2 = [(tail + i * block - a1, min(block, len(data) - i * block)) for i in range(mid, nblk)]
        f2, n2, t2, _ = E.index(front + data[a1:], blocks2, 0)
        assert f2 & 3 == 0 and t2 == 0 and n2 == len(rec) - 1 - n_a, (block, cut)
