"""Minimal BAM + BAI writer / reader used to make synthetic and fixture inputs.

This is input tooling (the reference repo has none: it relies on an external
``samtools view -bS`` + ``samtools index``).  The format written
is standard BGZF BAM (SAM spec §4) with a ``.bai`` that samtools-0.1.19's
``bam_index_load`` / ``bam_fetch`` accept (``samtools-0.1.19 -> bam_index.c``),
which is what the reference ``bam2depth`` needs (``bam2depth.c:112-119``).

The product's own BAM decoder is the C++ one in ``csrc/host/bam_reader.cpp``;
``read_bam_records`` below is an independent pure-Python decoder that tests use
to cross-check it.
"""
from __future__ import annotations

import gzip
import struct
import zlib
from dataclasses import dataclass, field
from typing import Iterable, List, Sequence, Tuple

import numpy as np

_CIGAR_OPS = "MIDNSHP=X"
_SEQ_CODES = "=ACMGRSVTWYHKDBN"  # bam_import.c:62 nibble table
_SEQ_LUT = {c: i for i, c in enumerate(_SEQ_CODES)}
_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
_MAX_BLOCK = 0xFF00  # payload bytes per BGZF block (samtools uses 0xff00)


def reg2bin(beg: int, end: int) -> int:
    """UCSC binning scheme (bam.h: bam_reg2bin)."""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def parse_cigar(text: str) -> List[int]:
    """'5M5D5M' -> packed uint32 ops (len<<4 | op)."""
    if text == "*":
        return []
    out, num = [], ""
    for ch in text:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | _CIGAR_OPS.index(ch))
            num = ""
    return out


def cigar_ref_len(cigar: Sequence[int]) -> int:
    """Reference span = M, D, N, =, X lengths (bam_calend, bam.c:17)."""
    n = 0
    for c in cigar:
        if (c & 0xF) in (0, 2, 3, 7, 8):
            n += c >> 4
    return n


def pack_seq(seq: str) -> bytes:
    if seq == "*":
        return b""
    codes = [_SEQ_LUT.get(c.upper(), 15) for c in seq]
    if len(codes) & 1:
        codes.append(0)
    return bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))


@dataclass
class BamRecord:
    tid: int
    pos: int  # 0-based
    flag: int
    cigar: List[int]
    seq: str = "*"
    qual: bytes = b""
    name: str = "r"
    mapq: int = 30
    mtid: int = -1
    mpos: int = -1
    isize: int = 0

    def encode(self) -> bytes:
        l_seq = 0 if self.seq == "*" else len(self.seq)
        rl = cigar_ref_len(self.cigar)
        end = self.pos + (rl if rl > 0 else 1)
        b = reg2bin(self.pos, end) if self.pos >= 0 else 4680
        name = self.name.encode() + b"\0"
        qual = self.qual if self.qual else b"\xff" * l_seq
        body = struct.pack(
            "<iiBBHHHiiii",
            self.tid, self.pos, len(name), self.mapq, b, len(self.cigar), self.flag,
            l_seq, self.mtid, self.mpos, self.isize,
        )
        body += name + struct.pack("<%dI" % len(self.cigar), *self.cigar)
        body += pack_seq(self.seq) + qual
        return struct.pack("<i", len(body)) + body


def records_from_sam(text: str) -> Tuple[List[Tuple[str, int]], List[BamRecord], str]:
    """Parse the small SAM subset used by the fixtures (@SQ lines + 11 columns)."""
    refs: List[Tuple[str, int]] = []
    recs: List[BamRecord] = []
    header = []
    for line in text.splitlines():
        if not line:
            continue
        if line.startswith("@"):
            header.append(line)
            if line.startswith("@SQ"):
                f = dict(x.split(":", 1) for x in line.split("\t")[1:])
                refs.append((f["SN"], int(f["LN"])))
            continue
        c = line.split("\t")
        names = [r[0] for r in refs]
        tid = names.index(c[2]) if c[2] != "*" else -1
        qual = b"" if c[10] == "*" else bytes(ord(ch) - 33 for ch in c[10])
        recs.append(BamRecord(tid=tid, pos=int(c[3]) - 1, flag=int(c[1]), cigar=parse_cigar(c[5]),
                              seq=c[9], qual=qual, name=c[0], mapq=int(c[4])))
    return refs, recs, "\n".join(header) + ("\n" if header else "")


class _Bgzf:
    """BGZF writer that never splits one write() across blocks."""

    def __init__(self, fh, level: int = 1):
        self.fh = fh
        self.level = level
        self.buf = bytearray()
        self.block_addr = 0  # compressed offset of the block being filled

    def tell(self) -> int:
        return (self.block_addr << 16) | len(self.buf)

    def _flush(self):
        if not self.buf:
            return
        co = zlib.compressobj(self.level, zlib.DEFLATED, -15)
        comp = co.compress(bytes(self.buf)) + co.flush()
        bsize = len(comp) + 25
        hdr = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, bsize)
        tail = struct.pack("<II", zlib.crc32(bytes(self.buf)) & 0xFFFFFFFF, len(self.buf))
        self.fh.write(hdr + comp + tail)
        self.block_addr += bsize + 1
        self.buf = bytearray()

    def write(self, data: bytes) -> int:
        """Append `data` whole; returns its virtual offset.  The virtual offset
        just past it is the value returned for the NEXT write (or end_offset()):
        that is what bgzf_tell reports after reading it, because a fully
        consumed block reports the next block's address (bgzf.c:342)."""
        if len(self.buf) + len(data) > _MAX_BLOCK:
            self._flush()
        beg = self.tell()
        if len(data) > _MAX_BLOCK:  # oversize item: own run of blocks
            for i in range(0, len(data), _MAX_BLOCK):
                self.buf += data[i:i + _MAX_BLOCK]
                self._flush()
            return beg
        self.buf += data
        return beg

    def end_offset(self) -> int:
        self._flush()
        return self.tell()

    def close(self):
        self._flush()
        self.fh.write(_BGZF_EOF)


def write_bam(path: str, refs: Sequence[Tuple[str, int]], records: Iterable[BamRecord],
              header_text: str | None = None, index: bool = True, level: int = 1) -> int:
    """Write coordinate-sorted `records` to `path` (+ `path`.bai). Returns #records."""
    if header_text is None:
        header_text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join(
            "@SQ\tSN:%s\tLN:%d\n" % (n, l) for n, l in refs)
    n_ref = len(refs)
    bins = [dict() for _ in range(n_ref)]  # bin -> list of [beg, end]
    lidx = [dict() for _ in range(n_ref)]  # 16 kb window -> min voffset
    n = 0
    with open(path, "wb") as fh:
        z = _Bgzf(fh, level)
        ht = header_text.encode()
        hdr = b"BAM\1" + struct.pack("<i", len(ht)) + ht + struct.pack("<i", n_ref)
        for name, ln in refs:
            nb = name.encode() + b"\0"
            hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln)
        z.write(hdr)
        z._flush()
        pending = None  # (record, beg) waiting for its end offset

        def commit(rec, beg, end):
            rl = cigar_ref_len(rec.cigar)
            rend = rec.pos + (rl if rl > 0 else 1)
            b = reg2bin(rec.pos, rend)
            ch = bins[rec.tid].setdefault(b, [])
            if ch and ch[-1][1] == beg:
                ch[-1][1] = end
            else:
                ch.append([beg, end])
            for w in range(rec.pos >> 14, ((rend - 1) >> 14) + 1):
                if w not in lidx[rec.tid] or beg < lidx[rec.tid][w]:
                    lidx[rec.tid][w] = beg

        for r in records:
            beg = z.write(r.encode())
            n += 1
            if pending is not None:
                commit(pending[0], pending[1], beg)
                pending = None
            if index and r.tid >= 0:
                pending = (r, beg)
        end = z.end_offset()
        if pending is not None:
            commit(pending[0], pending[1], end)
        z.close()
    if index:
        _write_bai(path + ".bai", n_ref, bins, lidx)
    return n


def _write_bai(path: str, n_ref: int, bins, lidx) -> None:
    """bins[t]: bin -> [[beg, end], ...] chunks of virtual offsets; lidx[t]: 16 kb window -> smallest voffset (bam_index.c)."""
    with open(path, "wb") as fh:
        fh.write(b"BAI\1" + struct.pack("<i", n_ref))
        for t in range(n_ref):
            fh.write(struct.pack("<i", len(bins[t])))
            for b in sorted(bins[t]):
                ch = bins[t][b]
                fh.write(struct.pack("<Ii", b, len(ch)))
                for beg, end in ch:
                    fh.write(struct.pack("<QQ", beg, end))
            n_intv = (max(lidx[t]) + 1) if lidx[t] else 0
            fh.write(struct.pack("<i", n_intv))
            last = 0
            for w in range(n_intv):  # fill_missing: carry the previous offset forward
                v = lidx[t].get(w, 0)
                if v == 0:
                    v = last
                last = v
                fh.write(struct.pack("<Q", v))


def repack_bam(src: str, dst: str, block: int = 20000, level: int = 6, index: bool = True) -> int:
    """The records of `src`, byte for byte, written the way htsjdk (Picard, GATK) writes a BAM: the uncompressed stream --
    header and records -- cut into BGZF blocks of `block` bytes wherever that falls, so records run across block ends
    (samtools never lets them, bam.c:238 bgzf_flush_try).  + `dst`.bai with the virtual offsets of the new blocks.
    Returns the number of records.  Input tooling for the tests of the straddling-record decode (kernels/bam_raw.hip)."""
    assert 0 < block <= 65536
    with gzip.open(src, "rb") as fh:
        data = fh.read()
    addr = []                                            # compressed offset of every new block
    with open(dst, "wb") as fh:
        at = 0
        for i in range(0, len(data), block):
            piece = data[i:i + block]
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = co.compress(piece) + co.flush()
            bsize = len(comp) + 26
            addr.append(at)
            fh.write(struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 66, 67, 2, bsize - 1) + comp +
                     struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
            at += bsize
        addr.append(at)
        fh.write(_BGZF_EOF)

    def voff(u: int) -> int:                             # what bgzf_tell reports with u bytes of the stream consumed (bgzf.c:342)
        return (addr[u // block] << 16) | (u % block)

    (l_text,) = struct.unpack_from("<i", data, 4)
    o = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, o)
    o += 4
    for _ in range(n_ref):
        o += 8 + struct.unpack_from("<i", data, o)[0]
    bins = [dict() for _ in range(n_ref)]
    lidx = [dict() for _ in range(n_ref)]
    n = 0
    while o < len(data):
        (bs,) = struct.unpack_from("<i", data, o)
        tid, pos, l_name, _mq, _bin, n_cig = struct.unpack_from("<iiBBHH", data, o + 4)
        if tid >= 0:
            cig = struct.unpack_from("<%dI" % n_cig, data, o + 36 + l_name)
            rl = cigar_ref_len(cig)
            rend = pos + (rl if rl > 0 else 1)
            beg, end = voff(o), voff(o + 4 + bs)
            ch = bins[tid].setdefault(reg2bin(pos, rend), [])
            if ch and ch[-1][1] == beg:
                ch[-1][1] = end
            else:
                ch.append([beg, end])
            for w in range(pos >> 14, ((rend - 1) >> 14) + 1):
                if w not in lidx[tid] or beg < lidx[tid][w]:
                    lidx[tid][w] = beg
        o += 4 + bs
        n += 1
    if index:
        _write_bai(dst + ".bai", n_ref, bins, lidx)
    return n


@dataclass
class BamSoA:
    """Decoded records as structure-of-arrays (mirror of hpn_bam_batch, include/hpngs.h)."""
    refs: List[Tuple[str, int]]
    tid: np.ndarray
    pos: np.ndarray
    flag: np.ndarray
    l_qseq: np.ndarray
    cigar_off: np.ndarray
    cigar: np.ndarray
    seq_off: np.ndarray
    seq4: np.ndarray
    names: List[str] = field(default_factory=list)


def read_bam_records(path: str) -> BamSoA:
    """Pure-Python sequential BAM decode (BGZF is multi-member gzip)."""
    with gzip.open(path, "rb") as fh:
        data = fh.read()
    assert data[:4] == b"BAM\1"
    (l_text,) = struct.unpack_from("<i", data, 4)
    o = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, o)
        name = data[o + 4:o + 4 + l_name - 1].decode()
        (l_ref,) = struct.unpack_from("<i", data, o + 4 + l_name)
        refs.append((name, l_ref))
        o += 8 + l_name
    tid, pos, flag, lq, coff, cig, soff, seq, names = [], [], [], [], [0], [], [0], bytearray(), []
    while o < len(data):
        (bs,) = struct.unpack_from("<i", data, o)
        t, p, l_name, _mq, _bin, n_cig, fl, l_seq = struct.unpack_from("<iiBBHHHi", data, o + 4)
        q = o + 36
        names.append(data[q:q + l_name - 1].decode())
        q += l_name
        cig.extend(struct.unpack_from("<%dI" % n_cig, data, q))
        q += 4 * n_cig
        seq += data[q:q + (l_seq + 1) // 2]
        tid.append(t), pos.append(p), flag.append(fl), lq.append(l_seq)
        coff.append(len(cig)), soff.append(len(seq))
        o += 4 + bs
    return BamSoA(refs=refs, tid=np.array(tid, np.int32), pos=np.array(pos, np.int32),
                  flag=np.array(flag, np.uint32), l_qseq=np.array(lq, np.int32),
                  cigar_off=np.array(coff, np.uint32), cigar=np.array(cig, np.uint32),
                  seq_off=np.array(soff, np.uint64), seq4=np.frombuffer(bytes(seq), np.uint8).copy(),
                  names=names)
