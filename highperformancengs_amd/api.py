"""Thin Python view of the C ABI, used by tests and bench.py.

The drop-in surface of this project is the C ABI (include/hpngs.h) and the CLI
tools built on it (csrc/tools); this module only moves pointers.  Host arrays are
numpy, device arrays are anything with ``data_ptr()`` (torch tensors) or an int
address.  Every failure raises ``HpnError``; nothing here computes on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import HpnError, Tally


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    raise TypeError(type(x))


class TallyResult:
    """Accumulators of count_read (fastq_count_kthread.c:116): numpy views over an hpn_tally."""

    def __init__(self, qual_hist=False, nuc_hist=False):
        self.c = Tally()
        self.qual_hist = np.zeros((_lib.QUAL_ROWS, _lib.LEN_BINS), np.uint64) if qual_hist else None
        self.nuc_hist = np.zeros((_lib.NUC_CODES, _lib.LEN_BINS), np.uint64) if nuc_hist else None
        if qual_hist:
            self.c.qual_hist = self.qual_hist.ctypes.data_as(C.POINTER(C.c_uint64))
        if nuc_hist:
            self.c.nuc_hist = self.nuc_hist.ctypes.data_as(C.POINTER(C.c_uint64))

    @property
    def seqlen(self):
        return np.frombuffer(self.c.seqlen, dtype=np.uint64)

    @property
    def total(self):
        return int(self.c.total)

    @property
    def q20(self):
        return int(self.c.q20)

    @property
    def q30(self):
        return int(self.c.q30)


class Context:
    """One per GPU (hpn_ctx)."""

    def __init__(self, device=0):
        self.L = _lib.lib()
        h = C.c_void_p()
        rc = self.L.hpn_ctx_create(device, C.byref(h))
        if rc != 0:
            raise HpnError(rc, "hpn_ctx_create")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.hpn_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise HpnError(rc, what, self.L.hpn_ctx_last_error(self.h).decode())

    def set_stream(self, hip_stream):
        self._ck(self.L.hpn_ctx_set_stream(self.h, C.c_void_p(hip_stream) if hip_stream else None), "set_stream")

    def sync(self):
        self._ck(self.L.hpn_ctx_sync(self.h), "sync")

    def last_kernel_ms(self, family=0):
        ms = C.c_float()
        self._ck(self.L.hpn_ctx_last_kernel_ms(self.h, family, C.byref(ms)), "last_kernel_ms")
        return ms.value

    # ---- fastq_count ------------------------------------------------------
    def fastq_tally(self, qual, off, base=None, acc=None, qual_hist=False, nuc_hist=False):
        """Host batch -> adds into `acc` (new TallyResult if None). Mirrors count_read."""
        qual = np.ascontiguousarray(qual, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        if base is not None:
            base = np.ascontiguousarray(base, np.uint8)
        if acc is None:
            acc = TallyResult(qual_hist, nuc_hist)
        self._ck(self.L.hpn_fastq_tally(self.h, _ptr(qual), _ptr(base), _ptr(off), len(off) - 1, C.byref(acc.c)),
                 "hpn_fastq_tally")
        return acc

    def fastq_tally_dev(self, d_qual, d_off, n, d_base=None, flags=0):
        self._ck(self.L.hpn_fastq_tally_dev(self.h, _ptr(d_qual), _ptr(d_base), _ptr(d_off), n, flags),
                 "hpn_fastq_tally_dev")

    def fastq_tally_fetch(self, acc=None, qual_hist=False, nuc_hist=False):
        if acc is None:
            acc = TallyResult(qual_hist, nuc_hist)
        self._ck(self.L.hpn_fastq_tally_fetch(self.h, C.byref(acc.c)), "hpn_fastq_tally_fetch")
        return acc

    def tally_devptr(self):
        p = C.c_void_p()
        self._ck(self.L.hpn_fastq_tally_devptr(self.h, C.byref(p)), "hpn_fastq_tally_devptr")
        return p.value

    # ---- R plugin tally (Rgzfastq_uniq.c) -----------------------------------
    def fastq_rqc(self, seq, qual, off, out=None):
        """-> dict(quality int32[300,128] view of [q+128*pos], nucleotide int32[300,5], length int32[300], gc f64[n]);
        pass the previous dict as `out` to keep adding into the matrices."""
        seq = np.ascontiguousarray(seq, np.uint8)
        qual = np.ascontiguousarray(qual, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        M = _lib.RQC_MAXLEN
        out = out or {"quality": np.zeros((M, 128), np.int32), "nucleotide": np.zeros((M, 5), np.int32),
                      "length": np.zeros(M, np.int32)}
        out["gc"] = np.zeros(max(n, 1), np.float64)
        r = _lib.Rqc()
        i32 = C.POINTER(C.c_int32)
        r.quality, r.nucleotide = out["quality"].ctypes.data_as(i32), out["nucleotide"].ctypes.data_as(i32)
        r.length, r.gc = out["length"].ctypes.data_as(i32), out["gc"].ctypes.data_as(C.POINTER(C.c_double))
        self._ck(self.L.hpn_fastq_rqc(self.h, _ptr(seq), _ptr(qual), _ptr(off), n, C.byref(r)), "hpn_fastq_rqc")
        out["gc"] = out["gc"][:n]
        return out

    def fastq_read_gc_dev(self, d_seq, d_off, n, d_gc):
        self._ck(self.L.hpn_fastq_read_gc_dev(self.h, _ptr(d_seq), _ptr(d_off), n, _ptr(d_gc)), "hpn_fastq_read_gc_dev")

    # ---- fastq_trim -------------------------------------------------------
    def fastq_trim(self, seq, qual, off, S, E):
        seq = np.ascontiguousarray(seq, np.uint8)
        qual = np.ascontiguousarray(qual, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        cap = max(int(off[-1] - off[0]), 1)
        oseq, oqual, ooff = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8), np.zeros(n + 1, np.uint64)
        self._ck(self.L.hpn_fastq_trim(self.h, _ptr(seq), _ptr(qual), _ptr(off), n, S, E, _ptr(oseq), _ptr(oqual),
                                       _ptr(ooff)), "hpn_fastq_trim")
        tot = int(ooff[-1])
        return oseq[:tot], oqual[:tot], ooff

    def fastq_trim_dev(self, d_seq, d_qual, d_off, n, S, E, d_out_seq, d_out_qual, d_out_off):
        self._ck(self.L.hpn_fastq_trim_dev(self.h, _ptr(d_seq), _ptr(d_qual), _ptr(d_off), n, S, E, _ptr(d_out_seq),
                                           _ptr(d_out_qual), _ptr(d_out_off)), "hpn_fastq_trim_dev")

    # ---- extension: quality-threshold trim points --------------------------
    def fastq_qtrim_points(self, qual, off, threshold):
        qual = np.ascontiguousarray(qual, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        n = len(off) - 1
        beg, end = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
        self._ck(self.L.hpn_fastq_qtrim_points(self.h, _ptr(qual), _ptr(off), n, threshold, _ptr(beg), _ptr(end)),
                 "hpn_fastq_qtrim_points")
        return beg[:n], end[:n]

    def fastq_trim_points(self, seq, qual, off, beg, end):
        seq = np.ascontiguousarray(seq, np.uint8)
        qual = np.ascontiguousarray(qual, np.uint8)
        off = np.ascontiguousarray(off, np.uint64)
        beg, end = np.ascontiguousarray(beg, np.uint32), np.ascontiguousarray(end, np.uint32)
        n = len(off) - 1
        cap = max(int(off[-1] - off[0]), 1)
        oseq, oqual, ooff = np.zeros(cap, np.uint8), np.zeros(cap, np.uint8), np.zeros(n + 1, np.uint64)
        self._ck(self.L.hpn_fastq_trim_points(self.h, _ptr(seq), _ptr(qual), _ptr(off), n, _ptr(beg), _ptr(end),
                                              _ptr(oseq), _ptr(oqual), _ptr(ooff)), "hpn_fastq_trim_points")
        tot = int(ooff[-1])
        return oseq[:tot], oqual[:tot], ooff

    # ---- raw-text front end (device-side framing) ------------------------------
    def text_begin(self):
        self._ck(self.L.hpn_fastq_text_begin(self.h), "hpn_fastq_text_begin")

    @staticmethod
    def _text(chunk):
        if isinstance(chunk, (bytes, bytearray, memoryview)):
            chunk = np.frombuffer(chunk, np.uint8)
        if isinstance(chunk, np.ndarray):
            chunk = np.ascontiguousarray(chunk, np.uint8)
            return chunk, chunk.size
        return chunk, chunk.numel()  # device tensor

    def text_count(self, chunk, last=False, flags=0):
        """One chunk of FASTQ text -> device accumulators; returns the hpn_text_info."""
        chunk, n = self._text(chunk)
        info = _lib.TextInfo()
        self._ck(self.L.hpn_fastq_text_count(self.h, _ptr(chunk) if n else None, n, int(bool(last)), flags, C.byref(info)),
                 "hpn_fastq_text_count")
        return info

    def text_count_inplace(self, d_text, nbytes, last=False, flags=0):
        """A chunk of FASTQ text that lies on the device, framed where it lies (8192 writable bytes in front of d_text)."""
        info = _lib.TextInfo()
        self._ck(self.L.hpn_fastq_text_count_inplace(self.h, _ptr(d_text) if d_text is not None else None, int(nbytes), int(bool(last)), flags,
                                                     C.byref(info)), "hpn_fastq_text_count_inplace")
        return info

    def text_records(self, chunk, last=False):
        """One chunk of FASTQ text -> hpn_text_info with the records it completes (gzfastq_sample's count_read)."""
        chunk, n = self._text(chunk)
        info = _lib.TextInfo()
        self._ck(self.L.hpn_fastq_text_records(self.h, _ptr(chunk) if n else None, n, int(bool(last)), C.byref(info)),
                 "hpn_fastq_text_records")
        return info

    def text_trim(self, chunk, S, E, last=False):
        """One chunk of FASTQ text -> (trimmed text bytes, hpn_text_info)."""
        chunk, n = self._text(chunk)
        info = _lib.TextInfo()
        out = np.zeros(n + 8192, np.uint8)
        self._ck(self.L.hpn_fastq_text_trim(self.h, _ptr(chunk) if n else None, n, int(bool(last)), S, E, _ptr(out), out.size,
                                            C.byref(info)), "hpn_fastq_text_trim")
        return out[:0 if info.irregular else int(info.n_bytes)].tobytes(), info

    # ---- one stream framed by several contexts: pieces ---------------------------------
    def text_piece_lines(self, text, head, own_bytes, last=False):
        """First half of a piece (hpn_fastq_text_piece_lines): text = head byte + piece + tail; returns hpn_text_piece."""
        text, n = self._text(text)
        out = _lib.TextPiece()
        self._ck(self.L.hpn_fastq_text_piece_lines(self.h, _ptr(text) if n else None, n, head, own_bytes, int(bool(last)), C.byref(out)),
                 "hpn_fastq_text_piece_lines")
        return out

    def text_piece_count(self, lines_before, flags=0):
        info = _lib.TextInfo()
        self._ck(self.L.hpn_fastq_text_piece_count(self.h, lines_before, flags, C.byref(info)), "hpn_fastq_text_piece_count")
        return info

    def text_piece_trim(self, lines_before, S, E, cap):
        info = _lib.TextInfo()
        out = np.zeros(cap, np.uint8)
        self._ck(self.L.hpn_fastq_text_piece_trim(self.h, lines_before, S, E, _ptr(out), out.size, C.byref(info)), "hpn_fastq_text_piece_trim")
        return out[:0 if info.irregular else int(info.n_bytes)].tobytes(), info

    # ---- BGZF inflate on the device ----------------------------------------------
    def bgzf_inflate_dev(self, d_comp, d_blocks, n_blocks, d_out, d_status):
        self._ck(self.L.hpn_bgzf_inflate_dev(self.h, _ptr(d_comp), _ptr(d_blocks), n_blocks, _ptr(d_out), _ptr(d_status)),
                 "hpn_bgzf_inflate_dev")

    # ---- one gzip member on the device (two passes) -------------------------------
    def gz_inflate_dev(self, d_comp, d_chunks, n_chunks, sym_cap, d_text, text_cap, d_window_in=None, d_window_out=None):
        """hpn_gz_inflate_dev; returns hpn_gz_info (status != 0: a stretch could not be decoded as given)."""
        info = _lib.GzInfo()
        self._ck(self.L.hpn_gz_inflate_dev(self.h, _ptr(d_comp), _ptr(d_chunks), n_chunks, sym_cap,
                                           _ptr(d_window_in) if d_window_in is not None else None, _ptr(d_text) if d_text is not None else None,
                                           text_cap, _ptr(d_window_out) if d_window_out is not None else None, C.byref(info)),
                 "hpn_gz_inflate_dev")
        return info

    def gz_members(self):
        """hpn_gz_members: [(text_end, isize)] of the members that ended inside the last hpn_gz_inflate_dev call."""
        n = C.c_uint32(0)
        rc = self.L.hpn_gz_members(self.h, None, 0, C.byref(n))
        if n.value == 0:
            self._ck(rc, "hpn_gz_members")
            return []
        buf = np.zeros(n.value, np.dtype([("text_end", "<u8"), ("isize", "<u4"), ("reserved", "<u4")]))
        self._ck(self.L.hpn_gz_members(self.h, buf.ctypes.data_as(C.c_void_p), n.value, C.byref(n)), "hpn_gz_members")
        return [(int(r["text_end"]), int(r["isize"])) for r in buf]

    def crc32_dev(self, d_data, spans):
        """CRC-32 of d_data[off : off + len) for every (off, len) in spans (hpn_crc32_dev) -> list of ints."""
        n = len(spans)
        arr = np.array(spans, np.uint64).reshape(-1, 2) if n else np.zeros((0, 2), np.uint64)
        out = np.zeros(max(n, 1), np.uint32)
        self._ck(self.L.hpn_crc32_dev(self.h, _ptr(d_data), _ptr(arr) if n else None, n, _ptr(out)), "hpn_crc32_dev")
        return [int(x) for x in out[:n]]

    def gz_find_starts_dev(self, d_comp, comp_bytes, slices):
        """First block start (bit position in d_comp) in every (lo_bit, n_bits) slice, or 2^64 - 1 (hpn_gz_find_starts_dev)."""
        n = len(slices)
        arr = np.array(slices, np.uint64).reshape(-1, 2) if n else np.zeros((0, 2), np.uint64)
        out = np.zeros(max(n, 1), np.uint64)
        self._ck(self.L.hpn_gz_find_starts_dev(self.h, _ptr(d_comp), int(comp_bytes), _ptr(arr) if n else None, n, _ptr(out)), "hpn_gz_find_starts_dev")
        return out[:n]

    # ---- BAM --------------------------------------------------------------
    @staticmethod
    def _batch(soa, keep):
        """hpn_bam_batch over numpy arrays / device tensors; `keep` pins the temporaries."""
        b = _lib.BamBatch()
        b.n = len(soa.tid)
        for f in ("tid", "pos", "flag", "l_qseq", "cigar_off", "cigar", "seq_off", "seq4"):
            a = getattr(soa, f, None)
            if a is not None and isinstance(a, np.ndarray):
                a = np.ascontiguousarray(a)
                keep.append(a)
            setattr(b, f, _ptr(a).value if a is not None else None)
        return b

    def depth_target(self, soa, tid, target_len, W, flag_mask=0x704, dev=False, runs_cap=None):
        """One chromosome of bam2depth: (runs[n,3] int32, win_sum uint64[target_len//W+1])."""
        keep = []
        self._ck(self.L.hpn_depth_begin(self.h, tid, target_len, flag_mask), "hpn_depth_begin")
        b = self._batch(soa, keep)
        add = self.L.hpn_depth_add_dev if dev else self.L.hpn_depth_add
        self._ck(add(self.h, C.byref(b)), "hpn_depth_add")
        return self.depth_finish(target_len, W, runs_cap)

    def depth_finish(self, target_len, W, runs_cap=None):
        win = np.zeros(target_len // W + 1, np.uint64)
        cap = runs_cap if runs_cap is not None else 1 << 16
        while True:
            runs = np.zeros((cap, 3), np.int32)
            nr = C.c_uint64(0)
            rc = self.L.hpn_depth_finish(self.h, W, _ptr(runs), cap, C.byref(nr), _ptr(win))
            if rc == _lib.E_CAPACITY:
                cap = int(nr.value)
                continue
            self._ck(rc, "hpn_depth_finish")
            return runs[:nr.value], win

    def depth_bedgraph_format(self, name):
        """Format the bedGraph text of the last depth_finish on the device; returns its size (the text stays there)."""
        nb = C.c_uint64(0)
        self._ck(self.L.hpn_depth_bedgraph_format(self.h, name.encode(), C.byref(nb)), "hpn_depth_bedgraph_format")
        return nb.value

    def depth_bedgraph(self, name):
        """bedGraph text of the runs of the last depth_finish, formatted on the device (read back in two pieces on purpose)."""
        nb = C.c_uint64(0)
        self._ck(self.L.hpn_depth_bedgraph_format(self.h, name.encode(), C.byref(nb)), "hpn_depth_bedgraph_format")
        out = np.zeros(max(nb.value, 1), np.uint8)
        cut = nb.value // 3
        self._ck(self.L.hpn_depth_bedgraph_read(self.h, 0, _ptr(out), cut), "hpn_depth_bedgraph_read")
        self._ck(self.L.hpn_depth_bedgraph_read(self.h, cut, out[cut:].ctypes.data, nb.value - cut), "hpn_depth_bedgraph_read")
        return out[:nb.value].tobytes()

    def window_counts(self, soa, win_off, W, dev=False):
        keep = []
        win_off = np.ascontiguousarray(win_off, np.uint64)
        nt = len(win_off) - 1
        self._ck(self.L.hpn_window_begin(self.h, nt, _ptr(win_off), W), "hpn_window_begin")
        b = self._batch(soa, keep)
        add = self.L.hpn_window_add_dev if dev else self.L.hpn_window_add
        self._ck(add(self.h, C.byref(b)), "hpn_window_add")
        tot = int(win_off[-1])
        bins, gc, ln = np.zeros(tot, np.uint32), np.zeros(tot, np.uint64), np.zeros(tot, np.uint32)
        touched = np.zeros(nt, np.uint8)
        nc = C.c_uint64(0)
        self._ck(self.L.hpn_window_finish(self.h, _ptr(bins), _ptr(gc), _ptr(ln), _ptr(touched), C.byref(nc)),
                 "hpn_window_finish")
        return bins, gc, ln, touched, nc.value

    # ---- BAM records in place in inflated BGZF blocks ------------------------------
    def bam_raw_index_dev(self, d_raw, d_blocks, n_blocks, first_off, d_status):
        info = _lib.RawInfo()
        self._ck(self.L.hpn_bam_raw_index_dev(self.h, _ptr(d_raw), _ptr(d_blocks), n_blocks, first_off, _ptr(d_status),
                                              C.byref(info)), "hpn_bam_raw_index_dev")
        return info

    def depth_target_raw(self, d_raw, tid, target_len, W, flag_mask=0x704):
        self._ck(self.L.hpn_depth_begin(self.h, tid, target_len, flag_mask), "hpn_depth_begin")
        self._ck(self.L.hpn_depth_add_raw_dev(self.h, _ptr(d_raw)), "hpn_depth_add_raw_dev")
        return self.depth_finish(target_len, W)

    def window_counts_raw(self, d_raw, win_off, W):
        win_off = np.ascontiguousarray(win_off, np.uint64)
        nt = len(win_off) - 1
        self._ck(self.L.hpn_window_begin(self.h, nt, _ptr(win_off), W), "hpn_window_begin")
        self._ck(self.L.hpn_window_add_raw_dev(self.h, _ptr(d_raw)), "hpn_window_add_raw_dev")
        tot = int(win_off[-1])
        bins, gc, ln = np.zeros(tot, np.uint32), np.zeros(tot, np.uint64), np.zeros(tot, np.uint32)
        touched = np.zeros(nt, np.uint8)
        nc = C.c_uint64(0)
        self._ck(self.L.hpn_window_finish(self.h, _ptr(bins), _ptr(gc), _ptr(ln), _ptr(touched), C.byref(nc)),
                 "hpn_window_finish")
        return bins, gc, ln, touched, nc.value

    # ---- collectives / synthetic -----------------------------------------
    def comm_init(self, rank, n_ranks, unique_id: bytes):
        buf = (C.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._ck(self.L.hpn_comm_init(self.h, rank, n_ranks, buf), "hpn_comm_init")

    def allreduce_u64(self, d_vec, n):
        self._ck(self.L.hpn_allreduce_u64(self.h, _ptr(d_vec), n), "hpn_allreduce_u64")

    def comm_count(self) -> int:
        """Ranks of this context's communicator as RCCL reports them (ncclCommCount)."""
        n = C.c_int(0)
        self._ck(self.L.hpn_comm_count(self.h, C.byref(n)), "hpn_comm_count")
        return n.value

    def synth_fastq_dev(self, seed, first, n, length, d_qual, d_base, d_off):
        self._ck(self.L.hpn_synth_fastq_dev(self.h, seed, first, n, length, _ptr(d_qual), _ptr(d_base), _ptr(d_off)),
                 "hpn_synth_fastq_dev")


def comm_init_all(ctxs):
    """One communicator per context of THIS process (ncclCommInitAll): the contexts must sit on distinct devices."""
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    rc = _lib.lib().hpn_comm_init_all(arr, len(ctxs))
    if rc != 0:
        raise HpnError(rc, "hpn_comm_init_all", _lib.lib().hpn_ctx_last_error(ctxs[0].h).decode() if ctxs else "")


def allreduce_u64_all(ctxs, d_vecs, n_words):
    """Sum d_vecs[i] (on ctxs[i]'s device) in place into every one of them, in one RCCL group."""
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    vec = (C.c_void_p * len(ctxs))(*[v if isinstance(v, int) else _ptr(v) for v in d_vecs])
    rc = _lib.lib().hpn_allreduce_u64_all(arr, vec, len(ctxs), n_words)
    if rc != 0:
        raise HpnError(rc, "hpn_allreduce_u64_all", _lib.lib().hpn_ctx_last_error(ctxs[0].h).decode() if ctxs else "")


def comm_library() -> str:
    """Path of the RCCL library the binding resolved to ('' before the first use / when none could be loaded)."""
    return _lib.lib().hpn_comm_library().decode()


def comm_unique_id() -> bytes:
    buf = (C.c_uint8 * _lib.UNIQUE_ID_BYTES)()
    rc = _lib.lib().hpn_comm_unique_id(buf)
    if rc != 0:
        raise HpnError(rc, "hpn_comm_unique_id")
    return bytes(buf)


def device_count() -> int:
    n = C.c_int(0)
    _lib.lib().hpn_device_count(C.byref(n))
    return n.value
