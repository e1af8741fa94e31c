"""ctypes binding of libhpngs.so (the C ABI in include/hpngs.h).

There is no fallback: if the library is missing or a call fails, this raises.
Inside a PyTorch process import torch BEFORE this module so that both share the
one libamdhip64.so.7 / librccl.so.1 torch ships (same sonames as /opt/rocm).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HPN_LIB") or os.path.join(HERE, "libhpngs.so")   # HPN_LIB: A/B runs of a variant build

LEN_BINS, QUAL_ROWS, NUC_CODES = 512, 128, 5
W_SEQLEN, W_TOTAL, W_Q20, W_Q30, W_BAD, W_QUAL = 0, 512, 513, 514, 515, 516
W_NUC = W_QUAL + QUAL_ROWS * LEN_BINS
TALLY_WORDS = W_NUC + NUC_CODES * LEN_BINS
TALLY_QUAL_HIST, TALLY_NUC_HIST = 1, 2
DEPTH_ANY_ORDER = 0x80000000
UNIQUE_ID_BYTES = 128

OK, E_NODEVICE, E_HIP, E_ARG, E_DOMAIN, E_NOMEM, E_STATE, E_RCCL, E_CAPACITY, E_PARTIAL = 0, -1, -2, -3, -4, -5, -6, -7, -8, -9


class HpnError(RuntimeError):
    def __init__(self, status, what, detail=""):
        self.status = status
        super().__init__(f"{what}: status {status} ({_strerror(status)}) {detail}".strip())


class Tally(C.Structure):
    _fields_ = [("seqlen", C.c_uint64 * LEN_BINS), ("total", C.c_uint64), ("q20", C.c_uint64),
                ("q30", C.c_uint64), ("qual_hist", C.POINTER(C.c_uint64)), ("nuc_hist", C.POINTER(C.c_uint64))]


class TextInfo(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("n_bytes", C.c_uint64), ("carry_bytes", C.c_uint64),
                ("irregular", C.c_uint32), ("reserved", C.c_uint32)]


class TextPiece(C.Structure):
    _fields_ = [("n_lines", C.c_uint64), ("irregular", C.c_uint32), ("reserved", C.c_uint32)]


TEXT_PIECE_TAIL = 4096
TEXT_NUL, TEXT_LONG_LINE, TEXT_RAGGED, TEXT_PARTIAL, TEXT_LEN, TEXT_DENSE, TEXT_STALE = 1, 2, 4, 8, 16, 32, 64


class Rqc(C.Structure):
    _fields_ = [("quality", C.POINTER(C.c_int32)), ("nucleotide", C.POINTER(C.c_int32)), ("length", C.POINTER(C.c_int32)),
                ("gc", C.POINTER(C.c_double))]


RQC_MAXLEN = 300


class RawInfo(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("tid_min", C.c_int32), ("tid_max", C.c_int32), ("flags", C.c_uint32),
                ("tail_bytes", C.c_uint32)]


class GzInfo(C.Structure):
    _fields_ = [("n_bytes", C.c_uint64), ("end_bit", C.c_uint64), ("status", C.c_uint32), ("bad_chunk", C.c_uint32),
                ("final_chunk", C.c_uint32), ("reserved", C.c_uint32)]


class Run(C.Structure):
    _fields_ = [("start", C.c_int32), ("end", C.c_int32), ("depth", C.c_int32)]


class BamBatch(C.Structure):
    _fields_ = [("n", C.c_uint64), ("tid", C.c_void_p), ("pos", C.c_void_p), ("flag", C.c_void_p),
                ("l_qseq", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("seq_off", C.c_void_p), ("seq4", C.c_void_p)]


# every symbol include/hpngs.h declares: (name, restype, argtypes)
_vp, _u64, _u32, _i32, _int, _sz = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_int, C.c_size_t
SYMBOLS = [
    ("hpn_abi_version", _int, []),
    ("hpn_strerror", C.c_char_p, [_int]),
    ("hpn_device_count", _int, [C.POINTER(_int)]),
    ("hpn_ctx_create", _int, [_int, C.POINTER(_vp)]),
    ("hpn_ctx_destroy", _int, [_vp]),
    ("hpn_ctx_set_stream", _int, [_vp, _vp]),
    ("hpn_ctx_sync", _int, [_vp]),
    ("hpn_ctx_device", _int, [_vp, C.POINTER(C.c_int)]),
    ("hpn_ctx_pci_address", _int, [_vp, C.c_char_p, _int]),
    ("hpn_ctx_last_error", C.c_char_p, [_vp]),
    ("hpn_ctx_last_kernel_ms", _int, [_vp, _int, C.POINTER(C.c_float)]),
    ("hpn_dev_malloc", _int, [_vp, _sz, C.POINTER(_vp)]),
    ("hpn_dev_mem_info", _int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("hpn_dev_free", _int, [_vp, _vp]),
    ("hpn_host_malloc", _int, [_vp, _sz, C.POINTER(_vp)]),
    ("hpn_host_free", _int, [_vp, _vp]),
    ("hpn_memcpy_h2d", _int, [_vp, _vp, _vp, _sz]),
    ("hpn_memcpy_d2h", _int, [_vp, _vp, _vp, _sz]),
    ("hpn_memcpy_d2d", _int, [_vp, _vp, _vp, _sz]),
    ("hpn_fastq_tally", _int, [_vp, _vp, _vp, _vp, _u64, C.POINTER(Tally)]),
    ("hpn_fastq_tally_dev", _int, [_vp, _vp, _vp, _vp, _u64, _u32]),
    ("hpn_fastq_tally_fetch", _int, [_vp, C.POINTER(Tally)]),
    ("hpn_fastq_tally_devptr", _int, [_vp, C.POINTER(_vp)]),
    ("hpn_fastq_rqc", _int, [_vp, _vp, _vp, _vp, _u64, C.POINTER(Rqc)]),
    ("hpn_fastq_read_gc_dev", _int, [_vp, _vp, _vp, _u64, _vp]),
    ("hpn_fastq_trim", _int, [_vp, _vp, _vp, _vp, _u64, _i32, _i32, _vp, _vp, _vp]),
    ("hpn_fastq_trim_dev", _int, [_vp, _vp, _vp, _vp, _u64, _i32, _i32, _vp, _vp, _vp]),
    ("hpn_fastq_qtrim_points", _int, [_vp, _vp, _vp, _u64, _u32, _vp, _vp]),
    ("hpn_fastq_qtrim_points_dev", _int, [_vp, _vp, _vp, _u64, _u32, _vp, _vp]),
    ("hpn_fastq_trim_points", _int, [_vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    ("hpn_fastq_trim_points_dev", _int, [_vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    ("hpn_fastq_text_begin", _int, [_vp]),
    ("hpn_fastq_text_count", _int, [_vp, _vp, _u64, _int, _u32, C.POINTER(TextInfo)]),
    ("hpn_fastq_text_count_inplace", _int, [_vp, _vp, _u64, _int, _u32, C.POINTER(TextInfo)]),
    ("hpn_fastq_text_records", _int, [_vp, _vp, _u64, _int, C.POINTER(TextInfo)]),
    ("hpn_fastq_text_trim", _int, [_vp, _vp, _u64, _int, _i32, _i32, _vp, _u64, C.POINTER(TextInfo)]),
    ("hpn_fastq_text_piece_lines", _int, [_vp, _vp, _u64, _u32, _u64, _int, C.POINTER(TextPiece)]),
    ("hpn_fastq_text_piece_count", _int, [_vp, _u64, _u32, C.POINTER(TextInfo)]),
    ("hpn_fastq_text_piece_trim", _int, [_vp, _u64, _i32, _i32, _vp, _u64, C.POINTER(TextInfo)]),
    ("hpn_bgzf_inflate_dev", _int, [_vp, _vp, _vp, _u64, _vp, _vp]),
    ("hpn_gz_inflate_dev", _int, [_vp, _vp, _vp, _u32, _u32, _vp, _vp, _u64, _vp, C.POINTER(GzInfo)]),
    ("hpn_inflate_slots", _int, [_vp, C.POINTER(_u32)]),
    ("hpn_gz_inflate_begin_dev", _int, [_vp, _vp, _vp, _u32, _u32]),
    ("hpn_gz_inflate_finish_dev", _int, [_vp, _vp, _vp, _u64, _vp, C.POINTER(GzInfo)]),
    ("hpn_gz_members", _int, [_vp, _vp, _u32, C.POINTER(_u32)]),
    ("hpn_crc32_dev", _int, [_vp, _vp, _vp, _u32, _vp]),
    ("hpn_gz_find_starts_dev", _int, [_vp, _vp, C.c_uint64, _vp, _u32, _vp]),
    ("hpn_crc32_join", _u32, [_u32, _u32, _u64]),
    ("hpn_bam_raw_index_dev", _int, [_vp, _vp, _vp, _u64, _u32, _vp, C.POINTER(RawInfo)]),
    ("hpn_depth_add_raw_dev", _int, [_vp, _vp]),
    ("hpn_window_add_raw_dev", _int, [_vp, _vp]),
    ("hpn_depth_begin", _int, [_vp, _i32, _u32, _u32]),
    ("hpn_depth_begin_w", _int, [_vp, _i32, _u32, _u32, _u32]),
    ("hpn_depth_progress", _int, [_vp, C.POINTER(_u64)]),
    ("hpn_depth_add", _int, [_vp, C.POINTER(BamBatch)]),
    ("hpn_depth_add_dev", _int, [_vp, C.POINTER(BamBatch)]),
    ("hpn_depth_finish", _int, [_vp, _u32, _vp, _u64, C.POINTER(_u64), _vp]),
    ("hpn_depth_bedgraph_format", _int, [_vp, C.c_char_p, C.POINTER(_u64)]),
    ("hpn_depth_bedgraph_read", _int, [_vp, _u64, _vp, _u64]),
    ("hpn_depth_bedgraph_dev", _int, [_vp, C.POINTER(_vp), C.POINTER(_u64)]),
    ("hpn_window_begin", _int, [_vp, _i32, _vp, _u32]),
    ("hpn_window_add", _int, [_vp, C.POINTER(BamBatch)]),
    ("hpn_window_add_dev", _int, [_vp, C.POINTER(BamBatch)]),
    ("hpn_window_finish", _int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_u64)]),
    ("hpn_comm_unique_id", _int, [_vp]),
    ("hpn_comm_init", _int, [_vp, _int, _int, _vp]),
    ("hpn_comm_destroy", _int, [_vp]),
    ("hpn_allreduce_u64", _int, [_vp, _vp, _sz]),
    ("hpn_comm_init_all", _int, [C.POINTER(_vp), _int]),
    ("hpn_allreduce_u64_all", _int, [C.POINTER(_vp), C.POINTER(_vp), _int, _sz]),
    ("hpn_comm_library", C.c_char_p, []),
    ("hpn_comm_count", _int, [_vp, C.POINTER(_int)]),
    ("hpn_synth_fastq_dev", _int, [_vp, _u64, _u64, _u64, _u32, _vp, _vp, _vp]),
]

_lib = None


def lib():
    """Load libhpngs.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C highperformancengs_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        try:
            fn = getattr(L, name)  # AttributeError if the ABI and the header drift apart
        except AttributeError:
            if os.environ.get("HPN_LIB"):   # an A/B build of an older tree (scripts/ab_build.sh) may lack the newest entry points
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    if L.hpn_abi_version() != 1:
        raise RuntimeError("libhpngs.so ABI version mismatch")
    _lib = L
    return L


def _strerror(status):
    try:
        return lib().hpn_strerror(status).decode()
    except Exception:
        return "?"
