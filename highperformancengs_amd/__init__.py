"""MI355X-native scan kernels behind HighPerformanceNGS' fastq_count / fastq_trim /
bam2depth / bam_sliding_count.  The product is the C ABI in
include/hpngs.h (libhpngs.so) plus the CLI tools; this package is the thin
Python plumbing tests and bench.py use."""
from ._lib import HpnError, LIB_PATH  # noqa: F401
from .api import Context, TallyResult, comm_unique_id, device_count  # noqa: F401
