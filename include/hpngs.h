/*
 * hpngs.h -- C ABI of the MI355X (gfx950) scan kernels that stand in for the
 * per-record loops of HighPerformanceNGS' fastq_count / fastq_trim / bam2depth /
 * bam_sliding_count.
 *
 * The reference has no library API; its seams are three C call signatures
 * (SURVEY.md §8b).  Each entry point below names the reference function it
 * replaces (file:line relative to the reference checkout).  Conventions:
 *   - plain C types, no exceptions; every call returns HPN_OK (0) or a negative
 *     hpn_status; hpn_ctx_last_error() has the detail text;
 *   - one hpn_ctx per GPU, used by one host thread at a time;
 *   - "host" entry points take caller-owned host buffers, copy, launch, wait and
 *     ADD into caller-owned accumulators, exactly like count_read adds into the
 *     arrays its caller zeroed (fastq_count_kthread.c:116,126);
 *   - "_dev" entry points take device pointers (hipMalloc / hpn_dev_malloc /
 *     torch tensors), enqueue on the context's stream and return immediately;
 *     results are collected by the matching "_fetch" call;
 *   - there is NO CPU fallback: without a usable GPU every compute call fails
 *     with HPN_E_NODEVICE.
 */
#ifndef HPNGS_H
#define HPNGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPN_ABI_VERSION 1
#define HPN_LEN_BINS 512   /* SeqLen[512]       fastq_count.c:111 */
#define HPN_QUAL_ROWS 128  /* Quality[128][512] fastq_count.c:110 */
#define HPN_NUC_CODES 5    /* T,C,A,G,N         Rgzfastq_uniq.c:97-108 */

typedef enum hpn_status {
    HPN_OK = 0,
    HPN_E_NODEVICE = -1, /* no HIP device / runtime */
    HPN_E_HIP = -2,      /* a HIP call failed */
    HPN_E_ARG = -3,      /* bad argument */
    HPN_E_DOMAIN = -4,   /* input on which the reference has undefined behaviour
                            (read length >= 512, quality byte >= 128, position >= 2^28 ...) */
    HPN_E_NOMEM = -5,
    HPN_E_STATE = -6,    /* call sequence error (e.g. depth_add before depth_begin) */
    HPN_E_RCCL = -7,
    HPN_E_CAPACITY = -8, /* caller-provided output buffer too small; required size reported */
    HPN_E_PARTIAL = -9   /* a grouped collective failed after some ranks were enqueued: the vectors
                            may or may not hold sums and must not be added again (abandon the input) */
} hpn_status;

typedef struct hpn_ctx hpn_ctx;

/* ---- library / context ----------------------------------------------------- */
int hpn_abi_version(void);
const char *hpn_strerror(int status);
int hpn_device_count(int *n);
/* `device` = HIP ordinal.  Creates the context's own non-blocking stream. */
int hpn_ctx_create(int device, hpn_ctx **ctx);
int hpn_ctx_destroy(hpn_ctx *ctx);
/* Run on a caller-owned hipStream_t instead (e.g. torch's current stream);
 * NULL restores the context's own stream. */
int hpn_ctx_set_stream(hpn_ctx *ctx, void *hip_stream);
int hpn_ctx_sync(hpn_ctx *ctx);
/* The HIP ordinal the context was created on (a helper context for uploads beside it: host/gz_gpu.hpp). */
int hpn_ctx_device(const hpn_ctx *ctx, int *device);
/* The device's PCI address ("0000:c1:00.0" + NUL; HPN_E_ARG when `len` < 16).  What a host needs to keep the threads that
 * feed the device on the CPUs next to it: /sys/bus/pci/devices/<address>/local_cpulist (host/cpus.hpp: on a two-socket box a
 * reader on the far socket moves 38 GB/s to the device, one on the near socket 51 -- profiles/r05/numa_probe.txt).  The
 * reference has no counterpart: its readers are the threads of kt_for, wherever the scheduler puts them (klib/kthread.c:48). */
int hpn_ctx_pci_address(const hpn_ctx *ctx, char *buf, int len);
const char *hpn_ctx_last_error(const hpn_ctx *ctx);
/* Milliseconds the device spent in the most recent kernel launch group of the
 * given family, measured with hipEvents on the context's stream (valid after
 * the matching fetch/sync). Families: 0 tally, 1 trim, 2 depth, 3 window,
 * 4 text framing, 5 BGZF inflate, 6 record index of hpn_bam_raw_index_dev (its four
 * kernels and the host's look at the counts between them), 7 the field view
 * hpn_window_add_raw_dev makes of the indexed records. */
int hpn_ctx_last_kernel_ms(hpn_ctx *ctx, int family, float *ms);

/* ---- memory helpers (thin wrappers; callers may use their own allocator) ---- */
int hpn_dev_malloc(hpn_ctx *ctx, size_t bytes, void **dptr);
int hpn_dev_free(hpn_ctx *ctx, void *dptr);
int hpn_host_malloc(hpn_ctx *ctx, size_t bytes, void **hptr); /* pinned */
int hpn_host_free(hpn_ctx *ctx, void *hptr);
int hpn_memcpy_h2d(hpn_ctx *ctx, void *dst, const void *src, size_t bytes); /* async on ctx stream */
int hpn_memcpy_d2h(hpn_ctx *ctx, void *dst, const void *src, size_t bytes); /* async on ctx stream */
int hpn_memcpy_d2d(hpn_ctx *ctx, void *dst, const void *src, size_t bytes); /* async on ctx stream; same device */
int hpn_dev_mem_info(hpn_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes); /* of the context's device, now */

/* ---- fastq_count: replaces count_read's scan loop ------------------------------
 * fastq_count.c:112-119 / fastq_count_kthread.c:126-135 (+ AssignQuality :29-35).
 *
 * A batch is structure-of-arrays: record i owns bytes [off[i], off[i+1]) of
 * `qual` (the first seqLen bytes of its quality line) and, optionally, of
 * `base` (its sequence line).  Adds into `acc`:
 *   seqlen[l]            += #records of length l            (SeqLen[512])
 *   total, q20, q30      += #bytes, #bytes >= 53, >= 63     (statQ(...,53,...,63,...), :37-47,124)
 *   qual_hist[q*512+pos] += 1 per byte        if non-NULL   (Quality[128][512])
 *   nuc_hist[c*512+pos]  += 1 per base        if non-NULL and `base` given
 *                           (c: T0 C1 A2 G3 N4, any other byte counts as T,
 *                            Rgzfastq_uniq.c:97-108)
 * Domain (else HPN_E_DOMAIN, accumulators untouched): every length < 512,
 * every quality byte < 128. */
typedef struct hpn_tally {
    uint64_t seqlen[HPN_LEN_BINS];
    uint64_t total, q20, q30;
    uint64_t *qual_hist; /* [HPN_QUAL_ROWS * HPN_LEN_BINS] or NULL */
    uint64_t *nuc_hist;  /* [HPN_NUC_CODES * HPN_LEN_BINS] or NULL */
} hpn_tally;

#define HPN_TALLY_QUAL_HIST 1u /* also build Quality[128][512] */
#define HPN_TALLY_NUC_HIST 2u  /* also build Nucleotide[5][512] (needs base) */

int hpn_fastq_tally(hpn_ctx *ctx, const uint8_t *qual, const uint8_t *base_or_null,
                    const uint64_t *off, uint64_t n_records, hpn_tally *acc);
/* Device-resident batch; accumulates into the context's device accumulators. */
int hpn_fastq_tally_dev(hpn_ctx *ctx, const uint8_t *d_qual, const uint8_t *d_base_or_null,
                        const uint64_t *d_off, uint64_t n_records, uint32_t flags);
/* Wait, add the device accumulators into `acc`, zero them. */
int hpn_fastq_tally_fetch(hpn_ctx *ctx, hpn_tally *acc);
/* The raw device accumulator block (uint64_t[HPN_TALLY_WORDS], layout below) for
 * callers that reduce it across GPUs themselves (RCCL all-reduce). */
#define HPN_TALLY_W_SEQLEN 0                                    /* [512] */
#define HPN_TALLY_W_TOTAL 512
#define HPN_TALLY_W_Q20 513
#define HPN_TALLY_W_Q30 514
#define HPN_TALLY_W_BAD 515                                     /* domain violations seen */
#define HPN_TALLY_W_QUAL 516                                    /* [128*512] */
#define HPN_TALLY_W_NUC (516 + HPN_QUAL_ROWS * HPN_LEN_BINS)    /* [5*512] */
#define HPN_TALLY_WORDS (HPN_TALLY_W_NUC + HPN_NUC_CODES * HPN_LEN_BINS)
int hpn_fastq_tally_devptr(hpn_ctx *ctx, uint64_t **d_acc);

/* ---- Rfastqc tally: the per-read statistics of the R plugin -------------
 * Rgzfastq_uniq.c: STATSEQ (:50-57), AssignQuality (:42-48), Length (:174), as returned by
 * qsort_hash_count (:250) in list elements 2..5 -- in the plugin's own layouts:
 *   quality[q + 128*pos]      += 1 per quality byte q at cycle pos          int[128*300]
 *   nucleotide[5*pos + code]  += 1 per base (T/U 0, C 1, A 2, G 3, N '.' 4,   int[5*300]
 *                                 every other byte 0; :97-108)
 *   length[len - 1]           += 1 per read                                 int[300]
 *   gc[i]                      = #{'G','C'} / len of read i, as double       double[n]
 * Adds into quality / nucleotide / length (the plugin callocs them, :37-40) and writes
 * gc.  Any of the four pointers may be NULL.  The duplicate-count vector (list element
 * 1: hash of the first 50 bases, qsort) is the dedup family and is not produced here.
 * Domain (else HPN_E_DOMAIN): 1 <= len <= 300 (MaxLen; the plugin writes out of bounds
 * otherwise), quality byte < 128. */
#define HPN_RQC_MAXLEN 300
typedef struct hpn_rqc {
    int32_t *quality;
    int32_t *nucleotide;
    int32_t *length;
    double *gc;
} hpn_rqc;
int hpn_fastq_rqc(hpn_ctx *ctx, const uint8_t *seq, const uint8_t *qual, const uint64_t *off,
                  uint64_t n_records, hpn_rqc *out);
/* The GC-fraction part alone, on device-resident arrays (async on the context's stream). */
int hpn_fastq_read_gc_dev(hpn_ctx *ctx, const uint8_t *d_seq, const uint64_t *d_off, uint64_t n_records,
                          double *d_gc);

/* ---- fastq_trim: replaces readNextNode's cut --------------------------------------
 * fastq_trim.c:76-77,83-84: out = line[min(S,len) .. min(E,len)) for the sequence
 * and the quality line of every record.  out_off[0] = 0, out_off[i+1] = running
 * total; out_seq/out_qual are packed.  Capacity of out_seq/out_qual: off[n]-off[0]
 * bytes is always enough.  Domain: 0 <= S <= E. */
int hpn_fastq_trim(hpn_ctx *ctx, const uint8_t *seq, const uint8_t *qual, const uint64_t *off,
                   uint64_t n_records, int32_t S, int32_t E, uint8_t *out_seq, uint8_t *out_qual,
                   uint64_t *out_off);
int hpn_fastq_trim_dev(hpn_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual,
                       const uint64_t *d_off, uint64_t n_records, int32_t S, int32_t E,
                       uint8_t *d_out_seq, uint8_t *d_out_qual, uint64_t *d_out_off);

/* ---- EXTENSION: quality-threshold trim points ----------------------------------------
 * No reference counterpart: the reference's fastq_trim is the fixed-cycle cut above
 * (SURVEY.md §0.1 D3); BASELINE.json's north_star asks for a quality-threshold
 * trim-point scan as well.  Definition used here, per record:
 *   beg = index of the first quality byte >= threshold
 *   end = 1 + index of the last quality byte >= threshold     (beg = end = 0 if none)
 * hpn_fastq_trim_points cuts every record to its own [beg[i], end[i]) (clamped to the
 * record), with the same packed outputs as hpn_fastq_trim. */
int hpn_fastq_qtrim_points(hpn_ctx *ctx, const uint8_t *qual, const uint64_t *off, uint64_t n_records,
                           uint32_t threshold, uint32_t *beg, uint32_t *end);
int hpn_fastq_qtrim_points_dev(hpn_ctx *ctx, const uint8_t *d_qual, const uint64_t *d_off, uint64_t n_records,
                               uint32_t threshold, uint32_t *d_beg, uint32_t *d_end);
int hpn_fastq_trim_points(hpn_ctx *ctx, const uint8_t *seq, const uint8_t *qual, const uint64_t *off,
                          uint64_t n_records, const uint32_t *beg, const uint32_t *end, uint8_t *out_seq,
                          uint8_t *out_qual, uint64_t *out_off);
int hpn_fastq_trim_points_dev(hpn_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual, const uint64_t *d_off,
                              uint64_t n_records, const uint32_t *d_beg, const uint32_t *d_end,
                              uint8_t *d_out_seq, uint8_t *d_out_qual, uint64_t *d_out_off);

/* ---- raw FASTQ text front end: the 4 x gzgets framing on the device ----------------------
 * fastq_count.c:112-118 and fastq_trim.c:67-89 frame a stream with four gzgets()
 * calls per record into one 1024-byte buffer.  On REGULAR text that loop is a pure
 * function of the newline positions, and these calls evaluate it on the GPU from the
 * raw (decompressed) bytes: no host pass over the text at all.  Regular means
 *   - no NUL byte, every line at most 1022 characters + '\n' (gzgets would split it);
 *   - the stream ends with a whole record (count: the very last '\n' may be missing,
 *     the reference then drops nothing it would tally);
 *   - count: the quality line is not shorter than the sequence line (the reference
 *     would tally stale buffer bytes), read length < 512;
 *   - trim: sequence and quality line have the same length, S <= every read's length.
 * Anything else is DETECTED, never mis-framed: the call reports the reasons in
 * info->irregular, adds nothing, and closes the text stream; the caller then frames
 * that input with the exact gzgets emulation (csrc/host/fastq_reader.hpp) and
 * hpn_fastq_tally / hpn_fastq_trim.
 *
 * A stream is fed in chunks of any size (< 2^31 bytes) cut anywhere; the bytes of an
 * unfinished trailing record stay on the device and are prepended to the next chunk.
 * `text` may be a host pointer (pinned: the copy is asynchronous; the call returns
 * after the copy has completed, so the buffer may be reused at once) or a device
 * pointer.  `last` != 0 marks the final chunk (nbytes may be 0). */
typedef struct hpn_text_info {
    uint64_t n_records;   /* records framed by this call */
    uint64_t n_bytes;     /* count: sum of their lengths; trim: bytes of output text */
    uint64_t carry_bytes; /* tail bytes kept for the next chunk */
    uint32_t irregular;   /* HPN_TEXT_* reasons, 0 = chunk processed */
    uint32_t reserved;
} hpn_text_info;

#define HPN_TEXT_NUL 1u        /* NUL byte in the text */
#define HPN_TEXT_LONG_LINE 2u  /* line of 1023+ characters */
#define HPN_TEXT_RAGGED 4u     /* quality line shorter than (trim: different from) the sequence line */
#define HPN_TEXT_PARTIAL 8u    /* stream ends inside a record */
#define HPN_TEXT_LEN 16u       /* read of 512+ bases (count) */
#define HPN_TEXT_DENSE 32u     /* more than one line per 4 bytes: not worth indexing */
#define HPN_TEXT_STALE 64u     /* trim: S beyond a read's end (the reference copies what the record's
                                  earlier lines left in its buffer; the host framer reproduces that) */

int hpn_fastq_text_begin(hpn_ctx *ctx);
/* count_read's loop: adds into the context's device accumulators exactly like
 * hpn_fastq_tally_dev (collect with hpn_fastq_tally_fetch).  tally_flags: HPN_TALLY_*. */
int hpn_fastq_text_count(hpn_ctx *ctx, const void *text, uint64_t nbytes, int last, uint32_t tally_flags,
                         hpn_text_info *info);
/* The same for text that lies on the device already (a batch inflated there: hpn_gz_inflate_dev, hpn_bgzf_inflate_dev),
 * framed WHERE IT LIES: d_text must have 8192 writable bytes of device memory in front of it (the carried bytes of the
 * chunk before are laid there) and 64 readable bytes behind d_text + nbytes; nothing else is copied.  The text may be
 * overwritten as soon as the call returns.  Chunks framed in place and chunks framed by hpn_fastq_text_count may follow each
 * other in one stream. */
int hpn_fastq_text_count_inplace(hpn_ctx *ctx, const uint8_t *d_text, uint64_t nbytes, int last, uint32_t tally_flags,
                                 hpn_text_info *info);
/* gzfastq_sample.c:214-225 count_read: the same four gzgets per record with nothing but i++ in the
 * loop (its "total_reads_num").  Frames the chunk like hpn_fastq_text_count and tallies nothing:
 * info->n_records of every chunk add up to the reference's i (= the ReadCount column of fastq_count
 * for the same file).  Irregular text is reported the same way; the host framer then counts. */
int hpn_fastq_text_records(hpn_ctx *ctx, const void *text, uint64_t nbytes, int last, hpn_text_info *info);
/* readNextNode + fprintf("%s\n%s\n+\n%s\n") (fastq_trim.c:67-89,101): writes the
 * trimmed records of this chunk as text into out_text (host or device pointer,
 * capacity out_cap; nbytes + 8192 always suffices). */
int hpn_fastq_text_trim(hpn_ctx *ctx, const void *text, uint64_t nbytes, int last, int32_t S, int32_t E,
                        void *out_text, uint64_t out_cap, hpn_text_info *info);

/* ---- ONE text stream framed by several contexts (one per GPU): pieces ----------------------------
 * Record-block sharding of a single FASTQ input (the reference's parallelism stops at
 * whole files, fastq_count.c:213-230, klib/kthread.c:34-60).  The stream's bytes T[0, N) are cut at
 * arbitrary positions 0 = b_0 < b_1 < ... into pieces [b_j, b_j+1).  A piece OWNS the records whose
 * first byte lies in it.  Which lines start a record depends on how many lines the stream has before
 * the piece, so a piece is framed in two steps, each on the context that holds it:
 *
 *   hpn_fastq_text_piece_lines   text = T[b_j - head, b_j+1 + tail): `head` = 1 byte in front of the
 *       piece (0 for the stream's first piece), own_bytes = b_j+1 - b_j, and a tail of the following
 *       bytes -- 4096 hold the rest of any regular record that starts in the piece (fewer only where the
 *       stream ends); last != 0: the piece ends the stream (no tail).  Copies, indexes the lines and
 *       returns out->n_lines = number of '\n' in T[b_j - head, b_j+1 - 1): the n_lines of pieces
 *       0 .. j-1 add up to the lines in front of piece j's text -- the only thing pieces tell each other.
 *   hpn_fastq_text_piece_count / _trim   lines_before = that sum.  Frames the records the piece owns
 *       and tallies / trims them exactly like hpn_fastq_text_count / _trim do for a chunk.
 *
 * Irregular text (same conditions; a record that does not end inside the tail counts as a long line)
 * is reported in ->irregular by either call and nothing is added: the caller frames the WHOLE stream
 * another way.  No bytes are carried between pieces, so different contexts may work on different
 * pieces at the same time; one context handles one piece at a time. */
typedef struct hpn_text_piece {
    uint64_t n_lines;
    uint32_t irregular; /* HPN_TEXT_* reasons visible without record framing (NUL, DENSE) */
    uint32_t reserved;
} hpn_text_piece;
#define HPN_TEXT_PIECE_TAIL 4096u
int hpn_fastq_text_piece_lines(hpn_ctx *ctx, const void *text, uint64_t nbytes, uint32_t head, uint64_t own_bytes,
                               int last, hpn_text_piece *out);
int hpn_fastq_text_piece_count(hpn_ctx *ctx, uint64_t lines_before, uint32_t tally_flags, hpn_text_info *info);
int hpn_fastq_text_piece_trim(hpn_ctx *ctx, uint64_t lines_before, int32_t S, int32_t E, void *out_text,
                              uint64_t out_cap, hpn_text_info *info);

/* ---- BGZF: inflate on the device ------------------------------------------------------------
 * A BAM / bgzip file is a chain of independent gzip members of at most 64 KiB (SAM spec 4.1);
 * the reference inflates them one by one on the host (samtools-0.1.19 bgzf.c:214-307).  Here
 * the host only walks the 18-byte block headers; every block is decoded by one wavefront.
 * blocks[i]: in_off / in_len = the raw DEFLATE payload inside the compressed buffer (after the
 * member header, before the 8-byte trailer), out_off / out_len = where its ISIZE bytes go.
 * d_comp must be readable 64 bytes past the last payload.  d_status[i] = 0 or a decoder error
 * code (malformed stream, or output != out_len); like the reference's reader the CRC32 is not
 * checked.  Asynchronous on the context's stream. */
typedef struct hpn_bgzf_block {
    uint64_t in_off;
    uint32_t in_len, out_len;
    uint64_t out_off;
} hpn_bgzf_block;
int hpn_bgzf_inflate_dev(hpn_ctx *ctx, const uint8_t *d_comp, const hpn_bgzf_block *d_blocks, uint64_t n_blocks,
                         uint8_t *d_out, uint32_t *d_status);

/* ---- one gzip member inflated on the device (two passes) -------------------------------------
 * gzread behind the 4 x gzgets loops (fastq_count.c:112-118, IO_stream.h:122-136) on a single-
 * member .fastq.gz: one DEFLATE stream, every byte of which may refer to the 32 KiB before it.
 * The host cuts the stream into stretches that start at deflate block boundaries (it finds them,
 * csrc/host/pgz_reader.hpp find_start); here every stretch is inflated by one wavefront with the
 * history unknown (16-bit symbols, placeholders for history bytes), the histories are resolved in
 * stream order, and the symbols are translated into one contiguous text.
 * chunks[i]: in_off = byte of d_comp holding the stretch's first bit, start_bit = 0..7, in_len =
 * bytes readable from in_off (d_comp must be readable 64 bytes beyond), end_bit = where the
 * stretch must end in bits from in_off * 8 (the next stretch's first bit), ~0 for "at the final
 * block".  A stretch that does not end exactly there on a block boundary is an error: that is what
 * proves every start.  sym_cap (multiple of 8) = symbols of scratch per stretch; n_chunks <= 65535.
 * d_window_in / d_window_out (32768 bytes, may be NULL): the history before the first stretch /
 * after the last, for streams inflated in several calls.  info: n_bytes = text produced (if it
 * exceeds text_cap: HPN_E_CAPACITY, nothing written), status = 0 or the first failing stretch's
 * decoder code (bad_chunk says which), final_chunk = 1 + index of the stretch that ended at a
 * final block (0: none), end_bit = bit position that stretch reached (the member trailer).
 * Synchronous.  CRC-32 is not checked (ISIZE is the caller's to check).  in_len < 2^31.
 * status codes: 1-11 malformed block header or code tables, 12/14 out of symbol scratch (sym_cap), 13/15 invalid
 * code in the data, 17 ran past in_len, 20 the stretch did not end on end_bit at a block boundary, 22 a final block
 * inside a stretch that was given an end and nothing that starts another member behind it (the next stretch's start is
 * unproven), 23 more members ended inside the call than it has room for (65536), 24 a header too long to back out of.
 *
 * Several members (cat a.gz b.gz): a stretch that meets a final block looks behind the 8-byte trailer for the next
 * member's header (RFC 1952) and goes on with its first block.  hpn_gz_members() lists the members that ended inside the
 * last call -- text_end = bytes of the call's text up to the member's end, isize = the ISIZE field of its trailer -- in
 * text order, so that the caller can check every member's size; the last member of the file ends the last stretch
 * (hpn_gz_info.final_chunk, .end_bit) as a single member does. */
typedef struct hpn_gz_chunk {
    uint64_t in_off;
    uint64_t end_bit;
    uint32_t in_len;
    uint32_t start_bit;
} hpn_gz_chunk;
typedef struct hpn_gz_info {
    uint64_t n_bytes;
    uint64_t end_bit;
    uint32_t status, bad_chunk, final_chunk, reserved;
} hpn_gz_info;
typedef struct hpn_gz_member {
    uint64_t text_end;
    uint32_t isize, crc32; /* the two fields of the member's trailer (RFC 1952) */
} hpn_gz_member;
int hpn_gz_inflate_dev(hpn_ctx *ctx, const uint8_t *d_comp, const hpn_gz_chunk *d_chunks, uint32_t n_chunks,
                       uint32_t sym_cap, const uint8_t *d_window_in, uint8_t *d_text, uint64_t text_cap,
                       uint8_t *d_window_out, hpn_gz_info *info);
int hpn_gz_members(hpn_ctx *ctx, hpn_gz_member *out, uint32_t cap, uint32_t *n); /* HPN_E_CAPACITY: *n tells how many */
/* Decoder wavefronts the context's device runs at once (CUs x decoders per CU): one stretch (hpn_gz_inflate_dev) or one BGZF
 * block (hpn_bgzf_inflate_dev) each -- the number of stretches that fills the chip exactly once.  Hosts size their batches by it. */
int hpn_inflate_slots(hpn_ctx *ctx, uint32_t *n_slots);
/* The same call in two halves, for ONE file whose batches of stretches go to several contexts (one per GPU) in turn: the
 * symbolic decode of a batch needs nothing of the text in front of it, only the resolution of the histories does (the 32 KiB
 * window the batch before ends with).  _begin_dev starts the decode on the context's stream and returns; _finish_dev takes the
 * window (d_window_in: on THIS context's device; NULL for the file's first batch), resolves, translates into d_text and
 * reports like hpn_gz_inflate_dev; d_window_out receives this batch's last 32 KiB for the next one (the caller carries it
 * to the next context's device).  HPN_E_CAPACITY from _finish_dev leaves the decoded symbols in place: call it again with
 * the room info->n_bytes asks for.  hpn_gz_inflate_dev = _begin_dev + _finish_dev.  gzread behind the reference's gzgets
 * (IO_stream.h:122-136, fastq_count.c:112-118) is what the pair stands in for. */
int hpn_gz_inflate_begin_dev(hpn_ctx *ctx, const uint8_t *d_comp, const hpn_gz_chunk *d_chunks, uint32_t n_chunks, uint32_t sym_cap);
int hpn_gz_inflate_finish_dev(hpn_ctx *ctx, const uint8_t *d_window_in, uint8_t *d_text, uint64_t text_cap, uint8_t *d_window_out,
                              hpn_gz_info *info);

/* CRC-32 (RFC 1952) of byte ranges of a device buffer: crc[k] of d_data[spans[k].off .. + spans[k].len).  gzread, which the
 * reference reads every input through (IO_stream.h:122-136), verifies each gzip member's CRC-32 and stops handing out bytes
 * where one fails; a member inflated on the device is verified with this (in pieces, as the text comes: hpn_crc32_join folds
 * CRC(A), CRC(B), |B| into CRC(A || B)).  spans and crc are host arrays.  Synchronous. */
typedef struct hpn_span {
    uint64_t off, len;
} hpn_span;
int hpn_crc32_dev(hpn_ctx *ctx, const uint8_t *d_data, const hpn_span *spans, uint32_t n_spans, uint32_t *crc);
uint32_t hpn_crc32_join(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);

/* Where deflate blocks start, by trial, on the device: found[k] = the first bit position in [slices[k].off, slices[k].off +
 * slices[k].len) (bits, from d_comp) at which a non-final dynamic-Huffman block decodes to text and something that begins
 * like a block follows, or ~0.  What the host's gz_find_block_start does on the cores (csrc/host/pgz_reader.hpp); a start
 * found here is a proposal like one found there: it is proven by the stretch before it arriving exactly there
 * (hpn_gz_inflate_dev, status 20 otherwise).  d_comp: comp_bytes bytes, readable for 256 more.  slices / found: host
 * arrays.  Synchronous. */
int hpn_gz_find_starts_dev(hpn_ctx *ctx, const uint8_t *d_comp, uint64_t comp_bytes, const hpn_span *slices, uint32_t n,
                           uint64_t *found);

/* ---- BAM record batches ---------------------------------------------------------------
 * What the reference's bam_fetch_f callback sees per record (bam.h:178-187,627),
 * flattened to SoA by the host decoder: core fields, CIGAR words (len<<4|op,
 * bam.h:97-110) and the 4-bit packed sequence (bam.h:260). */
typedef struct hpn_bam_batch {
    uint64_t n;
    const int32_t *tid;
    const int32_t *pos;
    const uint32_t *flag;
    const int32_t *l_qseq;
    const uint32_t *cigar_off; /* [n+1] into cigar */
    const uint32_t *cigar;
    const uint64_t *seq_off;   /* [n+1] into seq4 (bytes); may be NULL for depth */
    const uint8_t *seq4;       /* may be NULL for depth */
} hpn_bam_batch;

/* ---- bam2depth: replaces fetch_func + hash2BedGraph + overlap ----------------------
 * bam2depth.c:86-110 (CIGAR-M difference array), :203-236 (sorted sweep -> runs of
 * equal depth > 0), :132-176 (window sums).  One target (chromosome) at a time:
 *   hpn_depth_begin(tid, target_len, flag_mask)   flag_mask = 0x704 for bam2depth
 *                                                 (BAM_DEF_MASK, bam.h:124), 0x4 for bam2wig
 *   hpn_depth_add(batch)            any number of times; records of other tids are skipped
 *   hpn_depth_finish(W, ...)        runs (start,end,depth) ascending + per-window sum of coverage
 * win_sum has target_len/W+1 entries (bam2depth.c:326) holding the exact integer
 * sum of coverage over [kW, min((k+1)W, target_len)); the host prints win_sum/W
 * with %.2f (output_bins :238-246).
 * Domain: every breakpoint < 2^28 (int2char keeps 28 bits, hashtbl.c:243-249). */
typedef struct hpn_run {
    int32_t start, end, depth;
} hpn_run;

int hpn_depth_begin(hpn_ctx *ctx, int32_t tid, uint32_t target_len, uint32_t flag_mask);
/* The same, telling the window size hpn_depth_finish will be called with.  A BAM that bam2depth can read is coordinate-
 * sorted (it needs the index, bam2depth.c:112-119), and on sorted input the work is done WHILE the records come: every
 * stretch of the target that the records of an hpn_depth_add call have moved beyond is swept at once -- prefix sum, runs,
 * window sums, in the workgroup that gathered its breakpoints -- and never goes through memory as a difference array.
 * Window sums can only ride along if W is known by then; with W = 0 (hpn_depth_begin) or another W at hpn_depth_finish
 * they are taken from the runs afterwards (same numbers, one more pass over the runs).
 * This expects the records of the target in coordinate order ACROSS calls (inside a call any order is handled).  A record
 * that arrives behind positions already swept makes hpn_depth_finish fail with HPN_E_STATE; callers with input in any
 * order put HPN_DEPTH_ANY_ORDER into flag_mask: nothing is swept early then. */
#define HPN_DEPTH_ANY_ORDER 0x80000000u
int hpn_depth_begin_w(hpn_ctx *ctx, int32_t tid, uint32_t target_len, uint32_t flag_mask, uint32_t W);
/* How far the target has been swept by the hpn_depth_add calls so far: positions [0, *swept_positions) are final (waits for the
 * calls' kernels).  0 with HPN_DEPTH_ANY_ORDER, or while no batch could be taken that way. */
int hpn_depth_progress(hpn_ctx *ctx, uint64_t *swept_positions);
int hpn_depth_add(hpn_ctx *ctx, const hpn_bam_batch *host_batch);
int hpn_depth_add_dev(hpn_ctx *ctx, const hpn_bam_batch *dev_batch);
/* runs: caller buffer of runs_cap entries; *n_runs receives the number found
 * (HPN_E_CAPACITY if it exceeds runs_cap; call again with a larger buffer --
 * the scan result stays valid until the next hpn_depth_begin). */
/* (runs == NULL with runs_cap == 0: the runs stay on the device, *n_runs is still set -- for callers that
 * take the bedGraph text instead, below.) */
int hpn_depth_finish(hpn_ctx *ctx, uint32_t W, hpn_run *runs, uint64_t runs_cap, uint64_t *n_runs,
                     uint64_t *win_sum);
/* hash2BedGraph's fprintf(bedGraph, "%s\t%d\t%d\t%d\n", chr, start, end, depth) (bam2depth.c:217) for every
 * run of the last hpn_depth_finish, formatted on the device: *n_bytes = size of the text, which stays on
 * the device until the next hpn_depth_begin / hpn_depth_bedgraph_format; hpn_depth_bedgraph_read copies
 * text[offset, offset + nbytes) to `dst` (host; pinned memory makes the copy faster). */
int hpn_depth_bedgraph_format(hpn_ctx *ctx, const char *target_name, uint64_t *n_bytes);
int hpn_depth_bedgraph_read(hpn_ctx *ctx, uint64_t offset, void *dst, uint64_t nbytes);
/* Where the text of the last hpn_depth_bedgraph_format lies on the device, for a caller that copies it out itself -- through
 * another context's stream (hpn_memcpy_d2h), beside the kernels of the NEXT target: the bytes stay put until this context's
 * next hpn_depth_bedgraph_format (hpn_depth_begin / _add / _finish do not touch them).  bam2depth's writer does this. */
int hpn_depth_bedgraph_dev(hpn_ctx *ctx, const uint8_t **d_text, uint64_t *n_bytes);

/* ---- bam_sliding_count: replaces fetch_func + cal_GC ----------------------------------
 * bam_sliding_count.c:84-124.  Slot of a record = win_off[tid] + (uint16)(pos/W)
 * (:117, including the 16-bit wrap).  Adds per slot: bins += 1, gc += #nibbles
 * equal to 2 (C) or 4 (G), len += l_qseq; sets touched[tid].  Records with tid<0
 * or flag&4 are skipped (:96-97).  All integer; the float32 replay of
 * calc_winGC (:126-138) is host work (csrc/host/report.cpp). */
int hpn_window_begin(hpn_ctx *ctx, int32_t n_targets, const uint64_t *win_off /*[n_targets+1]*/,
                     uint32_t W);
int hpn_window_add(hpn_ctx *ctx, const hpn_bam_batch *host_batch);
int hpn_window_add_dev(hpn_ctx *ctx, const hpn_bam_batch *dev_batch);
int hpn_window_finish(hpn_ctx *ctx, uint32_t *bins, uint64_t *gc, uint32_t *len, uint8_t *touched,
                      uint64_t *n_count);

/* ---- BAM records in place in inflated BGZF blocks -----------------------------------------
 * bam_read1 (samtools-0.1.19 bam.c:191) on the device: after hpn_bgzf_inflate_dev the records
 * are indexed where they lie.  The inflated blocks of a call are ONE stream d_raw[0, out_off + out_len
 * of the last block): `first_off` is the stream offset of its first record (inside block 0: the block
 * the BAM header ends in; 0 when the caller has put the unfinished record of the call before in front,
 * see tail_bytes).  Where every block's first record starts is found on the device and proven for the
 * whole call (every block's chain must arrive exactly at the next block's found start: by induction
 * from first_off all starts are true), so both writers' files decode: samtools never lets a record
 * straddle a block (bam.c:238 bgzf_flush_try), htsjdk packs records across blocks -- which bgzf_read
 * (bgzf.c:342) hides from the reference.  info->flags: 1 = the proof failed or a record is impossible
 * (a damaged file, or records longer than a block), 2 = a block failed to inflate -- in both cases
 * nothing is indexed and the caller decodes the file on the host; 4 (informational) = records do run
 * across block ends.  info->tail_bytes: the call's stream ends inside a record of which that many bytes
 * are there (not indexed): the caller copies d_raw[stream end - tail_bytes, stream end) in front of the
 * next call's stream (out_off of its blocks moved up by as much) and passes first_off = 0.
 * Synchronous (returns the record count and the refID range of the batch).  The two add calls then run
 * the depth / window kernels over the indexed records, like hpn_depth_add_dev / hpn_window_add_dev.
 * d_raw must be readable 16 bytes past the end of the inflated stream (the window kernel
 * reads packed sequences in unaligned 16-byte pieces). */
typedef struct hpn_raw_info {
    uint64_t n_records;
    int32_t tid_min, tid_max;
    uint32_t flags, tail_bytes;
} hpn_raw_info;
int hpn_bam_raw_index_dev(hpn_ctx *ctx, const uint8_t *d_raw, const hpn_bgzf_block *d_blocks, uint64_t n_blocks,
                          uint32_t first_off, const uint32_t *d_status, hpn_raw_info *info);
int hpn_depth_add_raw_dev(hpn_ctx *ctx, const uint8_t *d_raw);
int hpn_window_add_raw_dev(hpn_ctx *ctx, const uint8_t *d_raw);

/* ---- multi-GPU reduction of count vectors -----------------------------------
 * One RCCL communicator per context; `unique_id` is the 128-byte ncclUniqueId
 * produced by hpn_comm_unique_id on rank 0 and handed to every rank by the host
 * program (file, pipe, torch.distributed ...).  hpn_allreduce_u64 sums a device
 * vector in place over xGMI. */
#define HPN_UNIQUE_ID_BYTES 128
int hpn_comm_unique_id(uint8_t id[HPN_UNIQUE_ID_BYTES]);
int hpn_comm_init(hpn_ctx *ctx, int rank, int n_ranks, const uint8_t id[HPN_UNIQUE_ID_BYTES]);
int hpn_comm_destroy(hpn_ctx *ctx);
int hpn_allreduce_u64(hpn_ctx *ctx, uint64_t *d_vec, size_t n);
/* The same for ONE process that owns several contexts (the C tools sharding one input over the node's
 * GPUs): hpn_comm_init_all builds one communicator per context (ncclCommInitAll; the contexts must sit
 * on n distinct devices, else HPN_E_ARG and nothing is made -- callers then add the vectors on the
 * host), hpn_allreduce_u64_all sums d_vecs[i] (on ctxs[i]'s device) in place into every one of them in
 * one RCCL group, each on its context's stream; the call returns when all have finished.  This is where
 * reduceStats' element-wise sum goes (fastq_count_kthread.c:180-210).  A failure before anything was
 * enqueued (HPN_E_RCCL / HPN_E_HIP) leaves the vectors as they were: the caller may add them on the
 * host; HPN_E_PARTIAL does not.  No path returns with the RCCL group open.  hpn_comm_library: the path of
 * the RCCL library the binding resolved to ("" before the first use / when none could be loaded). */
int hpn_comm_init_all(hpn_ctx **ctxs, int n);
int hpn_allreduce_u64_all(hpn_ctx **ctxs, uint64_t **d_vecs, int n, size_t n_words);
const char *hpn_comm_library(void);
/* Ranks of the context's communicator as RCCL itself reports them (ncclCommCount): what bench.py
 * prints beside n_gpus, so that a job that believes it has N ranks and a communicator of one cannot
 * pass for an N-GPU run.  HPN_E_STATE without a communicator. */
int hpn_comm_count(hpn_ctx *ctx, int *n_ranks);

/* ---- synthetic inputs, generated in HBM -----------------------------------
 * Counter-based: byte k of record r depends only on (seed, r, k), so any shard
 * can be produced independently and checked on the CPU.  Quality bytes uniform
 * in 35..74, bases A/C/G/T 24.75 % each + N 1 %.  Fixed read length `len`;
 * d_off gets (first_record-relative) offsets i*len as uint64. */
int hpn_synth_fastq_dev(hpn_ctx *ctx, uint64_t seed, uint64_t first_record, uint64_t n_records,
                        uint32_t len, uint8_t *d_qual, uint8_t *d_base_or_null, uint64_t *d_off);

#ifdef __cplusplus
}
#endif
#endif /* HPNGS_H */
