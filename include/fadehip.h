/*
 * fadehip.h — C ABI of the MI355X (gfx950) `fade annotate` hot path.
 *
 * Drop-in boundary.  The reference (blachlylab/fade) has no plugin API; its hot path is a
 * synchronous per-record FFI seam:
 *     auto p   = Parasail("ACTGN", 10, 2, 2, -3);          source/anno.d:36
 *     auto res = p.sw_striped(q_seq, ref_seq);             source/analysis.d:67
 * inside  annotateTask (source/anno.d:55-110)  <-  foreach(rec; parallel(bam.allRecords))
 * (source/anno.d:44-50).  One call per clip cannot feed a GPU, so this ABI exposes the same
 * contract in two batched forms:
 *
 *   Level 1  fadehip_sw_*        — a batch of (query, reference) strings -> {score, position, cigar}
 *                                   replaces analysis.d:67 (the dparasail/libparasail call).
 *   Level 2  fadehip_annotate_*  — a batch of BAM-native read records against a genome resident
 *                                   in HBM -> rs byte per read + the alignment of every read
 *                                   that was re-aligned.  Replaces the body of annotateTask:
 *                                   anno.d:61-74 (gate, parse_clips, SA), analysis.d:34-64
 *                                   (floor, reverse complement util.d:23-34, window, FASTA fetch),
 *                                   analysis.d:67 (SW), analysis.d:69-83,98-107 (artifact gates),
 *                                   readstatus.d:5-26 (rs).  The caller keeps I/O and formats the
 *                                   am/as/ar/ab strings (analysis.d:84-92,108-118, anno.d:94-107).
 *
 * Plain C: pointers and sizes only.  Every function returns 0 on success or a negative
 * FADEHIP_E_* code and never throws or aborts across the boundary; fadehip_last_error(ctx) returns
 * the message of the calling thread's last failure on that ctx.  A ctx is bound to one device; its slots may be driven
 * by one host thread or by one thread each (create, destroy, genome_upload and sw_batch by one thread while no slot is busy).
 * There is no CPU fallback: without a usable HIP device fadehip_create fails.
 */
#ifndef FADEHIP_H
#define FADEHIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FADEHIP_ABI_VERSION 3
#define FADEHIP_MAX_OPS 16   /* ops reported per alignment (FADE rejects > 10, analysis.d:69) */
#define FADEHIP_MAX_QUERY 512 /* longest query of the wave kernels (8 alignments per wavefront) */
#define FADEHIP_MAX_LONG_QUERY 32768 /* longer queries, up to this, take a thread-per-alignment kernel (slow path) */
#define FADEHIP_NUM_SLOTS 4   /* batches in flight per ctx (each slot has its own stream, device buffers and pinned result block) */
/* The CIGAR ops dhtslib's Cigar.alignedLength sums (analysis.d:53,111-113; filter.d:27,61): M, D, N, =, X — bit k set for
 * BAM op code k.  One definition for the gate kernel, the host-side bounds and the tag formatters (the oracle restates it
 * as fo_cigar_aligned_length).  [The dhtslib commit FADE pins is not available here; if its alignedLength also counted I,
 * this constant is the only place to change.] */
#define FADEHIP_REF_CONSUMING_OPS 0x18Du
#define FADEHIP_OP_CONSUMES_REF(op) ((FADEHIP_REF_CONSUMING_OPS >> ((op) & 15u)) & 1u)

enum {
    FADEHIP_OK = 0,
    FADEHIP_E_INVALID = -1,     /* bad argument */
    FADEHIP_E_NODEVICE = -2,    /* no HIP device / wrong architecture */
    FADEHIP_E_HIP = -3,         /* a HIP runtime call failed (message has the HIP error) */
    FADEHIP_E_NOMEM = -4,       /* host or device allocation failed */
    FADEHIP_E_UNSUPPORTED = -5, /* scoring parameters, read or window length outside kernel limits */
    FADEHIP_E_STATE = -6,       /* call out of order (e.g. wait on an idle slot) */
    FADEHIP_E_RESIDUE = -7,     /* FASTA contains a byte this encoding cannot represent ('=') */
    FADEHIP_E_RCCL = -8         /* an RCCL call failed */
};

typedef struct fadehip_ctx fadehip_ctx;

/* Parasail("ACTGN", open, ext, match, mismatch) — anno.d:36.  The alphabet is fixed to
 * A,C,T,G,N + wildcard (every other residue scores 0), as parasail_matrix_create builds it.
 * Accepted: 0 < ext <= open, and 0 <= score + open <= 15 for score in {match, mismatch, 0} (4-bit profile
 * entries).  FADE's 10/2/2/-3 and any match <= 2 run the two-pass path; larger match scores exceed the 16-bit
 * ranges of its score pass and take the single-pass packed kernel (match <= 7) or the int32 kernel: same results,
 * lower rate. */
typedef struct {
    int32_t open;      /* 10: cost of the first gap base */
    int32_t ext;       /* 2 : cost of each further gap base */
    int32_t match;     /* 2 */
    int32_t mismatch;  /* -3 */
    int32_t max_ref_len;   /* longest reference window re-aligned; 0 -> 2^20, which is also the most.  Windows of up to 32,000
                              columns take the 16-lane wave kernels (8 alignments per wavefront; the window streams through LDS in
                              chunks), up to 65,000 — and reads of 513 .. 4,096 bases — the one-alignment-per-wavefront kernel,
                              beyond that the thread-per-alignment kernel (slow path); a read whose window is longer than
                              max_ref_len is left un-re-aligned and counted in n_oversize */
    int32_t max_batch_reads; /* capacity of one annotate batch; 0 -> 1<<20 */
    int64_t trace_bytes;   /* device bytes reserved for trace tables / checkpoints; 0 -> sized on demand */
    int32_t trace_all;     /* 1: level 2 returns a CIGAR for every re-aligned read (default: only for reads whose
                              score and end cell can still pass FADE's gates; the others have sw.n_ops == 0) */
    uint32_t rules;        /* FADEHIP_RULE_* bits; 0 -> FADEHIP_RULES_DEFAULT */
} fadehip_params;

/* Behaviour of the un-vendored arithmetic behind analysis.d:67 (libparasail 2.4.3 through dparasail 0.3.3) that could
 * not be checked against its source in this environment (DESIGN.md "parity unpinned").  Each
 * assumption is a switch with the same meaning and bit as oracle/fade_oracle.h's FO_RULE_*: should a maintainer find
 * that parasail does the opposite, it is a parameter, not a kernel edit.  Non-default settings run a slower variant of
 * the traced re-computation; the default costs nothing. */
enum {
    FADEHIP_RULE_END_MIN_REF_THEN_QUERY = 1u << 0, /* A.3 end cell: max H, ties -> smallest ref index, then smallest query index (off: first in row-major order) */
    FADEHIP_RULE_HDIR_DIAG_F_E = 1u << 1,          /* A.4 traceback priority DIAG > F (query-only) > E (ref-only)  (off: DIAG > E > F) */
    FADEHIP_RULE_GAP_TIE_EXTENDS = 1u << 2,        /* A.4 a gap "opens" only on strict >, ties extend  (off: ties open) */
    FADEHIP_RULE_EQ_BY_CHAR = 1u << 3,             /* A.4 '=' vs 'X' by residue equality  (off: by the sign of the matrix entry) */
    FADEHIP_RULE_SAM_GAP_LETTERS = 1u << 4,        /* A.5 ref-only step 'D', query-only step 'I'  (off: swapped) */
    FADEHIP_RULE_PAD_SOFTCLIP = 1u << 5,           /* A.6 CIGAR padded with S for unaligned query ends  (off: no padding) */
    FADEHIP_RULE_N_MATCHES_N = 1u << 6,            /* A.1 N vs N scores `match`  (off: `mismatch`) */
    FADEHIP_RULES_DEFAULT = 0x7f
};

void fadehip_params_default(fadehip_params *p);
int fadehip_abi_version(void);

/* device < 0 -> current device.  Fails with FADEHIP_E_NODEVICE when no gfx950 device exists. */
int fadehip_create(fadehip_ctx **out, int device, const fadehip_params *params);
void fadehip_destroy(fadehip_ctx *ctx);
const char *fadehip_last_error(const fadehip_ctx *ctx); /* never NULL; ctx may be NULL */

/* Pinned host memory so that submit can use hipMemcpyAsync.  fadehip_host_register pins memory the caller already has
 * (and may have been filling before the ctx existed: a reader that starts while the device is still being brought up);
 * fadehip_host_free takes the registration back and leaves the memory to the caller. */
int fadehip_host_alloc(fadehip_ctx *ctx, size_t bytes, void **out);
int fadehip_host_free(fadehip_ctx *ctx, void *p);
int fadehip_host_register(fadehip_ctx *ctx, void *p, size_t bytes);

/* ------------------------------------------------------------------ Level 1: the SW seam -- */
/* What FADE reads from a parasail result (analysis.d:69-113): res.score, res.position
 * (= beg_ref), res.cigar (BAM-encoded len<<4|op with "MIDNSHP=X", soft-clip padded). */
typedef struct {
    int32_t score;
    int32_t end_query, end_ref; /* 0-based inclusive end cell */
    int32_t beg_query, beg_ref; /* 0-based; beg_ref is dparasail's `position` */
    int32_t n_ops;              /* true op count; only min(n_ops, FADEHIP_MAX_OPS) stored */
    uint32_t ops[FADEHIP_MAX_OPS];
} fadehip_sw_result;

/* n alignments; strings are ASCII, concatenated, q_off/r_off hold n+1 offsets.  Synchronous. */
int fadehip_sw_batch(fadehip_ctx *ctx, int32_t n, const uint8_t *q, const int64_t *q_off,
                     const uint8_t *r, const int64_t *r_off, fadehip_sw_result *out);

/* ------------------------------------------------------ Level 2: annotateTask over a batch -- */
/* Upload the indexed FASTA once (what IndexedFastaFile + fetchSequence serve, analysis.d:63).
 * seqs[c] holds lengths[c] ASCII residues (any case; upper-cased on device as analysis.d:63 does).
 * Stored in HBM as 4-bit codes, two bases per byte. */
int fadehip_genome_upload(fadehip_ctx *ctx, int32_t n_contigs, const int64_t *lengths,
                          const uint8_t *const *seqs);

/* BAM-native structure-of-arrays view of n records (what annotateTask reads from a bam1_t). */
typedef struct {
    int32_t n_reads;
    const int32_t *tid;        /* [n] core.tid */
    const int32_t *pos;        /* [n] core.pos, 0-based */
    const uint16_t *flag;      /* [n] core.flag */
    const uint8_t *has_sa;     /* [n] 1 iff the record carries an SA aux tag (anno.d:73) */
    const int32_t *l_seq;      /* [n] core.l_qseq */
    const uint32_t *cigar_off; /* [n+1] index of the record's first op in cigar_ops */
    const uint32_t *cigar_ops; /* BAM-encoded ops, all records concatenated */
    const uint32_t *seq_off;   /* [n+1] byte offset of the record's packed sequence */
    const uint8_t *seq_packed; /* BAM 4-bit sequence bytes, each record byte-aligned.  Only mapped records with an S op
                                  are ever read (anno.d:61): the others may have an empty slice, seq_off[i+1] == seq_off[i] */
    /* ABI 2.  anno.d:61-65 gives an unmapped record, or one without an S op, rs = 0 before looking at anything else:
     * a caller may leave such records out of the batch (they need not cross PCIe) and pass their number here; it is
     * added to the read_count of the batch's stats.  rs / read_idx then index the records that were sent. */
    int32_t n_skipped;
    /* Upper bound of cigar.alignedLength (analysis.d:53) over the batch, or 0 if the caller does not know it: the
     * launches of a run are sized from it (window <= bound + 2 * window-size).  With 0 the library scans the CIGARs
     * itself (about 1 ms per 10^5 records on one host thread).  A bound that turns out too small fails the batch at
     * collect (FADEHIP_E_INVALID), it is never trusted for memory safety. */
    int32_t ref_span_bound;
    /* ABI 3.  What a caller that fills its own block knows for free and the library otherwise finds with one pass over
     * the records (about 1 ms per 10^6 records on one host thread): the number of records that carry bases (a non-empty
     * seq_packed slice — only those can be re-aligned) and the shortest / longest l_seq among THOSE.  With
     * n_with_seq > 0 (and ref_span_bound > 0) fadehip_annotate_upload does no per-record host work: the launches and
     * result buffers of the run are sized from these bounds, and cigar_off / seq_off / l_seq are validated by the gate
     * kernel before anything is dereferenced through them.  A bound that turns out too small fails the batch at results
     * (FADEHIP_E_INVALID); it is never trusted for memory safety.  0 = unknown. */
    int32_t n_with_seq;
    int32_t l_seq_min, l_seq_max;
    int32_t reserved;
} fadehip_read_batch;

/* One pinned block for the nine arrays of a batch, so that upload is ONE hipMemcpyAsync at PCIe speed: allocate
 * fadehip_batch_bytes(...) with fadehip_host_alloc, let fadehip_batch_bind point the arrays into it (canonical order,
 * 256-byte aligned), fill them through the pointers (cigar_off[n] = n_cigar_ops, seq_off[n] = n_seq_bytes).  Arrays
 * laid out any other way are first gathered into the slot's own pinned staging block (one host copy more). */
size_t fadehip_batch_bytes(int32_t n_reads, int64_t n_cigar_ops, int64_t n_seq_bytes);
int fadehip_batch_bind(void *base, int32_t n_reads, int64_t n_cigar_ops, int64_t n_seq_bytes, fadehip_read_batch *b);

/* One entry per read that was re-aligned (clip longer than min-length), in no particular order.
 * sw.score / end_query / end_ref are always set.  Unless params.trace_all, the traceback (beg_*, ops)
 * is only completed when the result can still be an artifact call; otherwise sw.n_ops == 0, beg_* == -1
 * (score or end cell already fail analysis.d:74-80 / 98-104, or the CIGAR has more than 10 ops, which
 * analysis.d:69-70 rejects whatever they are). */
typedef struct {
    int32_t read_idx;
    int32_t art;         /* bit0 art_left, bit1 art_right (analysis.d:82,106) */
    int64_t win_start;   /* `start` of analysis.d:45-51 */
    int32_t win_len;     /* ref_seq.length */
    int32_t clip_left, clip_right; /* parse_clips(rec.cigar) lengths (anno.d:68) */
    int32_t aligned_len; /* rec.cigar.alignedLength (analysis.d:53) */
    fadehip_sw_result sw;
} fadehip_aln;

typedef struct {
    uint8_t *rs;          /* [n_reads] out: ReadStatus.raw per read (anno.d:63,94) */
    fadehip_aln *aln;     /* [aln_cap] out */
    int32_t aln_cap;      /* in: capacity of aln (n_reads is always enough) */
    int32_t n_aln;        /* out */
    int64_t stats[8];     /* out: stats.d:45-54 over this batch: read_count, clipped, sup, art_sup,
                             art, art_mate, aln_l, aln_r */
    int32_t n_oversize;   /* out: soft-clipped reads left un-re-aligned because the read (> FADEHIP_MAX_LONG_QUERY) or its
                             window (> max_ref_len) exceeds the kernels' limits; their rs keeps the sc / sup bits */
    int32_t reserved;
} fadehip_anno_out;

/* Zero-copy form of the results: pointers into the slot's pinned result block, valid until the slot is RUN again. */
typedef struct {
    const uint8_t *rs;        /* [n_reads] */
    const fadehip_aln *aln;   /* [n_aln] */
    int32_t n_reads, n_aln;
    int64_t stats[8];
    int32_t n_oversize;       /* as in fadehip_anno_out */
    int32_t reserved;
} fadehip_anno_view;

/* Asynchronous pipeline, slot in [0, FADEHIP_NUM_SLOTS); one host thread can keep every slot busy:
 *   upload  : ONE hipMemcpyAsync of the batch block (see fadehip_batch_bind) on the slot's copy stream, into the input
 *             buffer the slot's run in flight does not use; returns at once.  Uploading batch k + 1 right after run(k)
 *             takes the H2D off the slot's critical path; the results of run(k) stay valid until run(k + 1)
 *   run     : enqueues gate -> score pass (SW score, end cell, wave snapshots) -> selection of the alignments that can
 *             still be artifact calls -> device-side plan of the traced re-computation -> traced re-computation of
 *             their last steps -> traceback + artifact gates -> D2H of rs / aln / counters into the slot's pinned
 *             result block.  Nothing is read back in between: launches are sized from host-side bounds (read lengths,
 *             ref_span_bound), counts stay on the device, persistent waves draw work from device-side tables.
 *             Returns as soon as everything is enqueued (FADEHIP_KERNEL=pk|int32, the single-pass kernels kept for
 *             A/B runs, still block on the gate's counters).
 *   results : waits for the slot, then hands out views (collect: copies them into the caller's arrays).  Errors the
 *             device found in the batch (bad tid, window beyond max_ref_len, ...) are reported HERE, not by run.  If
 *             the traced re-computation needed more scratch than the slot held, the scratch grows and the batch runs
 *             again inside this call (a warm-up effect; FADEHIP_DEBUG=1 reports it).
 * submit = upload + run.  The batch arrays handed to upload / submit must stay valid and unchanged until the results /
 * collect of the run that consumes them has returned.  floor_len = --min-length (app.d:17), window = --window-size (app.d:18). */
int fadehip_annotate_upload(fadehip_ctx *ctx, int slot, const fadehip_read_batch *batch);
int fadehip_annotate_run(fadehip_ctx *ctx, int slot, int32_t floor_len, int32_t window);
int fadehip_annotate_submit(fadehip_ctx *ctx, int slot, const fadehip_read_batch *batch,
                            int32_t floor_len, int32_t window);
int fadehip_annotate_results(fadehip_ctx *ctx, int slot, fadehip_anno_view *out);
int fadehip_annotate_collect(fadehip_ctx *ctx, int slot, fadehip_anno_out *out);
int fadehip_sync(fadehip_ctx *ctx);

/* Measurement (after results / collect of that run): device time of the last run on `slot`, from hipEvents recorded
 * on the slot's stream around the kernels.  ms[0] gate, ms[1] the dominant kernel (the score pass; the single forward
 * kernel under FADEHIP_KERNEL=pk|int32), ms[2] everything after it up to the artifact gates (selection, plan, traced
 * pass 2, traceback, re-run rounds), ms[3] whole run.  With several slots in flight these are durations on a shared
 * device.  counts[0] alignments, counts[1] DP cells, counts[2] trace scratch bytes of the largest pass-2 plan,
 * counts[3] algorithmic bytes of the dominant kernel as SURVEY.md §8(d) defines them (packed query + packed window +
 * 16 B descriptor + 64 B result slot per alignment), counts[4] bytes of wave snapshots the score pass leaves for
 * pass 2 (the wave's loop-carried state every 128 sweep steps; written only when the slot's previous run sent more than
 * 1/32 of its alignments to pass 2 — 0 otherwise), counts[5] candidates traced by pass 2. */
int fadehip_last_run_profile(fadehip_ctx *ctx, int slot, float ms[4], int64_t counts[6]);

/* ------------------------------------------------------ BGZF compression of the output stream -- */
/* What htslib's bgzf_write + zlib do behind SAMWriter(..., SAMWriterTypes.BAM) (util.d:65-76) for the write at
 * anno.d:47-49: the uncompressed BAM byte stream cut into blocks of FADEHIP_BGZF_BLOCK bytes, each compressed into one
 * BGZF member (SAM spec 4.1: gzip header with the BC subfield, one final dynamic-Huffman or stored DEFLATE block,
 * CRC32, ISIZE).  On the device: one workgroup per block with the block in LDS (hash matching, minimum-redundancy
 * codes, bit packing, CRC-32), then the members packed into one contiguous byte stream — what goes to the file, minus
 * the end-of-file marker, which the caller appends once.  Any inflater reads it (zlib, htslib, samtools).
 *   submit : H2D of src[0, n_bytes) (host memory; pinned memory from fadehip_host_alloc copies at PCIe speed), the
 *            kernels; returns at once.  src must stay unchanged until wait returns.
 *   wait   : the members' bytes in a pinned buffer of the lane, valid until the lane's next submit.
 * Two lanes, each with its own stream: one compresses while the other's result is copied back.
 * A call's input is cut into blocks of FADEHIP_BGZF_BLOCK bytes (htslib's block size; the block fills a CU's LDS), or —
 * while the stream is mostly incompressible (the previous call's ratio above 0.45: packed bases, uniform qualities) —
 * of half that (0x7f00): two blocks then share a CU and the rate is 1.5x, for 0.5 % more output.  Either way the
 * stream stays smaller than zlib -6's over 0xff00-byte blocks (tests/test_gpu_bgzf.py).  FADEHIP_BGZF_GEOM=64|32 pins it. */
#define FADEHIP_BGZF_BLOCK 0xff00
#define FADEHIP_BGZF_LANES 2
int fadehip_bgzf_deflate_submit(fadehip_ctx *ctx, int lane, const void *src, size_t n_bytes);
int fadehip_bgzf_deflate_wait(fadehip_ctx *ctx, int lane, const uint8_t **out, size_t *out_bytes);

/* ------------------------------------------------------ BGZF decompression of the input stream -- */
/* What htslib's bgzf_read + zlib's inflate do under dhtslib's SAMReader for `bam.allRecords` (anno.d:44): whole BGZF
 * members in, their payloads out, in order and contiguous.  On the device: one wavefront per member (a DEFLATE stream
 * decodes serially; a file has tens of thousands of members), tables in LDS, CRC32 and ISIZE of every trailer checked.
 * members[0, n_bytes) must hold whole members only (a reader cuts at member boundaries: BSIZE of the BC subfield);
 * out receives at most out_cap bytes, *out_bytes their number.  Synchronous (H2D, kernel, D2H); the streaming file path
 * (fadehip_bam_*) keeps the bytes on the device instead.  A malformed member or a corrupt stream: FADEHIP_E_INVALID
 * with the member's index and the reason in fadehip_last_error. */
int fadehip_bgzf_inflate(fadehip_ctx *ctx, const void *members, size_t n_bytes, void *out, size_t out_cap, size_t *out_bytes);

/* ------------------------------------------------------ the file path on the device: BGZF in, BGZF out -- */
/* anno.d:44-50 as a byte stream: `foreach(rec; bam.allRecords) { annotateTask(rec); out.write(rec); }` with the reader's
 * inflate, the record framing, annotateTask (the level-2 kernels), the tag updates of anno.d:63,94-107 and the writer's
 * deflate all on the device.  The host reads compressed bytes from the input file and writes compressed bytes to the
 * output file; 1.0 byte in and about as much out cross PCIe per compressed byte of the file, nothing else.
 *
 * The caller parses the BAM header itself (it needs the contig names for fadehip_genome_upload anyway), writes the
 * output's header members itself, and then passes the input's members from the one that holds the first record on
 * (first_record = bytes of that member's payload in front of the first record: the BGZF virtual offset's low 16 bits).
 *   front : whole BGZF members of the input (any number; cut where the caller likes — records may span members and
 *           calls).  Inflates and frames the records that are complete (the tail of a record cut by the end of the call is
 *           kept for the next one), waits ONCE for the device — the buffers of the call are sized from what the framing
 *           found — and returns with the annotate kernels and the tag sizes of the call enqueued.  The input buffer may be
 *           reused when front returns.  Two calls are in flight: call k + 1 is framed while call k is annotated.
 *   back  : finishes the oldest front call (reads its sizes, enqueues the kernel that writes the records with their tags),
 *           compresses what it produced and returns the BGZF members in pinned memory — the kernel packs them there itself —,
 *           valid until the back call AFTER the next (a writer thread may still be writing them while the next call
 *           compresses).  If the call after is through its front half too, its compressor is enqueued before this call's
 *           members are waited for.  Outputs come in input order.  Without a front call waiting: FADEHIP_E_STATE (it never
 *           blocks for one).
 * front and back may run on two threads (one each): while back compresses chunk k, front works on chunks k+1 and k+2; front
 * blocks while FADEHIP_BAM_CHUNKS chunks await back (a single-threaded caller alternates front and back).  A record already
 * carrying rs / am / as / ar / ab is updated the way htslib's bam_aux_update_int / bam_aux_update_str do (first occurrence:
 * an integer rs keeps its slot and takes the unsigned type letter of its size; a string tag is replaced at its position; a
 * tag of the wrong kind — rs:Z, am:i — is left as it is, as htslib's EINVAL leaves it).  Errors (corrupt member, impossible
 * record, input ending inside a record when last != 0) fail the call and every later one; what the device finds in a
 * call's records after front has returned surfaces at the call's back.  fadehip_bam_totals counts the calls back has taken.
 * The genome must have been uploaded (fadehip_genome_upload) by the first front call; the stream uses the ctx's slots 0 and
 * 1 and both BGZF lanes.
 * fadehip_bam_prepare (optional, before the first front call, typically on a thread of its own beside the caller's start-up):
 * makes what the first calls would otherwise make one after the other — the streams of the second slot and of the
 * compressor, the staging buffers of the members, the buffers of call_bytes inflated bytes — side by side, and pays for
 * the first copy and the first wait of every stream.  The genome may go up meanwhile (fadehip_genome_upload on another
 * thread).  Results do not depend on it. */
typedef struct fadehip_bam_stream fadehip_bam_stream;
typedef struct fadehip_bam_config {
    int32_t floor_len;            /* --min-length (anno.d: artifact_floor_length) */
    int32_t window;               /* -w (align_buffer_size) */
    int32_t n_ref;                /* contigs of the BAM header: refID is checked against it, ref_names[refID] goes into am */
    int32_t flags;                /* FADEHIP_BAM_STORED: uncompressed BGZF out (`fade annotate -u`, htslib's level 0);
                                   * FADEHIP_BAM_NO_OUTPUT: back waits for the call's annotated records (on the device) and gives
                                   * their buffer back without making BGZF of them, *out_bytes = 0 — the rate of the record path
                                   * alone, from BAM record bytes to annotated record bytes (bench.py's value_from_records) */
    const char *const *ref_names; /* [n_ref] NUL-terminated */
    uint32_t first_record;        /* payload bytes of the first member passed to front that precede the first record */
    uint32_t tail_trim;           /* payload bytes at the END of the last member (front's last call) that are not this stream's:
                                   * a reader of a range of the file stops where the next range's first record starts */
} fadehip_bam_config;
#define FADEHIP_BAM_CHUNKS 3
#define FADEHIP_BAM_STORED 1
#define FADEHIP_BAM_NO_OUTPUT 2
int fadehip_bam_open(fadehip_ctx *ctx, const fadehip_bam_config *cfg, fadehip_bam_stream **out);
int fadehip_bam_prepare(fadehip_bam_stream *st, size_t call_bytes);
int fadehip_bam_front(fadehip_bam_stream *st, const void *members, size_t n_bytes, int last);
/* front for a caller that inflates itself (host cores otherwise idle; the device then spends its time on the rest):
 * payload = the members' inflated bytes, any cut, pinned memory for PCIe speed.  Calls of both kinds may alternate. */
int fadehip_bam_front_raw(fadehip_bam_stream *st, const void *payload, size_t n_bytes, int last);
/* (*out is good during the next back call and no longer: write it, or have it written, before the call after next) */
int fadehip_bam_back(fadehip_bam_stream *st, const uint8_t **out, size_t *out_bytes);
/* totals so far: the eight Stats.parse counters (stats.d:45-54), records, reads beyond the kernels' limits */
int fadehip_bam_totals(fadehip_bam_stream *st, int64_t stats[8], int64_t *n_records, int64_t *n_oversize);
void fadehip_bam_close(fadehip_bam_stream *st);

/* Sum counters over the ranks' devices with one ncclAllReduce (RCCL) — single process, one ctx
 * per device.  counters is [n_ctx][count] in, every row holds the sum on return. */
int fadehip_stats_allreduce(fadehip_ctx *const *ctxs, int n_ctx, int64_t *counters, int count);

/* The same sum with one PROCESS per GPU (the lanes of `fade annotate --gpus N`, bench.py's ranks use torch.distributed for
 * it): rank 0 makes the ncclUniqueId and leaves it in the file id_path (written under another name, then renamed), the
 * other ranks wait for the file (up to a minute); ncclCommInitRank, one ncclAllReduce(int64, sum), the communicator is
 * destroyed again.  counters[count] in, the sums out. */
int fadehip_stats_allreduce_rank(fadehip_ctx *ctx, int rank, int n_ranks, const char *id_path, int64_t *counters, int count);

#ifdef __cplusplus
}
#endif
#endif
