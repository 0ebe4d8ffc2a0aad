/*
 * fade_oracle.h — CPU restatement of FADE's `annotate` hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity oracle for the MI355X path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; nothing under fade_amd/ links or calls it.
 *
 * PARITY UNPINNED: the reference (blachlylab/fade, D) has no tests, fixtures or golden vectors,
 * and its arithmetic lives in un-vendored libraries absent from this environment
 * (dparasail ~>0.3.3 -> libparasail 2.4.3; dhtslib@c51b842 -> htslib 1.13).  The SW / trace /
 * CIGAR rules below restate parasail's published algorithm; every rule
 * that could not be checked against source is a named FO_RULE_* switch so it can be flipped in
 * one place.  FADE's own control flow (source/anno.d, source/analysis.d, source/util.d,
 * source/readstatus.d) is restated line by line with file:line citations.
 */
#ifndef FADE_ORACLE_H
#define FADE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- named behavioural rules of the un-vendored arithmetic ---- */
enum {
    /* A.3: end cell = max H; ties -> smallest ref index, then smallest query index. */
    FO_RULE_END_MIN_REF_THEN_QUERY = 1u << 0,
    /* A.4: H direction priority ZERO > DIAG > F (query-only) > E (ref-only). */
    FO_RULE_HDIR_DIAG_F_E = 1u << 1,
    /* A.4: gap origin "opened" only on strict >, ties extend. */
    FO_RULE_GAP_TIE_EXTENDS = 1u << 2,
    /* A.4: '=' vs 'X' decided by character equality, not by matrix sign. */
    FO_RULE_EQ_BY_CHAR = 1u << 3,
    /* A.5: ref-only step -> 'D', query-only step -> 'I' (SAM-compliant letters). */
    FO_RULE_SAM_GAP_LETTERS = 1u << 4,
    /* A.6: dparasail pads the CIGAR with leading/trailing S for unaligned query ends. */
    FO_RULE_PAD_SOFTCLIP = 1u << 5,
    /* A.1: alphabet letter vs itself scores `match` even for N. */
    FO_RULE_N_MATCHES_N = 1u << 6,
    FO_RULES_DEFAULT = 0x7f
};

typedef struct {
    int open;      /* first gap base costs `open` (anno.d:36 -> 10) */
    int ext;       /* each further gap base costs `ext` (2) */
    int match;     /* +2 */
    int mismatch;  /* -3 */
    const char *alphabet; /* "ACTGN" (anno.d:36) */
    uint32_t rules; /* FO_RULES_DEFAULT */
    int striped;    /* 1: fo_annotate_* call the striped AVX2 restatement instead of the scalar one */
} fo_params;

void fo_params_default(fo_params *p); /* Parasail("ACTGN", 10, 2, 2, -3), anno.d:36 */

/* Result of one `p.sw_striped(q, r)` call (analysis.d:67) as FADE reads it:
 * res.score, res.position (= beg_ref), res.cigar (BAM-encoded ops, len<<4|op, "MIDNSHP=X"). */
typedef struct {
    int score;
    int end_query, end_ref; /* 0-based, inclusive */
    int beg_query, beg_ref; /* 0-based */
    int n_ops;              /* total op count incl. S pads; may exceed the caller's capacity */
} fo_sw_result;

/* Scalar full-matrix affine-gap local alignment with per-cell trace and traceback.
 * q/r are ASCII.  ops_cap entries of `ops` are filled (front of the CIGAR); n_ops is the true count.
 * Returns 0, or -1 on allocation failure. */
int fo_sw_trace(const fo_params *p, const char *q, int lq, const char *r, int lr,
                fo_sw_result *res, uint32_t *ops, int ops_cap);

/* Same, additionally exporting the per-cell trace table (lq*lr bytes, row-major [i][j]):
 * bits 0-1 H dir (0 zero, 1 diag, 2 F/query-only, 3 E/ref-only), bit 2 E opened, bit 3 F opened. */
int fo_sw_trace_table(const fo_params *p, const char *q, int lq, const char *r, int lr,
                      fo_sw_result *res, uint32_t *ops, int ops_cap, uint8_t *trace);

/* Striped (Farrar) 16-bit AVX2 restatement of the same alignment; identical results by contract
 * (tests/test_oracle_striped.py).  Falls back to the scalar code for non-default rules or no AVX2. */
int fo_sw_striped(const fo_params *p, const char *q, int lq, const char *r, int lr,
                  fo_sw_result *res, uint32_t *ops, int ops_cap);

/* n alignments over concatenated ASCII buffers; q_off/r_off have n+1 entries.
 * res: n x 6 int32 (score,end_q,end_r,beg_q,beg_r,n_ops); ops: n x max_ops uint32 (front of CIGAR).
 * `variant` 0 = scalar, 1 = striped AVX2.  `threads` pthreads over contiguous ranges. */
int fo_sw_batch(const fo_params *p, int n, int threads, const uint8_t *q, const int64_t *q_off,
                const uint8_t *r, const int64_t *r_off, int32_t *res, uint32_t *ops, int max_ops,
                int variant);

/* ---- util.d restatements ---- */
/* util.d:18-20 */
extern const uint8_t fo_seq_comp_table[16];
/* util.d:23-34: BAM 4-bit packed seq -> reverse-complemented ASCII (out has l_seq bytes, no NUL). */
void fo_reverse_complement_packed(const uint8_t *seq4, int l_seq, char *out);
/* util.d:37-62: returns leftS op in clips[0], rightS op in clips[1] (BAM-encoded, 0 if none). */
void fo_parse_clips(const uint32_t *cigar, int n_cigar, uint32_t clips[2]);
/* dhtslib Cigar.alignedLength: sum of M,D,N,=,X lengths. */
int64_t fo_cigar_aligned_length(const uint32_t *cigar, int n_cigar);
/* dhtslib Cigar.toString into buf (cap bytes incl. NUL); returns strlen needed. */
int fo_cigar_to_string(const uint32_t *cigar, int n_cigar, char *buf, int cap);

/* ---- plain-struct mirror of what annotateTask touches (anno.d:55-110) ---- */
typedef struct {
    int n_contigs;
    const char *const *names;
    const int64_t *lengths;
    const char *const *seqs; /* raw FASTA residues, any case (analysis.d:63 upper-cases) */
} fo_genome;

typedef struct {
    const char *qname;
    uint16_t flag;
    int32_t tid;
    int64_t pos; /* 0-based */
    int n_cigar;
    const uint32_t *cigar;
    int l_seq;
    const uint8_t *seq4; /* BAM packed */
    const uint8_t *qual; /* raw phred, l_seq bytes */
    int has_sa;          /* rec["SA"].exists, anno.d:73 */
} fo_read;

typedef struct {
    uint8_t rs;  /* readstatus.d:5-26 raw */
    int has_tags; /* 1 iff art_left|art_right -> am/as/ar/ab set (anno.d:98-107) */
    char *am, *as_, *ar, *ab; /* malloc'd NUL-terminated; free with fo_anno_free */
    /* diagnostics (not part of the reference output): number of SW calls made (0..2). */
    int n_sw_calls;
} fo_anno;

/* anno.d:55-110 + analysis.d:22-124.  floor = --min-length (app.d:17), window = -w (app.d:18). */
int fo_annotate_task(const fo_params *p, const fo_genome *g, const fo_read *rd, int floor_len,
                     int window, fo_anno *out);
void fo_anno_free(fo_anno *a);

/* Batch helper for timing / tests: annotate n reads with `threads` pthreads (>=1).
 * rs_out[n]; am_out may be NULL, else receives malloc'd strings (NULL when no tags). */
int fo_annotate_batch(const fo_params *p, const fo_genome *g, const fo_read *reads, int n,
                      int floor_len, int window, int threads, uint8_t *rs_out, char **am_out);

/* Same over the BAM-native structure-of-arrays batch layout of include/fadehip.h (plus quals). */
int fo_annotate_batch_soa(const fo_params *p, const fo_genome *g, int n, const int32_t *tid, const int32_t *pos,
                          const uint16_t *flag, const uint8_t *has_sa, const int32_t *l_seq,
                          const uint32_t *cigar_off, const uint32_t *cigar_ops, const uint32_t *seq_off,
                          const uint8_t *seq_packed, const int64_t *qual_off, const uint8_t *qual,
                          int floor_len, int window, int threads, uint8_t *rs_out, char **am_out);

/* stats.d:45-54 Stats.parse over an rs byte; counters[8] = read_count, clipped, sup, art_sup,
 * art, art_mate, aln_l, aln_r. */
void fo_stats_parse(uint8_t rs, int64_t counters[8]);

#ifdef __cplusplus
}
#endif
#endif
