/*
 * fade_logic.c — restatement over plain structs of FADE's per-read annotate logic.
 * TEST INFRASTRUCTURE ONLY (see fade_oracle.h).  PARITY UNPINNED.
 *
 * Follows, line by line:
 *   source/anno.d:55-110      annotateTask
 *   source/analysis.d:22-124  align_clip!(left)
 *   source/util.d:18-62       seq_comp_table, reverse_complement_sam_record, parse_clips
 *   source/readstatus.d:5-26  ReadStatus bit layout
 *   source/stats.d:45-54      Stats.parse
 * As in the reference, a read with two qualifying clips runs the same alignment twice
 * (anno.d:79-91); the oracle does not de-duplicate.
 */
#include "fade_oracle.h"
#include <ctype.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* util.d:18-20 */
const uint8_t fo_seq_comp_table[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
/* htslib seq_nt16_str */
static const char nt16_str[] = "=ACMGRSVTWYHKDBN";

/* util.d:23-34 */
void fo_reverse_complement_packed(const uint8_t *seq4, int l_seq, char *out) {
    int j = l_seq - 1;
    for (int i = 0; i < l_seq; i++) {
        const int code = (seq4[i >> 1] >> ((~i & 1) << 2)) & 0xf;
        out[j--] = nt16_str[fo_seq_comp_table[code]];
    }
}

/* util.d:37-62 */
void fo_parse_clips(const uint32_t *cigar, int n_cigar, uint32_t clips[2]) {
    clips[0] = clips[1] = 0;
    int first = 1;
    for (int k = 0; k < n_cigar; k++) {
        const int op = cigar[k] & 0xf;
        if (op == 5) continue; /* skip hard clips, util.d:44-45 */
        const int is_sc = (op == 4);
        if (first && !is_sc) first = 0;
        else if (first && is_sc) clips[0] = cigar[k];
        else if (is_sc) clips[1] = cigar[k];
    }
}

int64_t fo_cigar_aligned_length(const uint32_t *cigar, int n_cigar) {
    int64_t n = 0;
    for (int k = 0; k < n_cigar; k++) {
        const int op = cigar[k] & 0xf;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) n += cigar[k] >> 4;
    }
    return n;
}

int fo_cigar_to_string(const uint32_t *cigar, int n_cigar, char *buf, int cap) {
    static const char opc[] = "MIDNSHP=XB??????";
    int n = 0;
    for (int k = 0; k < n_cigar; k++) {
        char tmp[16];
        int m = snprintf(tmp, sizeof tmp, "%u%c", cigar[k] >> 4, opc[cigar[k] & 0xf]);
        for (int t = 0; t < m; t++) {
            if (n + 1 < cap) buf[n] = tmp[t];
            n++;
        }
    }
    if (cap > 0) buf[n < cap ? n : cap - 1] = 0;
    return n;
}

typedef struct {
    char *alignment, *bq, *stem_loop, *stem_loop_rc; /* analysis.d:12-19; NULL == "" */
} align_result;

static char *dup_range(const char *s, int64_t from, int64_t to) {
    int64_t n = to - from;
    if (n < 0) n = 0;
    char *o = (char *)malloc((size_t)n + 1);
    memcpy(o, s + from, (size_t)n);
    o[n] = 0;
    return o;
}

/* analysis.d:22-124 */
static void align_clip(int left, const fo_params *p, const fo_genome *g, const fo_read *rec,
                       uint8_t *status, uint32_t clip_len, int floor_len, int window,
                       align_result *al, int *n_sw) {
    memset(al, 0, sizeof *al);
    /* analysis.d:34-37 */
    if ((int64_t)clip_len <= (int64_t)floor_len) return;
    /* analysis.d:40 */
    const int lq = rec->l_seq;
    char *q_seq = (char *)malloc((size_t)lq + 1);
    fo_reverse_complement_packed(rec->seq4, lq, q_seq);
    q_seq[lq] = 0;
    /* analysis.d:43 (float cutoff = clip_len * 0.9 * 2) */
    const float cutoff = (float)((double)clip_len * 0.9 * 2);
    /* analysis.d:45-51 */
    int64_t start = rec->pos - window;
    if (start < 0) start = 0;
    /* analysis.d:53-59 */
    const int64_t aligned = fo_cigar_aligned_length(rec->cigar, rec->n_cigar);
    int64_t end = rec->pos + aligned + window;
    if (end > g->lengths[rec->tid]) end = g->lengths[rec->tid];
    /* analysis.d:61-64 */
    int64_t lr = end - start;
    if (lr < 0) lr = 0;
    char *ref_seq = (char *)malloc((size_t)lr + 1);
    for (int64_t k = 0; k < lr; k++) ref_seq[k] = (char)toupper((unsigned char)g->seqs[rec->tid][start + k]);
    ref_seq[lr] = 0;
    /* analysis.d:67 */
    fo_sw_result res;
    const int cap = lq + (int)lr + 4;
    uint32_t *ops = (uint32_t *)malloc((size_t)cap * sizeof(uint32_t));
    if (p->striped) fo_sw_striped(p, q_seq, lq, ref_seq, (int)lr, &res, ops, cap);
    else fo_sw_trace(p, q_seq, lq, ref_seq, (int)lr, &res, ops, cap);
    (*n_sw)++;
    /* analysis.d:69-70 */
    if (res.n_ops == 0 || res.n_ops > 10) goto done;
    if (left) {
        /* analysis.d:74 */
        if ((ops[res.n_ops - 1] & 0xf) == 7) {
            /* analysis.d:76 */
            if ((float)res.score > cutoff) {
                /* analysis.d:78-80 */
                uint32_t clips[2];
                fo_parse_clips(ops, res.n_ops, clips);
                if ((clips[1] >> 4) != 0 || (clips[0] >> 4) == 0) goto done;
                /* analysis.d:82-83 */
                *status |= 1u << 1;           /* art_left = true */
                *status &= (uint8_t)~(1u << 3); /* mate_left = false */
                /* analysis.d:84-85 */
                char cig[256];
                fo_cigar_to_string(ops, res.n_ops, cig, sizeof cig);
                const size_t nlen = strlen(g->names[rec->tid]) + strlen(cig) + 32;
                al->alignment = (char *)malloc(nlen);
                snprintf(al->alignment, nlen, "%s,%lld,%s", g->names[rec->tid],
                         (long long)(start + res.beg_ref), cig);
                /* analysis.d:86-89 */
                const int64_t apos = start + res.beg_ref;
                const int64_t overlap = apos >= rec->pos - (int64_t)clip_len ? apos - (rec->pos - (int64_t)clip_len) : 0;
                int64_t plen = ((int64_t)lq - (int64_t)(clips[0] >> 4)) + overlap;
                plen = plen > lq ? lq : plen;
                /* analysis.d:90-92 */
                char *seq = (char *)malloc((size_t)lq + 1), *bq = (char *)malloc((size_t)lq + 1);
                for (int k = 0; k < lq; k++) {
                    seq[k] = nt16_str[(rec->seq4[k >> 1] >> ((~k & 1) << 2)) & 0xf];
                    bq[k] = (char)(rec->qual[k] + 33);
                }
                al->stem_loop = dup_range(seq, 0, plen);
                al->stem_loop_rc = dup_range(q_seq, lq - plen, lq);
                al->bq = dup_range(bq, 0, plen);
                free(seq); free(bq);
            }
        }
    } else {
        /* analysis.d:98 */
        if ((ops[0] & 0xf) == 7) {
            /* analysis.d:100 */
            if ((float)res.score > cutoff) {
                /* analysis.d:102-104 */
                uint32_t clips[2];
                fo_parse_clips(ops, res.n_ops, clips);
                if ((clips[0] >> 4) != 0 || (clips[1] >> 4) == 0) goto done;
                /* analysis.d:106-107 */
                *status |= 1u << 2;             /* art_right = true */
                *status &= (uint8_t)~(1u << 4); /* mate_right = false */
                /* analysis.d:108-109 */
                char cig[256];
                fo_cigar_to_string(ops, res.n_ops, cig, sizeof cig);
                const size_t nlen = strlen(g->names[rec->tid]) + strlen(cig) + 32;
                al->alignment = (char *)malloc(nlen);
                snprintf(al->alignment, nlen, "%s,%lld,%s", g->names[rec->tid],
                         (long long)(start + res.beg_ref), cig);
                /* analysis.d:110-115 */
                const int64_t res_aligned = fo_cigar_aligned_length(ops, res.n_ops);
                const int64_t lhs = rec->pos + aligned + (int64_t)clip_len;
                const int64_t rhs = start + res.beg_ref + res_aligned;
                const int64_t overlap = lhs >= rhs ? lhs - rhs : 0;
                int64_t plen = ((int64_t)lq - (int64_t)(clips[1] >> 4)) + overlap;
                plen = plen > lq ? lq : plen;
                /* analysis.d:116-118 */
                char *seq = (char *)malloc((size_t)lq + 1), *bq = (char *)malloc((size_t)lq + 1);
                for (int k = 0; k < lq; k++) {
                    seq[k] = nt16_str[(rec->seq4[k >> 1] >> ((~k & 1) << 2)) & 0xf];
                    bq[k] = (char)(rec->qual[k] + 33);
                }
                al->stem_loop = dup_range(seq, lq - plen, lq);
                al->stem_loop_rc = dup_range(q_seq, 0, plen);
                al->bq = dup_range(bq, lq - plen, lq);
                free(seq); free(bq);
            }
        }
    }
done:
    free(ops);
    free(ref_seq);
    free(q_seq);
}

static char *join2(const char *a, const char *b) {
    const size_t la = a ? strlen(a) : 0, lb = b ? strlen(b) : 0;
    char *o = (char *)malloc(la + lb + 2);
    if (la) memcpy(o, a, la);
    o[la] = ';';
    if (lb) memcpy(o + la + 1, b, lb);
    o[la + 1 + lb] = 0;
    return o;
}

static void free_align(align_result *a) {
    free(a->alignment); free(a->bq); free(a->stem_loop); free(a->stem_loop_rc);
}

/* anno.d:55-110 */
int fo_annotate_task(const fo_params *p, const fo_genome *g, const fo_read *rec, int floor_len,
                     int window, fo_anno *out) {
    memset(out, 0, sizeof *out);
    uint8_t status = 0;
    /* anno.d:61-65 */
    int n_soft = 0;
    for (int k = 0; k < rec->n_cigar; k++)
        if ((rec->cigar[k] & 0xf) == 4) n_soft++;
    if ((rec->flag & 0x4) || n_soft == 0) {
        out->rs = status;
        return 0;
    }
    /* anno.d:68-70 */
    uint32_t clips[2];
    fo_parse_clips(rec->cigar, rec->n_cigar, clips);
    if ((clips[0] >> 4) != 0 || (clips[1] >> 4) != 0) status |= 1u << 0;
    /* anno.d:73-74 */
    if (rec->has_sa) status |= 1u << 5;
    /* anno.d:78-91 */
    align_result a1, a2;
    memset(&a1, 0, sizeof a1);
    memset(&a2, 0, sizeof a2);
    if ((clips[0] >> 4) != 0)
        align_clip(1, p, g, rec, &status, clips[0] >> 4, floor_len, window, &a1, &out->n_sw_calls);
    if ((clips[1] >> 4) != 0)
        align_clip(0, p, g, rec, &status, clips[1] >> 4, floor_len, window, &a2, &out->n_sw_calls);
    /* anno.d:94 */
    out->rs = status;
    /* anno.d:98-107 */
    if (status & ((1u << 1) | (1u << 2))) {
        out->has_tags = 1;
        out->am = join2(a1.alignment, a2.alignment);
        out->as_ = join2(a1.stem_loop, a2.stem_loop);
        out->ar = join2(a1.stem_loop_rc, a2.stem_loop_rc);
        out->ab = join2(a1.bq, a2.bq);
    }
    free_align(&a1);
    free_align(&a2);
    return 0;
}

void fo_anno_free(fo_anno *a) {
    free(a->am); free(a->as_); free(a->ar); free(a->ab);
    memset(a, 0, sizeof *a);
}

/* stats.d:45-54 (read_count is bumped by the caller in filter.d; here per parse) */
void fo_stats_parse(uint8_t rs, int64_t c[8]) {
    const int sc = rs & 1, al = (rs >> 1) & 1, ar = (rs >> 2) & 1, ml = (rs >> 3) & 1,
              mr = (rs >> 4) & 1, sup = (rs >> 5) & 1;
    c[0] += 1;
    c[1] += sc;
    c[2] += sup;
    c[3] += (al | ar) & sup;
    c[4] += (al | ar);
    c[5] += ((al & ml) | (ar & mr));
    c[6] += al;
    c[7] += ar;
}

typedef struct {
    const fo_params *p; const fo_genome *g; const fo_read *reads; int n, floor_len, window;
    int tid, nthreads; uint8_t *rs_out; char **am_out;
} batch_arg;

static void *batch_worker(void *v) {
    batch_arg *a = (batch_arg *)v;
    /* contiguous ranges per thread */
    const int64_t lo = (int64_t)a->n * a->tid / a->nthreads, hi = (int64_t)a->n * (a->tid + 1) / a->nthreads;
    for (int64_t i = lo; i < hi; i++) {
        fo_anno an;
        fo_annotate_task(a->p, a->g, &a->reads[i], a->floor_len, a->window, &an);
        a->rs_out[i] = an.rs;
        if (a->am_out) {
            a->am_out[i] = an.am;
            an.am = NULL;
        }
        fo_anno_free(&an);
    }
    return NULL;
}

int fo_annotate_batch(const fo_params *p, const fo_genome *g, const fo_read *reads, int n,
                      int floor_len, int window, int threads, uint8_t *rs_out, char **am_out) {
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    batch_arg *args = (batch_arg *)malloc(sizeof(batch_arg) * (size_t)threads);
    for (int t = 0; t < threads; t++) {
        args[t] = (batch_arg){p, g, reads, n, floor_len, window, t, threads, rs_out, am_out};
        if (t > 0) pthread_create(&th[t], NULL, batch_worker, &args[t]);
    }
    batch_worker(&args[0]);
    for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    free(args);
    return 0;
}

/* Same over the BAM-native structure-of-arrays batch layout of include/fadehip.h (plus quals). */
int fo_annotate_batch_soa(const fo_params *p, const fo_genome *g, int n, const int32_t *tid, const int32_t *pos,
                          const uint16_t *flag, const uint8_t *has_sa, const int32_t *l_seq,
                          const uint32_t *cigar_off, const uint32_t *cigar_ops, const uint32_t *seq_off,
                          const uint8_t *seq_packed, const int64_t *qual_off, const uint8_t *qual,
                          int floor_len, int window, int threads, uint8_t *rs_out, char **am_out) {
    fo_read *reads = (fo_read *)calloc((size_t)(n > 0 ? n : 1), sizeof(fo_read));
    if (!reads) return -1;
    for (int i = 0; i < n; i++) {
        reads[i].qname = "r";
        reads[i].flag = flag[i];
        reads[i].tid = tid[i];
        reads[i].pos = pos[i];
        reads[i].n_cigar = (int)(cigar_off[i + 1] - cigar_off[i]);
        reads[i].cigar = cigar_ops + cigar_off[i];
        reads[i].l_seq = l_seq[i];
        reads[i].seq4 = seq_packed + seq_off[i];
        reads[i].qual = qual + qual_off[i];
        reads[i].has_sa = has_sa[i];
    }
    int rc = fo_annotate_batch(p, g, reads, n, floor_len, window, threads, rs_out, am_out);
    free(reads);
    return rc;
}
