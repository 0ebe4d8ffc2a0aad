/*
 * sw_scalar.c — scalar full-matrix restatement of the alignment FADE obtains from
 * `p.sw_striped(q_seq, ref_seq)` (reference call site source/analysis.d:67, parameters
 * source/anno.d:36).  TEST INFRASTRUCTURE ONLY (see fade_oracle.h).  PARITY UNPINNED.
 *
 * The arithmetic itself is third-party and absent from /root/reference: dparasail ~>0.3.3
 * (dub.json:9) wrapping libparasail 2.4.3 (Dockerfile:4).  What follows restates parasail's
 * published algorithm as recorded in SURVEY.md Appendix A:
 *   A.1 matrix_create(alphabet, match, mismatch): |alphabet|+1 symbols, last = wildcard scoring 0
 *   A.2 affine recurrence, first gap base costs `open`, further ones `ext`
 *   A.3 end cell: max H, smallest ref index, then smallest query index
 *   A.4 trace bits and traceback state machine
 *   A.5 CIGAR letters, A.6 dparasail soft-clip padding
 */
#include "fade_oracle.h"
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

void fo_params_default(fo_params *p) {
    p->open = 10;
    p->ext = 2;
    p->match = 2;
    p->mismatch = -3;
    p->alphabet = "ACTGN";
    p->rules = FO_RULES_DEFAULT;
    p->striped = 0;
}

/* A.1: 256-entry case-insensitive mapper; unknown characters map to the wildcard index. */
static void build_mapper(const fo_params *p, int mapper[256], int *size) {
    int n = (int)strlen(p->alphabet);
    for (int c = 0; c < 256; c++) mapper[c] = n;
    for (int k = 0; k < n; k++) {
        mapper[toupper((unsigned char)p->alphabet[k])] = k;
        mapper[tolower((unsigned char)p->alphabet[k])] = k;
    }
    *size = n + 1;
}

static inline int score_of(const fo_params *p, int n, int a, int b) {
    if (a == n || b == n) return 0; /* wildcard row/column */
    if (a == b) {
        if (!(p->rules & FO_RULE_N_MATCHES_N) && toupper((unsigned char)p->alphabet[a]) == 'N')
            return p->mismatch;
        return p->match;
    }
    return p->mismatch;
}

#define NEG_INF (INT_MIN / 4)

enum { T_ZERO = 0, T_DIAG = 1, T_F = 2, T_E = 3, T_EOPEN = 4, T_FOPEN = 8 };
/* BAM op codes "MIDNSHP=X" */
enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };

int fo_sw_trace_table(const fo_params *p, const char *q, int lq, const char *r, int lr,
                      fo_sw_result *res, uint32_t *ops, int ops_cap, uint8_t *trace_out) {
    int mapper[256], msize;
    build_mapper(p, mapper, &msize);
    const int nalpha = msize - 1;
    const int open = p->open, ext = p->ext;
    memset(res, 0, sizeof(*res));
    if (lq <= 0 || lr <= 0) {
        res->end_query = res->end_ref = -1;
        res->n_ops = 0;
        if (lq > 0 && (p->rules & FO_RULE_PAD_SOFTCLIP)) {
            /* nothing aligned: the whole query is unaligned */
            res->n_ops = 1;
            if (ops_cap > 0) ops[0] = ((uint32_t)lq << 4) | OP_S;
        }
        return 0;
    }
    uint8_t *trace = trace_out ? trace_out : (uint8_t *)malloc((size_t)lq * (size_t)lr);
    int *Hrow = (int *)calloc((size_t)lr + 1, sizeof(int)); /* H[i-1][*], index j+1 */
    int *Frow = (int *)malloc(((size_t)lr + 1) * sizeof(int)); /* F[i-1][*] */
    int *rmap = (int *)malloc((size_t)lr * sizeof(int));
    if (!trace || !Hrow || !Frow || !rmap) {
        if (!trace_out) free(trace);
        free(Hrow); free(Frow); free(rmap);
        return -1;
    }
    for (int j = 0; j <= lr; j++) Frow[j] = NEG_INF;
    for (int j = 0; j < lr; j++) rmap[j] = mapper[(unsigned char)r[j]];

    int score = -1, end_q = 0, end_r = 0;
    for (int i = 0; i < lq; i++) {
        const int qa = mapper[(unsigned char)q[i]];
        int NWH = 0;       /* H[i-1][-1] = 0 */
        int WH = 0;        /* H[i][-1] = 0 */
        int E = NEG_INF;   /* E[i][-1] */
        uint8_t *trow = trace + (size_t)i * lr;
        for (int j = 0; j < lr; j++) {
            const int NH = Hrow[j + 1]; /* H[i-1][j] */
            /* A.2 */
            const int F_opn = NH - open, F_ext = Frow[j + 1] - ext;
            const int F = F_opn > F_ext ? F_opn : F_ext;
            const int E_opn = WH - open, E_ext = E - ext;
            E = E_opn > E_ext ? E_opn : E_ext;
            const int D = NWH + score_of(p, nalpha, qa, rmap[j]);
            int H = D;
            if (E > H) H = E;
            if (F > H) H = F;
            if (H < 0) H = 0;
            /* A.4 trace bits */
            uint8_t t;
            if (H == 0) t = T_ZERO;
            else if (p->rules & FO_RULE_HDIR_DIAG_F_E) t = (H == D) ? T_DIAG : (H == F) ? T_F : T_E;
            else t = (H == D) ? T_DIAG : (H == E) ? T_E : T_F;
            if (p->rules & FO_RULE_GAP_TIE_EXTENDS) {
                if (E_opn > E_ext) t |= T_EOPEN;
                if (F_opn > F_ext) t |= T_FOPEN;
            } else {
                if (E_opn >= E_ext) t |= T_EOPEN;
                if (F_opn >= F_ext) t |= T_FOPEN;
            }
            trow[j] = t;
            /* A.3 end cell */
            if (H > score) {
                score = H; end_q = i; end_r = j;
            } else if (H == score) {
                if (p->rules & FO_RULE_END_MIN_REF_THEN_QUERY) {
                    if (j < end_r) { end_q = i; end_r = j; }
                }
            }
            NWH = NH;
            WH = H;
            Hrow[j + 1] = H; /* becomes H[i-1][j] for the next row */
            Frow[j + 1] = F;
        }
    }

    /* A.4 traceback */
    size_t cap = (size_t)lq + (size_t)lr + 4;
    uint32_t *rev = (uint32_t *)malloc(cap * sizeof(uint32_t));
    size_t nrev = 0;
    const int ref_only_op = (p->rules & FO_RULE_SAM_GAP_LETTERS) ? OP_D : OP_I;
    const int qry_only_op = (p->rules & FO_RULE_SAM_GAP_LETTERS) ? OP_I : OP_D;
    int i = end_q, j = end_r, state = 0; /* 0 H, 1 E, 2 F */
    int cur_op = -1;
    uint32_t cur_len = 0;
#define EMIT(o) do { if ((o) == cur_op) cur_len++; else { if (cur_op >= 0) rev[nrev++] = (cur_len << 4) | (uint32_t)cur_op; cur_op = (o); cur_len = 1; } } while (0)
    while (i >= 0 && j >= 0) {
        const uint8_t t = trace[(size_t)i * lr + j];
        if (state == 0) {
            const int d = t & 3;
            if (d == T_ZERO) break;
            if (d == T_DIAG) {
                int eq;
                if (p->rules & FO_RULE_EQ_BY_CHAR) eq = (q[i] == r[j]);
                else eq = score_of(p, nalpha, mapper[(unsigned char)q[i]], mapper[(unsigned char)r[j]]) > 0;
                EMIT(eq ? OP_EQ : OP_X);
                i--; j--;
            } else if (d == T_E) state = 1;
            else state = 2;
        } else if (state == 1) {
            EMIT(ref_only_op);
            j--;
            state = (t & T_EOPEN) ? 0 : 1;
        } else {
            EMIT(qry_only_op);
            i--;
            state = (t & T_FOPEN) ? 0 : 2;
        }
    }
    if (cur_op >= 0) rev[nrev++] = (cur_len << 4) | (uint32_t)cur_op;
#undef EMIT
    res->score = score < 0 ? 0 : score;
    res->end_query = end_q;
    res->end_ref = end_r;
    res->beg_query = i + 1;
    res->beg_ref = j + 1;

    /* A.6 assemble: [beg_query S] + ops + [(lq-1-end_query) S] */
    int n = 0;
    if (p->rules & FO_RULE_PAD_SOFTCLIP) {
        if (res->beg_query > 0) {
            if (n < ops_cap) ops[n] = ((uint32_t)res->beg_query << 4) | OP_S;
            n++;
        }
    }
    for (size_t k = nrev; k-- > 0;) {
        if (n < ops_cap) ops[n] = rev[k];
        n++;
    }
    if (p->rules & FO_RULE_PAD_SOFTCLIP) {
        const int tail = lq - 1 - end_q;
        if (tail > 0) {
            if (n < ops_cap) ops[n] = ((uint32_t)tail << 4) | OP_S;
            n++;
        }
    }
    res->n_ops = n;
    free(rev);
    if (!trace_out) free(trace);
    free(Hrow); free(Frow); free(rmap);
    return 0;
}

int fo_sw_trace(const fo_params *p, const char *q, int lq, const char *r, int lr,
                fo_sw_result *res, uint32_t *ops, int ops_cap) {
    return fo_sw_trace_table(p, q, lq, r, lr, res, ops, ops_cap, NULL);
}
