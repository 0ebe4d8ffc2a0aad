/*
 * sw_striped.c — striped (Farrar) 16-bit AVX2 restatement of the alignment behind
 * `p.sw_striped(q_seq, ref_seq)` (source/analysis.d:67; libparasail's sw_trace_striped_16, un-vendored),
 * plus the batch drivers.  TEST INFRASTRUCTURE ONLY (see fade_oracle.h).  PARITY UNPINNED.
 *
 * Purpose: (1) a third, structurally different implementation of SURVEY.md Appendix A that must agree
 * bit for bit with the scalar oracle and the HIP kernels; (2) the timed CPU baseline of bench.py, since
 * the reference runs exactly this kind of SIMD kernel on the host.
 *
 * Layout: query rows are striped over 16 int16 lanes (row = seg + lane * segLen).  Per reference
 * column: a main sweep with the running F, the lazy-F loop (continued on F_ext >= F_opn so that the
 * "ties extend" rule of Appendix A.4 is kept), then a sweep that derives the four trace bits of every
 * cell from the final H / F / diagonal values.  The trace table is stored in striped order.
 */
#include "fade_oracle.h"
#include <ctype.h>
#include <immintrin.h>
#include <limits.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define SEGW 16
enum { T_ZERO = 0, T_DIAG = 1, T_F = 2, T_E = 3, T_EOPEN = 4, T_FOPEN = 8 };
enum { OP_I = 1, OP_D = 2, OP_S = 4, OP_EQ = 7, OP_X = 8 };
#define NEG16 ((int16_t)-16384)

static void *alloc32(size_t bytes) { /* aligned_alloc wants a size that is a multiple of the alignment */
    return aligned_alloc(32, (bytes + 31) & ~(size_t)31);
}

static inline __m256i shift_in(__m256i a, int16_t fill) { /* lane l <- lane l-1, lane 0 <- fill */
    __m256i t = _mm256_permute2x128_si256(a, a, 0x08);
    __m256i r = _mm256_alignr_epi8(a, t, 14);
    return _mm256_insert_epi16(r, fill, 0);
}

int fo_sw_striped(const fo_params *p, const char *q, int lq, const char *r, int lr, fo_sw_result *res,
                  uint32_t *ops, int ops_cap) {
    /* the rule switches other than the defaults are served by the scalar version */
    if (p->rules != FO_RULES_DEFAULT || lq <= 0 || lr <= 0 || !__builtin_cpu_supports("avx2") ||
        2L * lq * (p->match > 0 ? p->match : 1) > 16000)
        return fo_sw_trace(p, q, lq, r, lr, res, ops, ops_cap);
    const int nalpha = (int)strlen(p->alphabet);
    int mapper[256];
    for (int c = 0; c < 256; c++) mapper[c] = nalpha;
    for (int k = 0; k < nalpha; k++) {
        mapper[toupper((unsigned char)p->alphabet[k])] = k;
        mapper[tolower((unsigned char)p->alphabet[k])] = k;
    }
    const int segLen = (lq + SEGW - 1) / SEGW;
    const int open = p->open, ext = p->ext;
    /* profile[k][seg] */
    __m256i *prof = (__m256i *)alloc32(sizeof(__m256i) * (size_t)(nalpha + 1) * segLen);
    __m256i *H = (__m256i *)alloc32(sizeof(__m256i) * segLen);
    __m256i *Hprev = (__m256i *)alloc32(sizeof(__m256i) * segLen);
    __m256i *E = (__m256i *)alloc32(sizeof(__m256i) * segLen);
    __m256i *Eo = (__m256i *)alloc32(sizeof(__m256i) * segLen); /* E-origin mask of the current column */
    __m256i *F = (__m256i *)alloc32(sizeof(__m256i) * segLen);
    __m256i *Dg = (__m256i *)alloc32(sizeof(__m256i) * segLen);
    uint8_t *trace = (uint8_t *)alloc32((size_t)lr * segLen * SEGW + 32);
    if (!prof || !H || !Hprev || !E || !Eo || !F || !Dg || !trace) return -1;
    for (int k = 0; k <= nalpha; k++)
        for (int s = 0; s < segLen; s++) {
            int16_t v[SEGW];
            for (int l = 0; l < SEGW; l++) {
                const int row = s + l * segLen;
                if (row >= lq) v[l] = -64; /* padded rows decay */
                else {
                    const int a = mapper[(unsigned char)q[row]];
                    v[l] = (int16_t)((a == nalpha || k == nalpha) ? 0 : (a == k ? p->match : p->mismatch));
                }
            }
            prof[k * segLen + s] = _mm256_loadu_si256((const __m256i *)v);
        }
    const __m256i vZero = _mm256_setzero_si256(), vOpen = _mm256_set1_epi16((short)open),
                  vExt = _mm256_set1_epi16((short)ext), vNeg = _mm256_set1_epi16(NEG16);
    for (int s = 0; s < segLen; s++) {
        H[s] = vZero;
        E[s] = vNeg;       /* E[i][-1] = -inf -> E[i][0] = H[i][-1] - open, "opened" */
    }
    int score = -1, end_q = 0, end_r = 0;
    for (int j = 0; j < lr; j++) {
        const __m256i *pr = prof + mapper[(unsigned char)r[j]] * segLen;
        __m256i *tmp = Hprev; Hprev = H; H = tmp;
        /* ---- main sweep */
        __m256i vHd = shift_in(Hprev[segLen - 1], 0); /* H[i-1][j-1] for segment 0 */
        __m256i vF = vNeg;
        for (int s = 0; s < segLen; s++) {
            /* E[i][j] = max(H[i][j-1] - open, E[i][j-1] - ext); origin decided here for this column */
            const __m256i eopn = _mm256_subs_epi16(Hprev[s], vOpen), eext = _mm256_subs_epi16(E[s], vExt);
            const __m256i vE = _mm256_max_epi16(eopn, eext);
            Eo[s] = _mm256_cmpgt_epi16(eopn, eext);
            E[s] = vE;
            const __m256i vD = _mm256_adds_epi16(vHd, pr[s]);
            Dg[s] = vD;
            __m256i vH = _mm256_max_epi16(_mm256_max_epi16(vD, vE), _mm256_max_epi16(vF, vZero));
            H[s] = vH;
            F[s] = vF;
            vF = _mm256_max_epi16(_mm256_subs_epi16(vH, vOpen), _mm256_subs_epi16(vF, vExt));
            vHd = Hprev[s];
        }
        /* ---- lazy F: carry the column's F across the lane boundary until it cannot matter */
        for (int k = 0; k < SEGW; k++) {
            vF = shift_in(vF, NEG16);
            for (int s = 0; s < segLen; s++) {
                F[s] = _mm256_max_epi16(F[s], vF);
                const __m256i vH = _mm256_max_epi16(H[s], vF);
                H[s] = vH;
                const __m256i opn = _mm256_subs_epi16(vH, vOpen), fext = _mm256_subs_epi16(vF, vExt);
                /* continue while F_ext >= F_opn somewhere (ties extend, Appendix A.4) */
                const __m256i ge = _mm256_or_si256(_mm256_cmpgt_epi16(fext, opn), _mm256_cmpeq_epi16(fext, opn));
                if (!_mm256_movemask_epi8(ge)) goto lazy_done;
                vF = fext;
            }
        }
    lazy_done:;
        /* ---- trace sweep from the final values of this column */
        __m256i vMax = vZero;
        __m256i pH = shift_in(H[segLen - 1], 0), pF = shift_in(F[segLen - 1], NEG16);
        uint8_t *tcol = trace + (size_t)j * segLen * SEGW;
        for (int s = 0; s < segLen; s++) {
            const __m256i vH = H[s], vFv = F[s], vD = Dg[s];
            const __m256i isz = _mm256_cmpeq_epi16(vH, vZero);
            const __m256i isd = _mm256_cmpeq_epi16(vH, vD);
            const __m256i isf = _mm256_cmpeq_epi16(vH, vFv);
            /* dir: zero 0, diag 1, F 2, E 3 */
            __m256i dir = _mm256_set1_epi16(T_E);
            dir = _mm256_blendv_epi8(dir, _mm256_set1_epi16(T_F), isf);
            dir = _mm256_blendv_epi8(dir, _mm256_set1_epi16(T_DIAG), isd);
            dir = _mm256_blendv_epi8(dir, vZero, isz);
            const __m256i fo = _mm256_cmpgt_epi16(_mm256_subs_epi16(pH, vOpen), _mm256_subs_epi16(pF, vExt));
            __m256i t = _mm256_or_si256(dir, _mm256_and_si256(Eo[s], _mm256_set1_epi16(T_EOPEN)));
            t = _mm256_or_si256(t, _mm256_and_si256(fo, _mm256_set1_epi16(T_FOPEN)));
            /* pack 16 x int16 -> 16 x uint8 */
            const __m256i pk = _mm256_packus_epi16(t, t);
            const __m128i lo = _mm256_castsi256_si128(pk), hi = _mm256_extracti128_si256(pk, 1);
            _mm_storeu_si128((__m128i *)(tcol + (size_t)s * SEGW), _mm_unpacklo_epi64(lo, hi));
            vMax = _mm256_max_epi16(vMax, vH);
            pH = vH;
            pF = vFv;
        }
        /* column maximum over real rows: padded rows can only hold copies of earlier columns' values */
        int16_t mx[SEGW];
        _mm256_storeu_si256((__m256i *)mx, vMax);
        int cm = 0;
        for (int l = 0; l < SEGW; l++) if (mx[l] > cm) cm = mx[l];
        if (cm > score) { /* strictly greater: smallest ref index wins; then the smallest query row */
            int best_row = -1;
            const int16_t *hp = (const int16_t *)H;
            for (int s = 0; s < segLen; s++)
                for (int l = 0; l < SEGW; l++) {
                    const int row = s + l * segLen;
                    if (row < lq && hp[s * SEGW + l] == cm && (best_row < 0 || row < best_row)) best_row = row;
                }
            if (best_row >= 0) { score = cm; end_q = best_row; end_r = j; }
        }
    }
    if (score < 0) { score = 0; end_q = 0; end_r = 0; }
    if (score == 0) { /* all-zero matrix: the scalar rule picks (0,0) */
        end_q = 0; end_r = 0;
    }
    /* ---- traceback (Appendix A.4) over the striped table */
    size_t cap = (size_t)lq + (size_t)lr + 4;
    uint32_t *rev = (uint32_t *)malloc(cap * sizeof(uint32_t));
    size_t nrev = 0;
    int i = end_q, j = end_r, state = 0, cur_op = -1;
    uint32_t cur_len = 0;
#define TR(ii, jj) trace[((size_t)(jj) * segLen + (size_t)((ii) % segLen)) * SEGW + (size_t)((ii) / segLen)]
#define EMIT(o) do { if ((o) == cur_op) cur_len++; else { if (cur_op >= 0) rev[nrev++] = (cur_len << 4) | (uint32_t)cur_op; cur_op = (o); cur_len = 1; } } while (0)
    while (i >= 0 && j >= 0) {
        const uint8_t t = TR(i, j);
        if (state == 0) {
            const int d = t & 3;
            if (d == T_ZERO) break;
            if (d == T_DIAG) { EMIT(q[i] == r[j] ? OP_EQ : OP_X); i--; j--; }
            else if (d == T_E) state = 1;
            else state = 2;
        } else if (state == 1) { EMIT(OP_D); j--; state = (t & T_EOPEN) ? 0 : 1; }
        else { EMIT(OP_I); i--; state = (t & T_FOPEN) ? 0 : 2; }
    }
    if (cur_op >= 0) rev[nrev++] = (cur_len << 4) | (uint32_t)cur_op;
#undef EMIT
#undef TR
    res->score = score;
    res->end_query = end_q;
    res->end_ref = end_r;
    res->beg_query = i + 1;
    res->beg_ref = j + 1;
    int n = 0;
    if (res->beg_query > 0) { if (n < ops_cap) ops[n] = ((uint32_t)res->beg_query << 4) | OP_S; n++; }
    for (size_t k = nrev; k-- > 0;) { if (n < ops_cap) ops[n] = rev[k]; n++; }
    if (lq - 1 - end_q > 0) { if (n < ops_cap) ops[n] = ((uint32_t)(lq - 1 - end_q) << 4) | OP_S; n++; }
    res->n_ops = n;
    free(rev); free(prof); free(H); free(Hprev); free(E); free(Eo); free(F); free(Dg); free(trace);
    return 0;
}

/* ------------------------------------------------------------------ batch driver */
typedef struct {
    const fo_params *p; int n, tid, nthreads; const uint8_t *q; const int64_t *q_off;
    const uint8_t *r; const int64_t *r_off; int32_t *res; uint32_t *ops; int max_ops, variant, rc;
} sw_batch_arg;

static void *sw_batch_worker(void *v) {
    sw_batch_arg *a = (sw_batch_arg *)v;
    const int64_t lo = (int64_t)a->n * a->tid / a->nthreads, hi = (int64_t)a->n * (a->tid + 1) / a->nthreads;
    for (int64_t k = lo; k < hi; k++) {
        fo_sw_result res;
        const char *qq = (const char *)a->q + a->q_off[k], *rr = (const char *)a->r + a->r_off[k];
        const int lq = (int)(a->q_off[k + 1] - a->q_off[k]), lr = (int)(a->r_off[k + 1] - a->r_off[k]);
        int rc = a->variant ? fo_sw_striped(a->p, qq, lq, rr, lr, &res, a->ops + (size_t)k * a->max_ops, a->max_ops)
                            : fo_sw_trace(a->p, qq, lq, rr, lr, &res, a->ops + (size_t)k * a->max_ops, a->max_ops);
        if (rc) a->rc = rc;
        int32_t *o = a->res + 6 * k;
        o[0] = res.score; o[1] = res.end_query; o[2] = res.end_ref;
        o[3] = res.beg_query; o[4] = res.beg_ref; o[5] = res.n_ops;
    }
    return NULL;
}

int fo_sw_batch(const fo_params *p, int n, int threads, const uint8_t *q, const int64_t *q_off,
                const uint8_t *r, const int64_t *r_off, int32_t *res, uint32_t *ops, int max_ops,
                int variant) {
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    sw_batch_arg *args = (sw_batch_arg *)malloc(sizeof(sw_batch_arg) * (size_t)threads);
    int rc = 0;
    for (int t = 0; t < threads; t++) {
        args[t] = (sw_batch_arg){p, n, t, threads, q, q_off, r, r_off, res, ops, max_ops, variant, 0};
        if (t > 0) pthread_create(&th[t], NULL, sw_batch_worker, &args[t]);
    }
    sw_batch_worker(&args[0]);
    for (int t = 1; t < threads; t++) pthread_join(th[t], NULL);
    for (int t = 0; t < threads; t++) if (args[t].rc) rc = args[t].rc;
    free(th); free(args);
    return rc;
}
